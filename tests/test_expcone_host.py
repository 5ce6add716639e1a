"""The device exponential-cone routines (cvxpylayers_amd/csrc/ce_expcone.h: bracketed Newton with a warm start, Jacobian,
3x3 eigen-rotation) are plain C++ apart from their qualifiers, so their logic is exercised here on the host -- compiled with g++
and compared with the oracle's bisection-based restatement on random points of every case, with arbitrary warm starts."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <cmath>
#define __device__
#define __forceinline__ inline
#define __noinline__
#include "ce_expcone.h"
using std::fmin; using std::fmax;
extern "C" {
void h_proj(double *v, double rho0, int dual, double *rho_out) {
    if (dual) { double r = rho0; exp_project_dual(v, &r); *rho_out = r; }
    else { ExpInfo inf; exp_project(v[0], v[1], v[2], rho0, inf); *rho_out = inf.rho; }
}
void h_dproj(const double *v, double *J) { exp_dproject(v, J); }
void h_eig(const double *v, double *W, double *th) { exp_dual_eig(v, W, th); }
void h_pproj(double *v, double a, double r0) { double r = r0; pow_project_dual_of_entry(v, a, &r); }
void h_pdproj(const double *v, double a, double *J) { pow_dproject(v, a, J); }
void h_peig(const double *v, double a, double *W, double *th) { pow_dual_eig(v, a, W, th); }
}
'''


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("expcone")
    (d / "host.cpp").write_text(SRC)
    so = str(d / "libexpcone_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "cvxpylayers_amd", "csrc"), "-o", so, str(d / "host.cpp")])
    L = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    L.h_proj.argtypes = [dp, C.c_double, C.c_int, dp]; L.h_dproj.argtypes = [dp, dp]; L.h_eig.argtypes = [dp, dp, dp]
    L.h_pproj.argtypes = [dp, C.c_double, C.c_double]; L.h_pdproj.argtypes = [dp, C.c_double, dp]; L.h_peig.argtypes = [dp, C.c_double, dp, dp]
    return L


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_device_exp_projection_matches_oracle(host):
    rng = np.random.default_rng(0)
    for t in range(4000):
        v = rng.standard_normal(3) * 10 ** rng.uniform(-3, 3, size=3 if t % 2 else 1)
        for dual in (0, 1):
            w = v.copy(); rho = np.zeros(1)
            host.h_proj(_p(w), float(rng.standard_normal() * 5), dual, _p(rho))      # any warm start must give the same point
            ref = oracle.proj_exp(v, dual=bool(dual))
            assert np.abs(w - ref).max() <= 1e-12 * (1 + np.linalg.norm(v)), (v, w, ref)
        J = np.zeros((3, 3)); host.h_dproj(_p(v.copy()), _p(J))
        assert np.abs(J - oracle.dproj_exp(v)).max() < 1e-8, v
        W = np.zeros((3, 3)); th = np.zeros(3)
        host.h_eig(_p(v.copy()), _p(W), _p(th))
        Sd = oracle.dproj_exp(v, dual=True); Sd = (Sd + Sd.T) / 2
        assert np.abs(W @ np.diag(th) @ W.T - Sd).max() < 1e-8 and np.abs(W.T @ W - np.eye(3)).max() < 1e-12, v


def test_device_exp_projection_special_points(host):
    for v in ([0.0, 0.0, 0.0], [0.0, 1.0, 1.0], [1.0, 0.0, 0.0], [-1.0, 0.0, 2.0], [0.0, -1.0, 1.0], [0.0, 0.0, -1.0], [5.0, -3.0, 0.0],
              [-0.144356093, 1.42172124e-04, -1.18506858e-02], [700.0, 1.0, 1.0], [-1e6, 1e-6, 1.0]):
        v = np.array(v); w = v.copy(); rho = np.zeros(1)
        host.h_proj(_p(w), 0.0, 0, _p(rho))
        assert np.isfinite(w).all()
        assert np.abs(w - oracle.proj_exp(v)).max() <= 1e-10 * (1 + np.linalg.norm(v)), (v, w, oracle.proj_exp(v))
        J = np.zeros((3, 3)); host.h_dproj(_p(v.copy()), _p(J))
        assert np.isfinite(J).all()


def test_device_power_cone_routines_match_oracle(host):
    rng = np.random.default_rng(0)
    for t in range(4000):
        a = rng.uniform(0.05, 0.95) * (1 if t % 3 else -1)            # negative entry: the dual cone (SCS convention)
        v = rng.standard_normal(3) * 10 ** rng.uniform(-2, 2)
        w = v.copy()
        host.h_pproj(_p(w), a, float(abs(rng.standard_normal())))     # projection onto the dual of the entry's cone, any warm start
        ref = oracle.proj_pow(v, a, dual=True)
        assert np.abs(w - ref).max() <= 1e-12 * (1 + np.linalg.norm(v)), (v, a, w, ref)
        if a > 0:
            J = np.zeros((3, 3)); host.h_pdproj(_p(v.copy()), a, _p(J))
            assert np.abs(J - oracle.dproj_pow(v, a)).max() < 1e-6, (v, a)
        W = np.zeros((3, 3)); th = np.zeros(3)
        host.h_peig(_p(v.copy()), a, _p(W), _p(th))
        Sd = oracle.dproj_pow(v, a, dual=True); Sd = (Sd + Sd.T) / 2
        assert np.abs(W @ np.diag(th) @ W.T - Sd).max() < 1e-6 and np.abs(W.T @ W - np.eye(3)).max() < 1e-12, (v, a)


MATH_SRC = r'''
#include <cmath>
#include <cstring>
#define __device__
#define __forceinline__ inline
#define CE_MATH_HOST
static inline int ce_fexp(double x) { int e; std::frexp(x, &e); return e; }
static inline double ce_fmant(double x) { int e; return std::frexp(x, &e); }
static inline double ce_bits(unsigned long long b) { double d; std::memcpy(&d, &b, 8); return d; }
#define CE_FREXP_EXP(x) ce_fexp(x)
#define CE_FREXP_MANT(x) ce_fmant(x)
#define CE_BITS_TO_DOUBLE(b) ce_bits(b)
using std::rint; using std::ldexp;
#include "ce_math.h"
extern "C" {
void h_logexp(const double *x, int n, double *lg, double *ex) {
    double tab[CE_MATH_TAB];
    for (int t = 0; t < CE_MATH_TAB; t++) ce_math_table_init(tab, t);
    for (int i = 0; i < n; i++) { lg[i] = x[i] > 0 ? ce_log(x[i], tab) : 0.0; ex[i] = std::fabs(x[i]) < 700 ? ce_exp(x[i], tab) : 0.0; }
}
}
'''


def test_table_log_exp_match_libm(tmp_path):
    """ce_math.h (the adaptive-scale update's log / exp with coefficients in an LDS table): within 1 ulp-class error of libm over
    the ranges the solver feeds them (ratios of residuals, 1e-300 .. 1e300; exponents of a few tens)."""
    (tmp_path / "m.cpp").write_text(MATH_SRC)
    so = str(tmp_path / "libm_host.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "cvxpylayers_amd", "csrc"), "-o", so, str(tmp_path / "m.cpp")])
    L = C.CDLL(so)
    rng = np.random.default_rng(0)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 20000), rng.uniform(0.4, 2.5, 20000), [5e-324, 1e-310, 1.0, 2.0, 0.5]])
    lg = np.zeros_like(x); ex = np.zeros_like(x)
    L.h_logexp(_p(x), len(x), _p(lg), _p(ex))
    assert np.abs(lg - np.log(x)).max() <= 4e-16 * np.abs(np.log(x)).max() and np.abs((lg - np.log(x))[-3:]).max() <= 1.2e-16
    y = np.concatenate([rng.uniform(-60, 60, 20000), rng.uniform(-1e-3, 1e-3, 1000), [0.0]])
    lg = np.zeros_like(y); ex = np.zeros_like(y)
    L.h_logexp(_p(y), len(y), _p(lg), _p(ex))
    assert (np.abs(ex - np.exp(y)) / np.exp(y)).max() <= 4e-16
