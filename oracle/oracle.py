"""ctypes wrapper around oracle/libcone_oracle.so  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of cone_oracle.c).  The product path (cvxpylayers_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATUS_NAMES = {1: "solved", 2: "solved_inaccurate", -1: "unbounded", -2: "infeasible",
                -6: "unbounded_inaccurate", -7: "infeasible_inaccurate", -4: "failed", 0: "unfinished"}


class Opts(C.Structure):
    _fields_ = [("eps_abs", C.c_double), ("eps_rel", C.c_double), ("eps_infeas", C.c_double),
                ("alpha", C.c_double), ("rho_x", C.c_double), ("scale", C.c_double),
                ("max_iters", C.c_int), ("normalize", C.c_int), ("adaptive_scale", C.c_int),
                ("adj_mode", C.c_int), ("lsqr_atol", C.c_double), ("lsqr_btol", C.c_double),
                ("lsqr_conlim", C.c_double), ("lsqr_iter_lim", C.c_int), ("warm_start", C.c_int),
                ("aa_mem", C.c_int), ("aa_interval", C.c_int)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libcone_oracle.so")
    src = os.path.join(_HERE, "cone_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcone_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oc_default_opts.argtypes = [C.POINTER(Opts)]
        _LIB.oc_num_threads.restype = C.c_int
    return _LIB


def default_threads() -> int:
    """OpenMP threads of a batch call with nthreads = 0: the runtime's own maximum capped by what this process may use (affinity mask, cgroup CPU quota).  On a box
    with 256 CPUs visible and a quota of 16 the runtime's default is 256 threads whose spinning gets the whole cgroup throttled."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, int(lib().oc_num_threads())))


def make_opts(**kw) -> Opts:
    """Accepts the reference's solver_args keys (eps, max_iters, ...; diffcp maps eps -> eps_abs & eps_rel)."""
    o = Opts()
    lib().oc_default_opts(C.byref(o))
    kw = dict(kw)
    if "eps" in kw:
        e = kw.pop("eps")
        o.eps_abs = e
        o.eps_rel = e
    mode = kw.pop("mode", None)
    if mode is not None:
        o.adj_mode = {"lsqr": 0, "dense": 1, "lsmr": 2}[mode]
    if "acceleration_lookback" in kw:       # SCS names: lookback = memory (0 = off, the default here), interval
        o.aa_mem = int(kw.pop("acceleration_lookback"))
    if "acceleration_interval" in kw:
        o.aa_interval = int(kw.pop("acceleration_interval"))
    kw.pop("verbose", None)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(f"unknown oracle option {k}")
        setattr(o, k, v)
    return o


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _cones(cones):
    q = np.ascontiguousarray(cones.get("q", []), dtype=np.int32)
    s = np.ascontiguousarray(cones.get("s", []), dtype=np.int32)
    pw = np.ascontiguousarray(cones.get("p", []), dtype=np.float64)
    return int(cones.get("z", 0)), int(cones.get("l", 0)), q, s, (int(cones.get("ep", 0)), pw)


def solve_batch(A, b, c, cones, nthreads=0, warm=None, P=None, **opts):
    """A (B,m,n) dense, b (B,m), c (B,n) float64.  Returns dict(x,y,s,iters,status,resid).  warm = (x, y, s): initial point.
    P (B,n,n): optional quadratic objective 1/2 x^T P x (symmetric PSD)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    B, m, n = A.shape
    z, l, q, s, nep = _cones(cones)
    o = make_opts(**opts)
    x = np.empty((B, n)); y = np.empty((B, m)); sv = np.empty((B, m))
    if warm is not None:
        o.warm_start = 1
        x[...] = warm[0]; y[...] = warm[1]; sv[...] = warm[2]
    iters = np.zeros(B, dtype=np.int32); status = np.zeros(B, dtype=np.int32); resid = np.zeros((B, 3))
    Pp = None
    if P is not None:
        P = np.ascontiguousarray(P, dtype=np.float64); Pp = _p(P)
    rc = lib().oc_solve_batch_qp(B, n, m, _p(A), _p(b), _p(c), Pp, z, l, len(q), _p(q, C.c_int), len(s), _p(s, C.c_int), nep[0], len(nep[1]), _p(nep[1]),
                              C.byref(o), _p(x), _p(y), _p(sv), _p(iters, C.c_int), _p(status, C.c_int), _p(resid),
                              int(nthreads) if int(nthreads) > 0 else default_threads())
    if rc != 0:
        raise ValueError("cone dims do not match m")
    return dict(x=x, y=y, s=sv, iters=iters, status=status, resid=resid)


def adjoint_batch(A, b, c, cones, x, y, s, dx, dy, ds=None, nthreads=0, P=None, **opts):
    """diffcp adj_batch restatement: returns dA (B,m,n) dense, db (B,m), dc (B,n), lsqr_iters (B,)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    B, m, n = A.shape
    z, l, q, sd, nep = _cones(cones)
    o = make_opts(**opts)
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    dx = np.ascontiguousarray(dx, dtype=np.float64); dy = np.ascontiguousarray(dy, dtype=np.float64)
    dsp = None
    if ds is not None:
        ds = np.ascontiguousarray(ds, dtype=np.float64)
        dsp = _p(ds)
    dA = np.empty((B, m, n)); db = np.empty((B, m)); dc = np.empty((B, n)); it = np.zeros(B, dtype=np.int32)
    Pp = dPp = None; dP = None
    if P is not None:
        P = np.ascontiguousarray(P, dtype=np.float64); Pp = _p(P)
        dP = np.empty((B, n, n)); dPp = _p(dP)
    rc = lib().oc_adjoint_batch_qp(B, n, m, _p(A), _p(b), _p(c), Pp, z, l, len(q), _p(q, C.c_int), len(sd), _p(sd, C.c_int), nep[0], len(nep[1]), _p(nep[1]),
                                C.byref(o), _p(x), _p(y), _p(s), _p(dx), _p(dy), dsp, _p(dA), _p(db), _p(dc), dPp,
                                   _p(it, C.c_int), int(nthreads) if int(nthreads) > 0 else default_threads())
    if rc != 0:
        raise ValueError("cone dims do not match m")
    out = dict(dA=dA, db=db, dc=dc, lsqr_iters=it)
    if dP is not None:
        out["dP"] = dP
    return out


def proj_exp(v, dual=False):
    """Projection of v (3,) onto the exponential cone (or its dual)."""
    w = np.array(v, dtype=np.float64)
    lib().oc_proj_exp(_p(w), int(dual))
    return w


def dproj_exp(v, dual=False):
    """Jacobian (3,3) of that projection."""
    w = np.array(v, dtype=np.float64); J = np.zeros((3, 3))
    lib().oc_dproj_exp(_p(w), int(dual), _p(J))
    return J


def proj_pow(v, a, dual=False):
    """Projection of v (3,) onto the power cone K_a (a > 0) / K_|a|^* (a < 0), or with dual=True onto that cone's dual."""
    w = np.array(v, dtype=np.float64)
    lib().oc_proj_pow(_p(w), C.c_double(a), int(dual))
    return w


def dproj_pow(v, a, dual=False):
    w = np.array(v, dtype=np.float64); J = np.zeros((3, 3))
    lib().oc_dproj_pow(_p(w), C.c_double(a), int(dual), _p(J))
    return J


def num_threads() -> int:
    return int(lib().oc_num_threads())
