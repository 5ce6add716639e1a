"""ctypes binding of the C ABI declared in include/cone_engine.h (csrc/libcone_engine.so).

There is NO fallback: if the HIP library is missing or fails to load this raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CE_ENGINE_SO") or os.path.join(_HERE, "csrc", "libcone_engine.so")     # CE_ENGINE_SO: a debug build (csrc: make timing)

# every symbol include/cone_engine.h declares
SYMBOLS = ["ce_abi_version", "ce_struct_size", "ce_acceleration_available", "ce_default_settings", "ce_create", "ce_destroy", "ce_last_error", "ce_solve", "ce_vjp", "ce_solve_shared_a", "ce_vjp_shared_a", "ce_vjp_lsqr", "ce_qp_native", "ce_solve_qp", "ce_vjp_qp",
           "ce_transpose", "ce_status_summary", "ce_parammap_apply", "ce_parammap_apply2", "ce_ca_step", "ce_ca_check", "ce_ca_psd", "ce_ca_psd_mfma", "ce_ca_triples", "ce_ca_triple_jac", "ce_ca_update", "ce_ca_finish", "ce_set_profiling", "ce_get_profile", "ce_reset_profile", "ce_get_launch_info", "ce_set_dispatch_history", "ce_set_adjoint_resolve", "ce_adjoint_ns_variant", "ce_set_lsqr_variant"]


class CeTemplate(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("nnz_aug", C.c_int), ("indices", C.POINTER(C.c_int)),
                ("indptr", C.POINTER(C.c_int)), ("z", C.c_int), ("l", C.c_int), ("nq", C.c_int),
                ("q", C.POINTER(C.c_int)), ("ns", C.c_int), ("s", C.POINTER(C.c_int)), ("nep", C.c_int),
                ("np", C.c_int), ("p", C.POINTER(C.c_double)),
                ("nnz_p", C.c_int), ("p_indices", C.POINTER(C.c_int)), ("p_indptr", C.POINTER(C.c_int))]


class CeSettings(C.Structure):
    _fields_ = [("eps_abs", C.c_double), ("eps_rel", C.c_double), ("eps_infeas", C.c_double), ("alpha", C.c_double),
                ("rho_x", C.c_double), ("scale", C.c_double), ("max_iters", C.c_int), ("normalize", C.c_int),
                ("adaptive_scale", C.c_int), ("warm_start", C.c_int),
                ("acceleration_lookback", C.c_int), ("acceleration_interval", C.c_int)]


ABI_VERSION = 11         # include/cone_engine.h CE_ABI_VERSION this binding was written against


def build(force: bool = False) -> str:
    """csrc/Makefile decides what is stale (every kernel header is a dependency of the objects that include it); force = rebuild all."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(csrc, "build.sh")], stdout=subprocess.DEVNULL)
    return SO_PATH


_LIB = None


def lib():
    """Loads csrc/libcone_engine.so.  Raises (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(cvxpylayers_amd has no CPU fallback).")
    L = C.CDLL(SO_PATH)
    # ABI guard: a stale .so (or a stale binding) must be rejected, not read out of bounds
    try:
        ver, st, ss = L.ce_abi_version(), L.ce_struct_size(0), L.ce_struct_size(1)
    except AttributeError as e:
        raise RuntimeError(f"{SO_PATH} predates the ABI guard (no ce_abi_version): rebuild it") from e
    if ver != ABI_VERSION or st != C.sizeof(CeTemplate) or ss != C.sizeof(CeSettings):
        raise RuntimeError(f"{SO_PATH}: ABI mismatch (library version {ver}, ce_template {st} B, ce_settings {ss} B; binding version "
                           f"{ABI_VERSION}, {C.sizeof(CeTemplate)} B, {C.sizeof(CeSettings)} B): rebuild the library")
    vp, dp, ip, lg = C.c_void_p, C.c_void_p, C.c_void_p, C.c_long
    L.ce_default_settings.argtypes = [C.POINTER(CeSettings)]
    L.ce_default_settings.restype = None
    L.ce_create.argtypes = [C.POINTER(CeTemplate), C.c_int, C.POINTER(vp)]
    L.ce_destroy.argtypes = [vp]
    L.ce_last_error.restype = C.c_char_p
    L.ce_solve.argtypes = [vp, C.c_int, dp, lg, lg, dp, lg, lg, C.POINTER(CeSettings), dp, dp, dp, ip, ip, dp, vp]
    L.ce_vjp.argtypes = [vp, C.c_int, dp, lg, lg, dp, lg, lg, dp, dp, dp, dp, dp, dp, lg, lg, dp, lg, lg, ip, vp]
    L.ce_solve_shared_a.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, ip, ip, dp, ip, ip, dp, dp, dp, dp, dp, dp, dp, dp, C.POINTER(CeSettings),
                                    dp, dp, dp, dp, dp, dp, ip, ip, dp, vp]
    L.ce_vjp_shared_a.argtypes = [vp, C.c_int, dp, lg, dp, lg, lg, dp, dp, dp, dp, dp, dp, dp, lg, lg, ip, ip, C.c_double, C.c_double, C.c_double, C.c_int, vp]
    L.ce_vjp_lsqr.argtypes = [vp, C.c_int, dp, lg, dp, lg, lg, dp, dp, dp, dp, dp, dp, dp, lg, lg, ip, ip, C.c_double, C.c_double, C.c_double, C.c_int, vp]
    L.ce_qp_native.argtypes = [vp]
    L.ce_acceleration_available.argtypes = [vp]
    L.ce_solve_qp.argtypes = [vp, C.c_int, dp, lg, lg, dp, lg, lg, dp, C.POINTER(CeSettings), dp, dp, dp, ip, ip, dp, vp]
    L.ce_vjp_qp.argtypes = [vp, C.c_int, dp, lg, lg, dp, dp, dp, dp, dp, dp, dp, lg, lg, dp, lg, lg, dp, ip, vp]
    L.ce_transpose.argtypes = [vp, C.c_int, C.c_int, dp, dp, vp]
    L.ce_status_summary.argtypes = [vp, C.c_int, ip, ip, vp]
    L.ce_parammap_apply.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, lg, dp, lg, vp]
    L.ce_parammap_apply2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, lg, dp, lg, vp]
    L.ce_ca_step.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, lg, dp, lg, dp, dp, dp, dp, ip, C.c_int, C.c_int, C.c_double, vp]
    L.ce_ca_check.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(CeSettings), dp, dp, dp, dp, lg, dp, lg, dp, dp, dp, dp, dp, dp, dp,
                              dp, dp, ip, ip, ip, ip, ip, dp, ip, vp]
    L.ce_ca_psd.argtypes = [vp, C.c_int, C.c_int, dp, ip, vp]
    L.ce_ca_psd_mfma.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_int, ip, vp]
    L.ce_ca_triples.argtypes = [vp, C.c_int, C.c_int, dp, dp, ip, vp]
    L.ce_ca_triple_jac.argtypes = [vp, C.c_int, dp, lg, dp, vp]
    L.ce_ca_update.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, ip, C.c_int, C.c_double, vp]
    L.ce_ca_finish.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, ip, dp, dp, dp, vp]
    L.ce_set_profiling.argtypes = [vp, C.c_int]
    L.ce_get_profile.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.ce_reset_profile.argtypes = [vp]
    L.ce_set_dispatch_history.argtypes = [vp, C.c_int]
    L.ce_adjoint_ns_variant.argtypes = [vp]
    L.ce_set_adjoint_resolve.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]
    L.ce_set_lsqr_variant.argtypes = [vp, C.c_int]; L.ce_set_lsqr_variant.restype = C.c_int
    L.ce_get_launch_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _LIB = L
    return L


class EngineError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().ce_last_error()
        raise EngineError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
