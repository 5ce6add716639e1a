"""MI355 solver plugin: the HIP engine behind cvxpylayers' solver-plugin boundary.

Mirrors the reference DIFFCP plugin  cvxpylayers/interfaces/diffcp_if.py  (same constructor
arguments as DIFFCP_ctx :105-120, same `_CvxpyLayer.apply(P_eval, q_eval, A_eval, cl_ctx,
solver_args, needs_grad, warm_start) -> (primal, dual, aux, data)` convention :329-377 and the same
7-tuple backward :385-403), but every instance is solved and differentiated on the GPU by
csrc/libcone_engine.so through the C ABI in include/cone_engine.h.  No CPU fallback exists.

Not thread-safe (engines are created lazily and cached on MI355_ctx), like moreau_if.py:14-15.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Any

import numpy as np
import torch

from cvxpylayers_amd import _lib

try:  # subclass diffcp.SolverError when diffcp is importable so `pytest.raises(diffcp.SolverError)` keeps working
    import diffcp as _diffcp  # type: ignore

    _SolverErrorBase = _diffcp.SolverError
except Exception:  # pragma: no cover - diffcp is not installed in this image
    _SolverErrorBase = Exception


class SolverError(_SolverErrorBase):
    """Raised when any instance of the batch is infeasible / unbounded / failed
    (reference contract: tests/test_torch.py:299-316 expects diffcp.SolverError for the whole batch)."""


STATUS_NAMES = {1: "Solved", 2: "Solved/Inaccurate", -1: "Unbounded", -2: "Infeasible", -6: "Unbounded/Inaccurate",
                -7: "Infeasible/Inaccurate", -4: "Failed", 0: "Unfinished"}

_KNOWN_ARGS = {"eps", "eps_abs", "eps_rel", "eps_infeas", "max_iters", "alpha", "rho_x", "scale", "normalize",
               "adaptive_scale", "acceleration_lookback", "acceleration_interval", "verbose", "mode", "solve_method",
               "n_jobs_forward", "n_jobs_backward", "warm_starts", "raise_on_error", "dispatch_history",
               "lsqr_atol", "lsqr_btol", "lsqr_iter_lim", "adjoint_system"}

# Stopping rule of the LSQR adjoint (shared-A templates).  diffcp's adjoint (diffcp_if.py:86 -> adj_batch, mode="lsqr") runs LSQR with atol = btol = 1e-8 and an
# iteration limit of 2 N on its N = n + m + 1 operator; the oracle restates exactly that (oracle/cone_oracle.c:85,712).  solver_args may override:
# lsqr_atol / lsqr_btol / lsqr_iter_lim (callers that need gradients to 1e-5 against a direct elimination pass tight values explicitly).
# adjoint_system: "full" (default) = diffcp's (n + m + 1) system M^T r = dz, tau row and column included -- LSQR then returns diffcp's minimum-norm element on
# rank-deficient systems and takes diffcp's number of iterations;  "reduced" = r_tau pinned to 0 (the system of rounds 1-4: the same gradients wherever the
# system is regular and the point accurate, a quarter of the LSQR iterations at loose eps, where the full system is nearly singular AND inconsistent).
LSQR_ATOL, LSQR_BTOL = 1e-8, 1e-8


def lsqr_rule(merged_args: dict, n: int, m: int) -> tuple:
    """(atol, btol, iter_lim, system, method) of the iterative adjoint from merged solver_args; defaults = diffcp's.  method: "lsqr", or "lsmr" for diffcp's mode="lsmr"
    (the same operator and tolerances under Fong & Saunders' LSMR recurrences and stopping tests: ce_set_lsqr_variant)"""
    lim = merged_args.get("lsqr_iter_lim")
    system = str(merged_args.get("adjoint_system", "full"))
    if system not in ("full", "reduced"):
        raise ValueError(f"MI355 solver: adjoint_system must be 'full' or 'reduced', got {system!r}")
    return (float(merged_args.get("lsqr_atol", LSQR_ATOL)), float(merged_args.get("lsqr_btol", LSQR_BTOL)),
            int(lim) if lim not in (None, 0) else 2 * (n + m + 1), system, "lsmr" if str(merged_args.get("mode", "")) == "lsmr" else "lsqr")


def unpack_rule(lsqr, n: int, m: int) -> tuple:
    """(atol, btol, iter_lim, system, method) from a rule of three to five entries (callers of ConeEngine.vjp pass what they care about); None = diffcp's defaults"""
    t = tuple(lsqr) if lsqr is not None else lsqr_rule({}, n, m)
    return t + ("full", "lsqr")[len(t) - 3:] if len(t) < 5 else t[:5]



def adjoint_mode(merged_args: dict) -> str:
    """diffcp's `mode` (adj_batch / solve_and_derivative_batch; diffcp_if.py:86 runs its default "lsqr") for PER-INSTANCE-A templates:
    absent -> "direct": the rank-revealing elimination (k_backward_rt / k_backward: the same gradients as LSQR wherever the adjoint system is regular) and,
    behind it on the device, diffcp's LSQR for exactly the instances the elimination found RANK DEFICIENT (ce_vjp with q_vals; include/cone_engine.h) -- the
    default answer is diffcp's minimum-norm element everywhere, regular instances pay nothing;
    "dense" -> "dense": the elimination alone (a basic solution on rank-deficient systems);
    "lsqr" -> diffcp's LSQR on the full (n + m + 1) system with its stopping rule for every instance (ce_vjp_lsqr).  Shared-A templates run LSQR whatever the mode says."""
    mode = str(merged_args.get("mode", ""))
    return "lsqr" if mode in ("lsqr", "lsmr") else ("dense" if mode == "dense" else "direct")          # ("lsmr": the iterative path with LSMR's recurrences, lsqr_rule()[4])


_WARNED: set = set()


def _warn_once(key: str, msg: str):
    """one warning per process and topic (the plugin is called once per training step: repeating it would drown the log)"""
    if key not in _WARNED:
        _WARNED.add(key)
        warnings.warn(msg, stacklevel=3)


def note_ignored_args(merged_args: dict, explicit_lookback: bool):
    """The reference's solver arguments this plugin ACCEPTS but does not act on, said once instead of swallowed silently (diffcp_if.py:356-367 forwards them to
    diffcp / SCS):  acceleration_lookback > 1 -- the kernels keep ONE secant pair whatever the lookback (SCS keeps `lookback` pairs; same fixed point,
    iteration counts within 2.5 % on the BASELINE configurations);  mode other than "lsqr" / "dense" (diffcp's "lsmr") and solve_method -- see adjoint_mode();
    n_jobs_forward / n_jobs_backward -- the batch runs on the GPU."""
    lb = merged_args.get("acceleration_lookback")
    if explicit_lookback and lb is not None and int(lb) > 1:      # (the DEFAULT configuration stays silent -- valid calls must survive `-W error`; info["acceleration"] and the docs carry the one-pair fact)
        _warn_once("lookback", f"MI355 solver: acceleration_lookback={int(lb)}" + ("" if explicit_lookback else " (SCS's default, which the reference forwards)") +
                   " runs as type-I Anderson acceleration with a ONE-pair history (memory 1), not a " + str(int(lb)) + "-pair history; "
                   "pass acceleration_lookback=1 to say so explicitly, 0 to iterate plainly")
    for k in ("mode", "solve_method", "n_jobs_forward", "n_jobs_backward"):
        if k == "mode" and str(merged_args.get(k)) in ("lsqr", "lsmr", "dense"):          # acted on: adjoint_mode()
            continue
        if k in merged_args:
            _warn_once(k, f"MI355 solver: solver_args[{k!r}]={merged_args[k]!r} is accepted for compatibility with the DIFFCP plugin and ignored "
                          "(the adjoint method is fixed per template, the batch is solved on the GPU)")


def dims_to_solver_dict(dims) -> dict:
    """ConeDims (attrs zero/nonneg/soc/exp/psd/p3d) or an SCS-style dict -> {"z","l","q","ep","s","p"}
    (what cvxpy.reductions.solvers.conic_solvers.scs_conif.dims_to_solver_dict returns; diffcp_if.py:8,150)."""
    if isinstance(dims, dict):
        return {"z": int(dims.get("z", dims.get("f", 0))), "l": int(dims.get("l", 0)), "q": [int(v) for v in dims.get("q", [])],
                "ep": int(dims.get("ep", 0)), "s": [int(v) for v in dims.get("s", [])], "p": list(dims.get("p", []))}
    return {"z": int(dims.zero), "l": int(dims.nonneg), "q": [int(v) for v in dims.soc], "ep": int(getattr(dims, "exp", 0)),
            "s": [int(v) for v in getattr(dims, "psd", [])], "p": list(getattr(dims, "p3d", []))}


def make_settings(merged_args: dict) -> _lib.CeSettings:
    """solver_args (SCS / diffcp keyword names) -> ce_settings.  diffcp maps `eps` to eps_abs and eps_rel."""
    unknown = set(merged_args) - _KNOWN_ARGS
    if unknown:
        raise ValueError(f"MI355 solver: unknown solver_args {sorted(unknown)}")
    s = _lib.CeSettings()
    _lib.lib().ce_default_settings(C.byref(s))
    a = dict(merged_args)
    if "eps" in a:
        s.eps_abs = s.eps_rel = float(a["eps"])
    for k in ("eps_abs", "eps_rel", "eps_infeas", "alpha", "rho_x", "scale"):
        if k in a:
            setattr(s, k, float(a[k]))
    for k in ("max_iters", "normalize", "adaptive_scale"):
        if k in a:
            setattr(s, k, int(a[k]))
    # Anderson acceleration: ce_default_settings carries SCS's defaults (lookback 10, interval 10; diffcp forwards them,
    # diffcp_if.py:356-367); acceleration_lookback=0 switches it off.  The kernels keep a one-pair history whatever the lookback.
    if a.get("acceleration_lookback") is not None:
        s.acceleration_lookback = max(int(a["acceleration_lookback"]), 0)
    if a.get("acceleration_interval") not in (0, None):
        s.acceleration_interval = int(a["acceleration_interval"])
    return s


class ConeEngine:
    """Owns one ce_handle (one template, one device)."""

    def __init__(self, indices, indptr, n, m, cone_dict, device: torch.device, p_structure=None):
        L = _lib.lib()
        self.device = device
        self.n, self.m = int(n), int(m)
        self._indices = np.ascontiguousarray(indices, dtype=np.int32)
        self._indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.nnz_aug = int(self._indptr[-1])
        self.nnzA = int(self._indptr[self.n])
        self.cone_dict = {k: (list(v) if isinstance(v, (list, tuple, np.ndarray)) else v) for k, v in dict(cone_dict).items()}
        q = np.ascontiguousarray(cone_dict.get("q", []), dtype=np.int32)
        s = np.ascontiguousarray(cone_dict.get("s", []), dtype=np.int32)
        t = _lib.CeTemplate()
        t.n, t.m, t.nnz_aug = self.n, self.m, self.nnz_aug
        t.indices = self._indices.ctypes.data_as(C.POINTER(C.c_int))
        t.indptr = self._indptr.ctypes.data_as(C.POINTER(C.c_int))
        t.z, t.l = int(cone_dict.get("z", 0)), int(cone_dict.get("l", 0))
        t.nq, t.q = len(q), q.ctypes.data_as(C.POINTER(C.c_int))
        t.ns, t.s = len(s), s.ctypes.data_as(C.POINTER(C.c_int))
        self._pw = np.ascontiguousarray(cone_dict.get("p", []), dtype=np.float64)
        t.nep, t.np = int(cone_dict.get("ep", 0)), len(self._pw)
        t.p = self._pw.ctypes.data_as(C.POINTER(C.c_double))
        self.nnz_p = 0
        if p_structure is not None:       # quadratic objective: CSC structure of P (n x n)
            self._p_idx = np.ascontiguousarray(p_structure[0], dtype=np.int32)
            self._p_ptr = np.ascontiguousarray(p_structure[1], dtype=np.int32)
            self.nnz_p = int(self._p_ptr[-1])
            t.nnz_p = self.nnz_p
            t.p_indices = self._p_idx.ctypes.data_as(C.POINTER(C.c_int))
            t.p_indptr = self._p_ptr.ctypes.data_as(C.POINTER(C.c_int))
        h = C.c_void_p()
        rc = L.ce_create(C.byref(t), device.index or 0, C.byref(h))
        if rc == -2:
            raise NotImplementedError(L.ce_last_error().decode())
        _lib.check(rc, "ce_create")
        self._h = h
        self.qp_native = bool(self.nnz_p) and bool(L.ce_qp_native(h))     # P runs inside the kernels (else: epigraph form upstream)

    def set_dispatch_history(self, on: bool):
        """Longest-first dispatch from the previous call's iteration counts (include/cone_engine.h ce_set_dispatch_history): a scheduling hint, results are
        bit-identical either way."""
        _lib.check(_lib.lib().ce_set_dispatch_history(self._h, int(bool(on))), "ce_set_dispatch_history")
        self.dispatch_history = bool(on)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ce_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def to_batch_major(self, A_eval: torch.Tensor) -> torch.Tensor:
        """(nnz_aug, B) in any layout -> (B, nnz_aug) contiguous fp64 (zero-copy when it already is)."""
        K, B = A_eval.shape
        if A_eval.dtype != torch.float64:
            A_eval = A_eval.double()
        if B == 0:
            return A_eval.new_empty((0, K))
        if A_eval.stride(0) == 1 and (A_eval.stride(1) == K or B == 1):
            return A_eval.t()
        if not A_eval.is_contiguous():
            A_eval = A_eval.contiguous()
        out = torch.empty((B, K), dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().ce_transpose(self._h, K, B, A_eval.data_ptr(), out.data_ptr(), self._stream()), "ce_transpose")
        return out

    def solve(self, A_bm: torch.Tensor, q_eval: torch.Tensor, settings, warm=None, P_bm=None):
        """A_bm (B, nnz_aug) contiguous, q_eval (n+1, B) any strides.  Returns x, y, s, iters, status, resid.
        warm = (x, y, s) of shapes (B, n), (B, m), (B, m): initial point (instances with non-finite entries start cold)."""
        B = A_bm.shape[0]
        dev = self.device
        if B == 0:        # empty batch: nothing to launch (the C ABI refuses B <= 0)
            f64 = dict(dtype=torch.float64, device=dev)
            self.last_path = "per_instance"
            return (torch.empty((0, self.n), **f64), torch.empty((0, self.m), **f64), torch.empty((0, self.m), **f64),
                    torch.empty((0,), dtype=torch.int32, device=dev), torch.empty((0,), dtype=torch.int32, device=dev), torch.empty((0, 3), **f64))
        if warm is not None:
            if tuple(warm[0].shape) != (B, self.n) or tuple(warm[1].shape) != (B, self.m) or tuple(warm[2].shape) != (B, self.m):
                raise ValueError(f"warm start: expected x {(B, self.n)}, y {(B, self.m)}, s {(B, self.m)}, got "
                                 f"{tuple(warm[0].shape)}, {tuple(warm[1].shape)}, {tuple(warm[2].shape)}")
        if P_bm is not None and not self.qp_native:
            raise RuntimeError("quadratic objective on an engine without native P support (use the epigraph form)")
        # direct engine users (tests, probes): vjp() without q_eval may use the objective of the most recent solve() -- but only for the SAME value buffer
        # (the plugin always passes q_eval explicitly; a q paired with another call's A would silently change the full adjoint system: ADVICE round 5)
        self._last_q, self._last_q_key = q_eval.detach(), (A_bm.data_ptr(), B)
        if P_bm is None and self._use_const_a(A_bm):
            from cvxpylayers_amd.interfaces.const_a import solve_const_a
            self.last_path = "const_a"
            return solve_const_a(self, A_bm, q_eval, settings, warm=warm)      # (notes itself whether its kernel honours the acceleration)
        self.last_path = "per_instance"
        self._note_acceleration(settings, honoured=bool(_lib.lib().ce_acceleration_available(self._h)), path="size-generic forward kernels")
        if warm is not None:       # the engine reads the initial point from the output buffers (ce_settings.warm_start)
            x, y, s = (t.detach().to(device=dev, dtype=torch.float64).clone().contiguous() for t in warm)
            settings.warm_start = 1
        else:
            x = torch.empty((B, self.n), dtype=torch.float64, device=dev)
            y = torch.empty((B, self.m), dtype=torch.float64, device=dev)
            s = torch.empty((B, self.m), dtype=torch.float64, device=dev)
            settings.warm_start = 0
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        resid = torch.empty((B, 3), dtype=torch.float64, device=dev)
        if P_bm is not None:
            rc = _lib.lib().ce_solve_qp(self._h, B, A_bm.data_ptr(), 1, self.nnz_aug, q_eval.data_ptr(), q_eval.stride(0),
                                        q_eval.stride(1), P_bm.data_ptr(), C.byref(settings), x.data_ptr(), y.data_ptr(), s.data_ptr(),
                                        iters.data_ptr(), status.data_ptr(), resid.data_ptr(), self._stream())
        else:
            rc = _lib.lib().ce_solve(self._h, B, A_bm.data_ptr(), 1, self.nnz_aug, q_eval.data_ptr(), q_eval.stride(0),
                                     q_eval.stride(1), C.byref(settings), x.data_ptr(), y.data_ptr(), s.data_ptr(),
                                     iters.data_ptr(), status.data_ptr(), resid.data_ptr(), self._stream())
        _lib.check(rc, "ce_solve")
        return x, y, s, iters, status, resid

    def _recent_q(self, A_bm):
        lq, key = getattr(self, "_last_q", None), getattr(self, "_last_q_key", None)
        return lq if (lq is not None and key == (A_bm.data_ptr(), A_bm.shape[0])) else None

    def zeros_like_cached(self, t: torch.Tensor) -> torch.Tensor:
        """A read-only zero tensor of t's shape on this engine's device, allocated once per shape (the cotangent of an output the loss does not use)."""
        cache = self.__dict__.setdefault("_zero_cache", {})
        key = tuple(t.shape)
        z = cache.get(key)
        if z is None:
            if len(cache) > 8:
                cache.clear()
            z = cache[key] = torch.zeros(key, dtype=torch.float64, device=self.device)
        return z

    def enqueue_summary(self, vec: torch.Tensor, slot: int):
        """ce_status_summary of an int32 device vector into pinned slot `slot` (0: status of this forward; 1, 2: adjoint flags of backward calls);
        read with read_summaries() once the LAST one enqueued is ready."""
        if getattr(self, "_summary_host", None) is None:
            self._summary_host = torch.zeros((3, 4), dtype=torch.int32).pin_memory()          # slot 0: status of a forward; slots 1, 2: adjoint flags of backward calls, alternating
            self._summary_np = self._summary_host.numpy()          # (shares the pinned memory)
            self._summary_vec = [None, None, None]
        self._summary_np[slot, 3] = 0                               # "ready" flag, set by the device after the three values
        self._summary_last_slot = slot
        self._summary_vec[slot] = vec                               # (kept for ensure_summary: the counts can be recomputed from the vector if the device's stores never arrive)
        stream = torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib().ce_status_summary(self._h, int(vec.numel()), vec.data_ptr(), self._summary_host[slot].data_ptr(), C.c_void_p(stream.cuda_stream)), "ce_status_summary")

    def ensure_summary(self, slot: int):
        """Call with the stream drained.  The three counts of a slot are only valid once its ready flag is set; a flag that is still clear AFTER a synchronisation
        means the device's stores did not reach this buffer (a mapping of the pinned buffer that went stale, a box misbehaving): then the counts are recomputed
        from the device vector itself -- correctness does not hang on the fast path -- and the spin-poll is switched off for this engine (every later call would
        otherwise burn its full 0.25 s guard)."""
        arr = self._summary_np
        if arr[slot, 3] != 0:
            return
        vec = self._summary_vec[slot]
        if vec is None:
            return
        v = vec.detach().to("cpu").numpy()
        arr[slot, 0] = int(v.min()) if v.size else 0
        arr[slot, 1] = int((v == 2).sum()); arr[slot, 2] = int(((v & 3) != 0).sum()); arr[slot, 3] = 1
        if not getattr(self, "_summary_no_spin", False):
            self._summary_no_spin = True
            warnings.warn("MI355 solver: the status summary written by the device did not arrive in pinned host memory; recomputed from the status vector, "
                          "polling disabled for this engine (stream synchronisation from now on)")

    def read_summaries(self):
        """Waits for the summaries enqueued on this stream and returns them.  The last one enqueued carries a ready flag in pinned memory: polling it
        (a few hundred microseconds at most -- the stream holds one solve) spares the wake-up latency of a blocking stream synchronisation, which sits
        in the gap between the forward and the backward kernel of a training step.  CE_SPIN_WAIT=0: always synchronise the stream."""
        import os
        import time
        slot = getattr(self, "_summary_last_slot", None)
        if slot is not None and os.environ.get("CE_SPIN_WAIT") != "0" and not getattr(self, "_summary_no_spin", False):
            arr, t0 = self._summary_np, time.perf_counter()
            while arr[slot, 3] == 0:
                if time.perf_counter() - t0 > 0.25:                 # long solves: hand the core back
                    torch.cuda.current_stream(self.device).synchronize()
                    self.ensure_summary(slot)
                    break
        else:
            torch.cuda.current_stream(self.device).synchronize()
            if slot is not None:
                self.ensure_summary(slot)
        return self._summary_np.tolist()

    def status_summary(self, status: torch.Tensor) -> tuple[int, int]:
        """(min status, number of Solved/Inaccurate) of a status vector on this engine's device: one launch + a 12-byte pinned copy + one stream sync."""
        if status.numel() == 0:
            return 1, 0
        with torch.cuda.device(self.device):
            self.enqueue_summary(status, 0)
            r = self.read_summaries()
        return int(r[0][0]), int(r[0][1])

    def _note_acceleration(self, settings, honoured: bool, path: str):
        """Records whether this solve ran with Anderson acceleration (`last_acceleration`, surfaced as info["acceleration"]) and warns
        ONCE per engine when a positive acceleration_lookback -- explicit or the SCS default -- is not honoured by the selected path."""
        self.last_acceleration = bool(settings.acceleration_lookback > 0 and honoured)
        if settings.acceleration_lookback > 0 and not honoured and not getattr(self, "_aa_warned", False):
            self._aa_warned = True
            warnings.warn(f"MI355 solver: acceleration_lookback={settings.acceleration_lookback} is not implemented on the {path}; "
                          "iterating without Anderson acceleration (pass acceleration_lookback=0 to silence this)")

    def _use_const_a(self, A_bm) -> bool:
        """The batch-GEMM path pays off when the instance is too large for the register / LDS-resident kernels (those are
        faster for small instances even when A is shared).  CE_CONST_A=1 forces it whenever A is batch-invariant, =0 disables it."""
        import os
        from cvxpylayers_amd.interfaces.const_a import is_constant_A
        env = os.environ.get("CE_CONST_A")
        if env == "0" or A_bm.shape[0] < 2:
            return False
        if env != "1" and self.launch_info()["fwd_mode"] not in (1, 2):
            return False
        known = getattr(self, "A_is_constant", None)          # structural answer of MI355_ctx (parameter map), when the layer supplied it
        if known is not None:
            return bool(known)
        return is_constant_A(A_bm, self.nnzA)

    def vjp(self, A_bm, x, y, s, dx, dy, batch_minor_out: bool = False, P_bm=None, path: str | None = None, lsqr: tuple | None = None, q_eval=None):
        """Returns dA (nnz_aug, B), dq (n+1, B), adj_status (B,).  dA is a transposed view of a batch-major buffer (the
        engine-native layout, no extra pass) unless batch_minor_out: then it is (nnz_aug, B) contiguous -- the layout of a
        reference-style A_eval, so that autograd can accumulate it into the leaf without a strided copy (one engine layout pass).
        path: the path ("per_instance" / "const_a") of the forward call being differentiated, as recorded by the caller right
        after solve() -- the autograd node keeps it, so interleaved forward calls of one layer cannot redirect a pending backward.
        None (direct engine users with one solve in flight): the path of the most recent solve().
        lsqr: (atol, btol, iter_lim[, system]) of the shared-A LSQR adjoint (lsqr_rule); None = diffcp's 1e-8 / 1e-8 / 2 (n + m + 1) on the full system.  Ignored by the direct eliminations.
        q_eval: the forward call's (n+1, B) objective values; the shared-A LSQR adjoint then solves diffcp's full (n + m + 1) system (const_a.vjp_const_a)."""
        B = A_bm.shape[0]
        dev = self.device
        if B == 0:
            return (torch.empty((self.nnz_aug, 0), dtype=torch.float64, device=dev), torch.empty((self.n + 1, 0), dtype=torch.float64, device=dev),
                    torch.empty((0,), dtype=torch.int32, device=dev))
        if path is None:
            path = getattr(self, "last_path", None)
        if path == "const_a" and (self.launch_info()["bwd_mode"] in (1, 2) or __import__("os").environ.get("CE_CONST_A") == "1"):
            from cvxpylayers_amd.interfaces.const_a import vjp_const_a      # shared A: batched LSQR with GEMMs over the batch
            if q_eval is None:          # direct engine users: the objective of the most recent solve() of this very value buffer, else the reduced system
                q_eval = self._recent_q(A_bm)
            atol, btol, lim, system, method = unpack_rule(lsqr, self.n, self.m)
            if system == "reduced":
                q_eval = None
            _lib.check(_lib.lib().ce_set_lsqr_variant(self._h, 1 if method == "lsmr" else 0), "ce_set_lsqr_variant")
            try:
                return vjp_const_a(self, A_bm, x, y, s, dx, dy, batch_minor_out=batch_minor_out, atol=atol, btol=btol, iter_lim=lim, q_eval=q_eval)
            finally:
                _lib.lib().ce_set_lsqr_variant(self._h, 0)
        if path == "per_instance_lsqr":      # solver_args mode="lsqr" on a per-instance-A template: diffcp's LSQR instead of the direct elimination
            if P_bm is not None:
                raise ValueError("MI355 solver: mode='lsqr' is not available with a quadratic objective inside the kernels (CE_QP_EPIGRAPH=1 brings the problem to cone form)")
            atol, btol, lim, system, method = unpack_rule(lsqr, self.n, self.m)
            if q_eval is None and system != "reduced":
                q_eval = self._recent_q(A_bm)
            _lib.check(_lib.lib().ce_set_lsqr_variant(self._h, 1 if method == "lsmr" else 0), "ce_set_lsqr_variant")
            try:
                out = self._vjp_lsqr(A_bm, x, y, s, dx, dy, batch_minor_out, atol, btol, lim, None if system == "reduced" else q_eval)
            finally:
                _lib.lib().ce_set_lsqr_variant(self._h, 0)
            if out is not None:
                return out
            path = "per_instance"          # (the LSQR vectors of one instance exceed LDS: warned once, the direct elimination + re-solve serves the call)
        dq = torch.empty((self.n + 1, B), dtype=torch.float64, device=dev)
        adj = torch.empty((B,), dtype=torch.int32, device=dev)
        # rank-deficient instances are re-solved on the device by diffcp's LSQR when the call's q_eval is at hand (ce_vjp); path "per_instance_dense" (solver_args
        # mode="dense") keeps the elimination's basic solution
        q_args = (None, 0, 0)
        if P_bm is None and q_eval is not None and path != "per_instance_dense":          # (q_eval must be given explicitly here: without it, the elimination alone)
            qd = q_eval.detach().to(dtype=torch.float64, device=dev)
            q_args = (qd.data_ptr(), qd.stride(0), qd.stride(1))
            rule = unpack_rule(lsqr, self.n, self.m)[:4]
            if rule[:3] != getattr(self, "_resolve_rule", None):
                _lib.check(_lib.lib().ce_set_adjoint_resolve(self._h, 1, float(rule[0]), float(rule[1]), 1e8, int(rule[2])), "ce_set_adjoint_resolve")
                self._resolve_rule = rule[:3]
        if batch_minor_out:
            dA = torch.empty((self.nnz_aug, B), dtype=torch.float64, device=dev)
            sk, sb = B, 1
        else:
            dA = torch.empty((B, self.nnz_aug), dtype=torch.float64, device=dev)
            sk, sb = 1, self.nnz_aug
        if P_bm is not None:      # quadratic objective: also dP (B, nnz_p) batch-major
            dP = torch.empty((B, self.nnz_p), dtype=torch.float64, device=dev)
            rc = _lib.lib().ce_vjp_qp(self._h, B, A_bm.data_ptr(), 1, self.nnz_aug, P_bm.data_ptr(), x.data_ptr(), y.data_ptr(),
                                      s.data_ptr(), dx.data_ptr(), dy.data_ptr(), dA.data_ptr(), sk, sb,
                                      dq.data_ptr(), B, 1, dP.data_ptr(), adj.data_ptr(), self._stream())
            _lib.check(rc, "ce_vjp_qp")
            return (dA if batch_minor_out else dA.t()), dq, adj, dP
        rc = _lib.lib().ce_vjp(self._h, B, A_bm.data_ptr(), 1, self.nnz_aug, *q_args, x.data_ptr(), y.data_ptr(),
                               s.data_ptr(), dx.data_ptr(), dy.data_ptr(), dA.data_ptr(), sk, sb,
                               dq.data_ptr(), B, 1, adj.data_ptr(), self._stream())
        _lib.check(rc, "ce_vjp")
        return (dA if batch_minor_out else dA.t()), dq, adj

    def _vjp_lsqr(self, A_bm, x, y, s, dx, dy, batch_minor_out, atol, btol, iter_lim, q_eval, conlim=1e8):
        """ce_vjp_lsqr: one workgroup per instance runs Paige & Saunders' LSQR on diffcp's adjoint system M^T r = dz with THIS instance's A (include/cone_engine.h)"""
        dev, B = self.device, A_bm.shape[0]
        f64 = dict(dtype=torch.float64, device=dev)
        dA_bm = torch.empty((B, self.nnz_aug), **f64); dq = torch.empty((self.n + 1, B), **f64)
        adj = torch.empty((B,), dtype=torch.int32, device=dev); its = torch.empty((B,), dtype=torch.int32, device=dev)
        A_c = A_bm if (A_bm.stride(1) == 1 and (B == 1 or A_bm.stride(0) >= self.nnz_aug)) else A_bm.contiguous()
        xc, yc, sc_, dxc, dyc = (t.to(torch.float64).contiguous() for t in (x, y, s, dx, dy))
        if q_eval is not None:
            qd = q_eval.detach().to(**f64)
            q_args = (qd.data_ptr(), qd.stride(0), qd.stride(1))
        else:
            q_args = (None, 0, 0)
        rc = _lib.lib().ce_vjp_lsqr(self._h, B, A_c.data_ptr(), A_c.stride(0), *q_args, xc.data_ptr(), yc.data_ptr(), sc_.data_ptr(), dxc.data_ptr(), dyc.data_ptr(),
                                    dA_bm.data_ptr(), dq.data_ptr(), B, 1, adj.data_ptr(), its.data_ptr(), float(atol), float(btol), float(conlim), int(iter_lim), self._stream())
        if rc == -3:          # CE_E_TOO_LARGE: mode="lsqr" cannot be honoured for this template
            _warn_once("lsqr_too_large", "MI355 solver: solver_args mode='lsqr' needs the LSQR vectors of one instance in LDS, which this template exceeds; "
                                         "falling back to the direct elimination (rank-deficient instances are flagged in info['adjoint'])")
            return None
        _lib.check(rc, "ce_vjp_lsqr")
        self.last_lsqr_iters = its
        return (dA_bm.t().contiguous() if batch_minor_out else dA_bm.t()), dq, adj

    # introspection (bench / tests)
    def set_profiling(self, on):
        """False / True, or a sum of 2 (forward), 4 (adjoint), 8 (layout passes): which launches are bracketed by HIP events (include/cone_engine.h)"""
        _lib.lib().ce_set_profiling(self._h, int(on))

    def reset_profile(self):
        _lib.lib().ce_reset_profile(self._h)

    def profile(self, which: int):
        ms, nl = C.c_double(), C.c_int()
        _lib.check(_lib.lib().ce_get_profile(self._h, which, C.byref(ms), C.byref(nl)), "ce_get_profile")
        return ms.value, nl.value

    def launch_info(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.lib().ce_get_launch_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(fwd_lds_bytes=a.value, bwd_lds_bytes=b.value, fwd_mode=c.value, bwd_mode=d.value)



class QuadEpigraph:
    """min 1/2 x^T P x + q^T x + d  s.t.  A x + s = b, s in K      ==>      min t + q^T x + d  s.t. (same), and
           (t + 1, sqrt(2) L^T x, t - 1) in SOC(n + 2),   P = L L^T,
    because ||(sqrt(2) L^T x, t - 1)|| <= t + 1  <=>  1/2 x^T P x <= t.  This is the reduction CVXPY itself applies when a solver has
    no quadratic objective (what DIFFCP receives); doing it here, on device tensors under autograd, lets P be a *parameter*: the
    Cholesky factor is computed per instance (batched, differentiable), its entries become entries of A_eval, and the gradient
    with respect to P flows back through torch's Cholesky derivative.  One extra variable (t, last) and one extra SOC block,
    placed after the template's own SOC blocks (SCS row order z, l, q, s, ep, p); the template's rows keep their relative order.

    P_eval holds the values of P in the CSC structure `objective_structure = (indices, indptr, (n, n))`; a structure with all
    entries on or above (or on or below) the diagonal is read as one triangle of the symmetric matrix."""

    def __init__(self, objective_structure, A_structure, A_shape, cone_dict):
        p_indices, p_indptr, (n, n2) = objective_structure
        m, np1 = A_shape
        assert n == n2 == np1 - 1, "P must be n x n"
        self.n, self.m = int(n), int(m)
        self.p_rows = np.asarray(p_indices, dtype=np.int64)
        self.p_cols = np.repeat(np.arange(n), np.diff(np.asarray(p_indptr))).astype(np.int64)
        self.one_triangle = bool(len(self.p_rows)) and (bool((self.p_rows <= self.p_cols).all()) or bool((self.p_rows >= self.p_cols).all()))
        self.p_indices, self.p_indptr = np.asarray(p_indices, dtype=np.int32), np.asarray(p_indptr, dtype=np.int32)
        # native (in-kernel) P needs symmetric values: for a full structure, entry (i, j) is averaged with entry (j, i); sym_perm
        # is that pairing (identity for one-triangle structures, None when the structure is not symmetric -> epigraph form only)
        if self.one_triangle or len(self.p_rows) == 0:
            self.sym_perm = np.arange(len(self.p_rows))
        else:
            pos = {(int(r), int(c)): k for k, (r, c) in enumerate(zip(self.p_rows, self.p_cols))}
            perm = [pos.get((int(c), int(r)), -1) for r, c in zip(self.p_rows, self.p_cols)]
            self.sym_perm = np.asarray(perm) if min(perm) >= 0 else None
        a_idx, a_ptr = np.asarray(A_structure[0], dtype=np.int64), np.asarray(A_structure[1], dtype=np.int64)
        nnz_old = int(a_ptr[-1])
        r0 = int(cone_dict.get("z", 0)) + int(cone_dict.get("l", 0)) + int(sum(cone_dict.get("q", [])))      # first row of the new SOC block
        self.r0 = r0
        d = n + 2
        self.m_aug = m + d
        remap = np.where(np.arange(m) < r0, np.arange(m), np.arange(m) + d)       # template row -> augmented row
        self.dual_rows = remap
        # entries of the augmented [A_cvx | b_cvx] (columns x_0..x_{n-1}, t, b), sorted by (column, row); source index into
        # cat([A_eval (nnz_old), sqrt(2) * L[j, k] for the n(n+1)/2 pairs j >= k, +1, -1]) per instance
        tri_j, tri_k = np.tril_indices(n)
        self.tri_j, self.tri_k = tri_j, tri_k
        ntri = len(tri_j)
        ONE, MINUS = nnz_old + ntri, nnz_old + ntri + 1
        a_cols = np.repeat(np.arange(np1), np.diff(a_ptr))
        ent = []          # (col, row, source)
        for kk in range(nnz_old):
            c_ = int(a_cols[kk])
            ent.append((c_ if c_ < n else n + 1, int(remap[a_idx[kk]]), kk))
        for e in range(ntri):          # A_cvx[r0 + 1 + k, j] = sqrt(2) L[j, k]
            ent.append((int(tri_j[e]), r0 + 1 + int(tri_k[e]), nnz_old + e))
        ent.append((n, r0, ONE)); ent.append((n, r0 + n + 1, ONE))               # t in the first and the last row of the block
        ent.append((n + 1, r0, ONE)); ent.append((n + 1, r0 + n + 1, MINUS))     # b = (1, 0, ..., 0, -1)
        ent.sort()
        self.aug_indices = np.asarray([e[1] for e in ent], dtype=np.int32)
        counts = np.bincount(np.asarray([e[0] for e in ent]), minlength=n + 2)
        self.aug_indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.src = np.asarray([e[2] for e in ent], dtype=np.int64)
        self.aug_cones = {**cone_dict, "q": list(cone_dict.get("q", [])) + [d]}
        self._dev = {}

    def _idx(self, device):
        key = str(device)
        if key not in self._dev:
            t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int64)).to(device)
            self._dev[key] = dict(src=t(self.src), pr=t(self.p_rows), pc=t(self.p_cols), tj=t(self.tri_j), tk=t(self.tri_k), dual=t(self.dual_rows))
        return self._dev[key]

    def assemble(self, P_eval, q_eval, A_eval):
        """(nnz_P, B), (n+1, B), (nnz_aug, B)  ->  q_aug (n+2, B), A_aug (nnz_aug', B); differentiable torch ops"""
        ix = self._idx(A_eval.device)
        n, B = self.n, A_eval.shape[1]
        f64 = dict(dtype=torch.float64, device=A_eval.device)
        Pd = torch.zeros((B, n * n), **f64).index_add(1, ix["pr"] * n + ix["pc"], P_eval.to(torch.float64).t()).reshape(B, n, n)
        if self.one_triangle:
            Pd = Pd + Pd.transpose(1, 2) - torch.diag_embed(torch.diagonal(Pd, dim1=1, dim2=2))
        else:
            Pd = 0.5 * (Pd + Pd.transpose(1, 2))
        # P is positive semidefinite, possibly singular: a relative jitter keeps the factorisation defined (1e-12 of the largest
        # diagonal entry: far below the solver tolerance)
        jit = 1e-12 * torch.diagonal(Pd, dim1=1, dim2=2).abs().amax(dim=1).clamp_min(1e-300) + 1e-300
        Lf, info = torch.linalg.cholesky_ex(Pd + jit[:, None, None] * torch.eye(n, **f64))
        if bool((info != 0).any()):
            raise SolverError("MI355 solver: the quadratic objective matrix P is not positive semidefinite "
                              f"(Cholesky failed for {int((info != 0).sum())} of {B} instances)")
        Lvals = (2.0 ** 0.5) * Lf[:, ix["tj"], ix["tk"]].t()                      # (n(n+1)/2, B)
        one = torch.ones((1, B), **f64)
        source = torch.cat([A_eval.to(torch.float64), Lvals, one, -one], dim=0)
        A_aug = source.index_select(0, ix["src"])
        q64 = q_eval.to(torch.float64)
        q_aug = torch.cat([q64[:n], one, q64[n:n + 1]], dim=0)
        return q_aug, A_aug

    def split(self, primal_aug, dual_aug):
        ix = self._idx(dual_aug.device)
        return primal_aug[:, :self.n], dual_aug.index_select(1, ix["dual"])


class MI355_ctx:
    """Built once per layer from CVXPY's ParamConeProg; same constructor signature as DIFFCP_ctx
    (diffcp_if.py:105-120): constraint_structure = (indices, indptr, (m, n+1)) is the CSC structure of the
    augmented matrix [A_cvx | b_cvx]."""

    def __init__(self, objective_structure, constraint_structure, dims, lower_bounds=None, upper_bounds=None, options=None, reduced_A_mat=None):
        """reduced_A_mat (optional; the keyword MOREAU_ctx takes, moreau_if.py:159-256): the parameter map of the constraint values, (nnz_aug, P + 1) with the
        constant column last.  When given, whether the A part is batch-invariant is decided HERE, once, from the map's structure -- rows of A entries with no
        parameter column (the reference's PA_is_constant test, moreau_if.py:234-256, restricted to the A rows: b may vary) -- and no call compares values."""
        con_indices, con_ptr, (m, np1) = constraint_structure
        self.A_structure = (np.asarray(con_indices), np.asarray(con_ptr))
        self.A_shape = (int(m), int(np1))
        self.b_idx = np.asarray(con_indices)[con_ptr[-2]:con_ptr[-1]]
        self.dims = dims
        self.cone_dict = dims_to_solver_dict(dims)
        self.options = options or {}
        self.default_device = torch.device("cuda", 0)
        self._engines: dict[int, ConeEngine] = {}
        self.A_is_constant = None          # None: unknown (decided per call from the values, one compare + one host sync); True / False: structural
        if reduced_A_mat is not None:
            nnzA = int(np.asarray(con_ptr)[int(np1) - 1])
            self.A_is_constant = bool(reduced_A_mat[:nnzA, :-1].nnz == 0) if nnzA > 0 else True
        # Quadratic objective 1/2 x^T P x (plugins registered in SUPPORTS_QUAD_OBJ receive P_eval, _quad_form_dpp.py:32,
        # interfaces/__init__.py:35-42): handled as an epigraph SOC block over the Cholesky factor of P (see QuadEpigraph).
        self.quad = QuadEpigraph(objective_structure, self.A_structure, self.A_shape, self.cone_dict) if objective_structure is not None else None
        self._aug_ctx = None

    def augmented(self) -> "MI355_ctx":
        """the cone-program context of the epigraph form (one extra variable, one extra SOC of dimension n + 2)"""
        if self._aug_ctx is None:
            q = self.quad
            self._aug_ctx = MI355_ctx(None, (q.aug_indices, q.aug_indptr, (q.m_aug, q.n + 2)), q.aug_cones, None, None, self.options)
            self._aug_ctx.A_is_constant = False if self.A_is_constant is False else None      # (the Cholesky factor of P enters A: constant only if P is, which the values decide)
            self._aug_ctx.default_device = self.default_device
        return self._aug_ctx

    def engine(self, device: torch.device) -> ConeEngine:
        idx = device.index or 0
        if idx not in self._engines:
            pst = (self.quad.p_indices, self.quad.p_indptr) if self.quad is not None and self.quad.sym_perm is not None else None
            self._engines[idx] = ConeEngine(self.A_structure[0], self.A_structure[1], self.A_shape[1] - 1, self.A_shape[0],
                                            self.cone_dict, torch.device("cuda", idx), p_structure=pst)
            # A layer is called again and again on related batches (training steps, sweeps): dispatch the instances that ran longest last time first
            # (options={"dispatch_history": False} or CE_DISPATCH_HISTORY=0 switch it off; see include/cone_engine.h)
            import os
            self._engines[idx].A_is_constant = self.A_is_constant
            self._engines[idx].set_dispatch_history(bool(self.options.get("dispatch_history", True)) and os.environ.get("CE_DISPATCH_HISTORY") != "0")
        return self._engines[idx]


def _note_adjoint_flags(eng, adj, bs):
    """Called at the END of a backward call: the per-instance flags of the adjoint (more active rows than the direct solve holds, LSQR iteration limit: such
    gradients are zero or inexact) are summarised on the device behind the adjoint kernel -- no host synchronisation on the backward path -- into one of two
    alternating pinned slots.  The NEXT forward call reports them (everything enqueued before its status summary is complete when that is read).  A layer
    applied several times in one graph (the 20 time steps of the supply-chain loop) runs several backward calls between two forwards: the slot a new call is
    about to reuse is folded into the running count first."""
    if adj.numel() == 0:
        return
    pend = getattr(eng, "_adj_pending", None)
    if pend is None:
        pend = eng._adj_pending = []
        eng._adj_seq, eng._adj_count, eng._adj_total = 0, 0, 0
    slot = 1 + (eng._adj_seq & 1)
    eng._adj_seq += 1
    for k in [k for k, (sl, _) in enumerate(pend) if sl == slot]:          # written two backward calls ago: long complete, but make sure before it is overwritten
        if getattr(eng, "_summary_np", None) is not None and eng._summary_np[slot, 3] == 0:
            if getattr(eng, "_async_mode", False):      # asynchronous forward (raise_on_error=False): the host runs ahead of the device and must never wait -- this call's flags go unreported
                eng._adj_seq -= 1
                return
            torch.cuda.current_stream(eng.device).synchronize()
        _fold_adjoint(eng, pend.pop(k))
    eng.enqueue_summary(adj, slot)
    pend.append((slot, bs))


def _fold_adjoint(eng, entry):
    slot, bs = entry
    # the slot's ready flag, not stream order, says that its three counts have landed: a backward that ran on another stream than the forward whose summary
    # was just read is not ordered by that read
    if getattr(eng, "_summary_np", None) is not None and eng._summary_np[slot, 3] == 0:
        torch.cuda.synchronize(eng.device)
        eng.ensure_summary(slot)
    eng._adj_count += int(eng._summary_np[slot, 2])     # bits 0-1; bit 2 (4) = rank-deficient system, basic solution returned like the reference's LSQR does -- not a failure
    eng._adj_total += bs


def _report_previous_async(eng):
    """raise_on_error=False: the outcome of the PREVIOUS asynchronous forward, if its summary has landed in the pinned slot (never waits)"""
    bs = getattr(eng, "_async_pending", None)
    arr = getattr(eng, "_summary_np", None)
    if not bs or arr is None or arr[0, 3] == 0:
        return
    eng._async_pending = None
    mn, n_inacc = int(arr[0, 0]), int(arr[0, 1])
    if mn < 0:
        warnings.warn(f"MI355 solver: instances of the previous forward call (batch of {bs}) failed (worst status {STATUS_NAMES.get(mn, mn)}); their rows were returned as NaN "
                      "with zero gradients (raise_on_error=False); info['status'] of that call says which")
    if n_inacc:
        warnings.warn("Solved/Inaccurate.")


def _report_flagged_adjoints(eng, block: bool = True):
    """forward side: every backward enqueued before this forward's status summary has finished by the time that summary is read
    (block=False -- the asynchronous forward of raise_on_error=False: only the slots whose ready flag is already set are folded)"""
    pend = getattr(eng, "_adj_pending", None)
    if not pend and not getattr(eng, "_adj_total", 0):
        return
    if not block:
        arr = getattr(eng, "_summary_np", None)
        ready = [k for k, (sl, _) in enumerate(pend or []) if arr is not None and arr[sl, 3] != 0]
        for k in reversed(ready):
            _fold_adjoint(eng, pend.pop(k))
        if pend:
            return
    while pend:
        _fold_adjoint(eng, pend.pop())
    nbad, bs = eng._adj_count, eng._adj_total
    eng._adj_count = eng._adj_total = 0
    if nbad:
        warnings.warn(f"MI355 adjoint: {nbad} of {bs} instances of the previous backward pass were flagged (degenerate active "
                      "set or iteration limit); their gradients are unreliable")


def adjoint_report(info: dict) -> dict:
    """Counts of the last backward through the node that returned `info` (synchronises): instances whose adjoint system was rank deficient, how many of them
    were re-solved by diffcp's LSQR on the device, how many LSQR runs stopped at the iteration limit / were left without gradient."""
    adj = (info.get("adjoint") or {}).get("status")
    if adj is None:
        return dict(rank_deficient=0, lsqr_resolved=0, lsqr_iteration_limit=0, no_gradient=0, backward_ran=False)
    a = adj.detach().cpu().numpy()
    return dict(rank_deficient=int(((a & 4) != 0).sum()), lsqr_resolved=int(((a & 8) != 0).sum()), lsqr_iteration_limit=int(((a & 1) != 0).sum()),
                no_gradient=int(((a & 2) != 0).sum()), backward_ran=True)


def _detect_batch_size(con_values) -> tuple[int, bool]:
    """diffcp_if.py:34-43"""
    if con_values.dim() == 1:
        return 1, True
    return con_values.shape[1], False


class _ConeLayer(torch.autograd.Function):
    """Same calling convention as diffcp_if._CvxpyLayer (diffcp_if.py:327-403); linear objective (P_eval is None)."""

    @staticmethod
    def forward(P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad=True, warm_start=None):
        ctx = cl_ctx.solver_ctx if hasattr(cl_ctx, "solver_ctx") else cl_ctx
        # P_eval given: only for engines that run the quadratic objective inside the kernels (_CvxpyLayer.apply decides)
        batch_size, originally_unbatched = _detect_batch_size(A_eval)
        if originally_unbatched:
            A_eval = A_eval.unsqueeze(1)
            q_eval = q_eval.unsqueeze(1)
        in_device = A_eval.device
        dev = in_device if in_device.type == "cuda" else ctx.default_device
        if not torch.cuda.is_available():
            raise RuntimeError("MI355 solver needs a ROCm GPU; there is no CPU fallback (use solver='DIFFCP' on CPU)")
        eng = ctx.engine(dev)
        merged_args = {**ctx.options}
        if solver_args:
            merged_args.update(solver_args)
        # SCS runs with Anderson acceleration by default (acceleration_lookback = 10, acceleration_interval = 10) and diffcp forwards
        # SCS's defaults (diffcp_if.py:356-367); ce_default_settings carries the same defaults, so identical solver_args mean the same
        # algorithm here, at the C ABI and in the reference.  The engine keeps a one-pair history whatever the lookback (iteration
        # counts within 2.5 % of lookback 10 on every BASELINE configuration, profiles/r02/aa_memory.json).
        settings = make_settings(merged_args)
        note_ignored_args({"acceleration_lookback": settings.acceleration_lookback, **{k: merged_args[k] for k in ("mode", "solve_method", "n_jobs_forward", "n_jobs_backward") if k in merged_args}},
                          explicit_lookback="acceleration_lookback" in merged_args)
        if warm_start is None and merged_args.get("warm_starts") is not None:
            # diffcp's solve argument (diffcp_if.py:365-367 forwards it): one (x, y, s) triple per instance
            ws = merged_args["warm_starts"]
            if len(ws) != batch_size:
                raise ValueError(f"warm_starts: expected one (x, y, s) triple per instance ({batch_size}), got {len(ws)}")
            warm_start = tuple(torch.stack([torch.as_tensor(np.asarray(t[k]), dtype=torch.float64) for t in ws]) for k in range(3))
        with torch.cuda.device(dev):
            A_dev = A_eval.detach().to(device=dev, dtype=torch.float64)
            q_dev = q_eval.detach().to(device=dev, dtype=torch.float64)
            batch_minor_in = A_dev.dim() == 2 and A_dev.is_contiguous() and A_dev.shape[1] > 1
            A_bm = eng.to_batch_major(A_dev)
            P_bm = None
            if P_eval is not None:
                if originally_unbatched:
                    P_eval = P_eval.unsqueeze(1)
                P_bm = P_eval.detach().to(device=dev, dtype=torch.float64).t().contiguous()        # (B, nnz_p)
            warm = None
            if warm_start is True:                                   # re-use the previous solution of this layer (same batch size)
                prev = getattr(eng, "_last_solution", None)
                if prev is not None and prev[0].shape[0] == A_bm.shape[0]:
                    warm = prev
            elif warm_start not in (None, False):
                warm = tuple(t if t.dim() == 2 else t.unsqueeze(0) for t in warm_start)     # (x, y, s) tensors
            x, y, s, iters, status, resid = eng.solve(A_bm, q_dev, settings, warm=warm, P_bm=P_bm)
            path = eng.last_path          # recorded per call: the backward of THIS node must not follow a later solve's path
            if path == "per_instance" and P_bm is None and adjoint_mode(merged_args) == "lsqr":
                path = "per_instance_lsqr"          # (the adjoint of this node: diffcp's LSQR instead of the direct elimination)
            elif path == "per_instance" and P_bm is not None and adjoint_mode(merged_args) == "lsqr":
                _warn_once("lsqr_qp", "MI355 solver: solver_args mode='lsqr' is not available with a quadratic objective inside the kernels; the direct elimination "
                                      "differentiates this layer (CE_QP_EPIGRAPH=1 brings the problem to cone form, where mode='lsqr' applies)")
            elif path == "per_instance" and adjoint_mode(merged_args) == "dense":
                path = "per_instance_dense"         # (the elimination alone: no LSQR re-solve of rank-deficient instances)
            eng._last_solution = (x.detach(), y.detach(), s)
            # The reference raises from forward() when an instance fails (diffcp_if.py:365-372), so the host has to learn the outcome here: one tiny
            # reduction kernel + 8 bytes into pinned memory behind the solve (ce_status_summary) and ONE stream synchronisation -- not the status
            # vector through a pageable copy plus host-side reductions.  Per-instance inspection happens only on the failure path.
            raise_on = bool(merged_args.get("raise_on_error", True))
            eng._async_mode = not raise_on
            if not raise_on:
                _report_previous_async(eng)          # (what the previous asynchronous forward left in the pinned slot, if it has landed: warnings only, never a wait)
            if status.numel():
                eng.enqueue_summary(status, 0)
            # everything the host can prepare without knowing the outcome happens BEFORE the one synchronisation of this call: the GPU is idle from the
            # end of the solve until the caller's backward reaches it, so host work placed behind the wait is added to that gap
            primal = x.to(in_device)
            dual = y.to(in_device)
            # info["adjoint"] is filled by backward(): "status" = the per-instance bit field of include/cone_engine.h ce_vjp (4: rank-deficient system, 8: gradients
            # are diffcp's LSQR element from the device-side re-solve); adjoint_report(info) counts them
            info = dict(iters=iters, status=status, resid=resid, acceleration=getattr(eng, "last_acceleration", False), adjoint={"status": None, "path": path})
            lsqr = lsqr_rule(merged_args, eng.n, eng.m)
            saved = (eng, A_bm, x.detach(), y.detach(), s, batch_minor_in, P_bm, path, None, lsqr, q_dev if P_bm is None else None) if needs_grad else None
            if status.numel() and not raise_on:
                # raise_on_error=False: the caller has waived the reference's "raise from forward()" contract, so NOTHING forces a host round trip here.  Failed
                # instances are masked ON THE DEVICE (two small elementwise launches, unconditionally), the outcome summary lands in pinned memory behind the
                # solve and is reported -- as warnings -- by the next call that finds it there.  The GPU never waits for the host between the forward and the
                # adjoint kernel (bench.py `async_forward`: the host gap of the step disappears).
                failed = (status < 0)
                nanv = float("nan")
                x = torch.where(failed[:, None], nanv, x); y = torch.where(failed[:, None], nanv, y)
                eng._async_pending = batch_size
                saved = (eng, A_bm, x.detach(), y.detach(), s, batch_minor_in, P_bm, path, failed, lsqr, q_dev if P_bm is None else None) if needs_grad else None
                _report_flagged_adjoints(eng, block=False)
                return x.to(in_device), y.to(in_device), info, (saved, batch_size, originally_unbatched, in_device)
            if status.numel():
                summ = eng.read_summaries()
                min_status, n_inaccurate = int(summ[0][0]), int(summ[0][1])
                _report_flagged_adjoints(eng)          # (the flags of the backward calls since the last forward: summarised behind their kernels, complete by now)
            else:
                min_status, n_inaccurate = 1, 0
        any_failed = min_status < 0
        if any_failed and raise_on:
            st = status.cpu()
            bad = int((st < 0).nonzero()[0])
            raise SolverError(f"Solver mi355 returned status {STATUS_NAMES.get(int(st[bad]), int(st[bad]))} "
                              f"for instance {bad} ({int((st < 0).sum())} of {batch_size} instances failed)")
        if n_inaccurate:
            warnings.warn("Solved/Inaccurate.")
        # (raise_on_error=False never reaches this point: its failure masking happens on the device, above)
        # x / y are handed back as `primal` / `dual` (same objects when the input lives on the engine's device), and autograd
        # attaches this node to them: keeping the SAME objects on the node would form a reference cycle (node -> saved ->
        # primal -> grad_fn -> node) that only the cyclic GC breaks, i.e. 167 MB buffers pile up for many steps and the
        # caching allocator falls back to hipMalloc (3 ms each).  Detached aliases share the storage without the cycle.
        return primal, dual, info, (saved, batch_size, originally_unbatched, in_device)

    @staticmethod
    def setup_context(ctx, inputs, outputs):
        _, _, info, backward_data = outputs
        ctx.info = info
        ctx.backward_data = backward_data
        ctx.set_materialize_grads(False)      # an unused output (the duals, most of the time) arrives as None instead of a freshly filled zero tensor: one launch less per step

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dprimal, ddual, _info, _data):
        saved, batch_size, originally_unbatched, in_device = ctx.backward_data
        if saved is None:
            raise RuntimeError("backward called on a layer evaluated with needs_grad=False")
        eng, A_bm, x, y, s, batch_minor_in, P_bm, path, failed, lsqr, q_saved = saved
        dP = None
        if dprimal is None and ddual is None:         # nothing flows back through this node
            return None, None, None, None, None, None, None
        with torch.cuda.device(eng.device):
            dx = dprimal.to(device=eng.device, dtype=torch.float64).contiguous() if dprimal is not None else eng.zeros_like_cached(x)
            dy = ddual.to(device=eng.device, dtype=torch.float64).contiguous() if ddual is not None else eng.zeros_like_cached(y)
            if failed is not None:          # masked instances: NaN outputs upstream produce NaN cotangents; they contribute nothing
                keep = ~failed[:, None]
                dx = torch.where(keep, dx, torch.zeros_like(dx)); dy = torch.where(keep, dy, torch.zeros_like(dy))
                x = torch.where(keep, x, torch.zeros_like(x)); y = torch.where(keep, y, torch.zeros_like(y)); s = torch.where(keep, s, torch.zeros_like(s))
            if P_bm is not None:
                dA, dq, adj, dP_bm = eng.vjp(A_bm, x, y, s, dx, dy, batch_minor_out=batch_minor_in, P_bm=P_bm, path=path, lsqr=lsqr)
                dP = dP_bm.t().to(in_device)
            else:
                dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy, batch_minor_out=batch_minor_in, path=path, lsqr=lsqr, q_eval=q_saved)
        # (masked instances: zero cotangents at a zero point give exactly zero dA / dq / dP rows from every adjoint kernel -- r = 0 --, no pass over the gradients needed)
        ctx.adj_status = adj
        if isinstance(ctx.info, dict) and isinstance(ctx.info.get("adjoint"), dict):
            ctx.info["adjoint"]["status"] = adj
        with torch.cuda.device(eng.device):
            _note_adjoint_flags(eng, adj, batch_size)            # reported by the next forward call (no host sync on the backward path)
        dA = dA.to(in_device)
        dq = dq.to(in_device)
        if originally_unbatched:
            dq = dq.squeeze(1)
            dA = dA.squeeze(1)
            dP = dP.squeeze(1) if dP is not None else None
        return dP, dq, dA, None, None, None, None


class _CvxpyLayer:
    """What get_torch_cvxpylayer("MI355") returns: `apply(P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad, warm_start)
    -> (primal, dual, aux, data)` like every reference plugin (torch/cvxpylayer.py:475-483).  A linear objective goes straight to
    the autograd Function; a quadratic objective (P_eval given, the ctx built with an objective structure) is first brought to
    epigraph cone form by differentiable torch ops (QuadEpigraph), so gradients reach P_eval through autograd."""

    @staticmethod
    def apply(P_eval, q_eval, A_eval, cl_ctx, solver_args=None, needs_grad=True, warm_start=None):
        if P_eval is None:
            return _ConeLayer.apply(None, q_eval, A_eval, cl_ctx, solver_args, needs_grad, warm_start)
        ctx = cl_ctx.solver_ctx if hasattr(cl_ctx, "solver_ctx") else cl_ctx
        if ctx.quad is None:
            raise ValueError("MI355 solver: P_eval was given but the context was built without an objective structure")
        import os
        dev0 = A_eval.device if A_eval.device.type == "cuda" else ctx.default_device
        if ctx.quad.sym_perm is not None and os.environ.get("CE_QP_EPIGRAPH") != "1" and ctx.engine(dev0).qp_native:
            # P inside the kernels (SCS 3's QP embedding; plain cones, register-tiled sizes): symmetric values in, dP out
            if not ctx.quad.one_triangle:
                perm = torch.from_numpy(ctx.quad.sym_perm).to(P_eval.device)
                P_eval = 0.5 * (P_eval + P_eval.index_select(0, perm))
            return _ConeLayer.apply(P_eval, q_eval, A_eval, ctx, solver_args, needs_grad, warm_start)
        unbatched = A_eval.dim() == 1
        if unbatched:
            P_eval, q_eval, A_eval = P_eval.unsqueeze(1), q_eval.unsqueeze(1), A_eval.unsqueeze(1)
        if A_eval.device.type != "cuda":
            P_eval, q_eval, A_eval = (t.to(ctx.default_device) for t in (P_eval, q_eval, A_eval))
        q_aug, A_aug = ctx.quad.assemble(P_eval, q_eval, A_eval)
        primal_a, dual_a, info, data = _ConeLayer.apply(None, q_aug, A_aug, ctx.augmented(), solver_args, needs_grad, warm_start)
        primal, dual = ctx.quad.split(primal_a, dual_a)
        return primal, dual, info, data
