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
               "n_jobs_forward", "n_jobs_backward", "warm_starts", "raise_on_error"}


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
    if a.get("acceleration_lookback", 0) not in (0, None):
        # Anderson acceleration is not implemented on the device path; the iteration converges to the same
        # optimum without it (README.md:233-236 recommends acceleration_lookback=0 for robustness).
        pass
    return s


class ConeEngine:
    """Owns one ce_handle (one template, one device)."""

    def __init__(self, indices, indptr, n, m, cone_dict, device: torch.device):
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
        h = C.c_void_p()
        rc = L.ce_create(C.byref(t), device.index or 0, C.byref(h))
        if rc == -2:
            raise NotImplementedError(L.ce_last_error().decode())
        _lib.check(rc, "ce_create")
        self._h = h

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
        if A_eval.stride(0) == 1 and (A_eval.stride(1) == K or B == 1):
            return A_eval.t()
        if not A_eval.is_contiguous():
            A_eval = A_eval.contiguous()
        out = torch.empty((B, K), dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().ce_transpose(self._h, K, B, A_eval.data_ptr(), out.data_ptr(), self._stream()), "ce_transpose")
        return out

    def solve(self, A_bm: torch.Tensor, q_eval: torch.Tensor, settings, warm=None):
        """A_bm (B, nnz_aug) contiguous, q_eval (n+1, B) any strides.  Returns x, y, s, iters, status, resid.
        warm = (x, y, s) of shapes (B, n), (B, m), (B, m): initial point (instances with non-finite entries start cold)."""
        B = A_bm.shape[0]
        dev = self.device
        if warm is not None:
            if tuple(warm[0].shape) != (B, self.n) or tuple(warm[1].shape) != (B, self.m) or tuple(warm[2].shape) != (B, self.m):
                raise ValueError(f"warm start: expected x {(B, self.n)}, y {(B, self.m)}, s {(B, self.m)}, got "
                                 f"{tuple(warm[0].shape)}, {tuple(warm[1].shape)}, {tuple(warm[2].shape)}")
        if self._use_const_a(A_bm):
            from cvxpylayers_amd.interfaces.const_a import solve_const_a
            self.last_path = "const_a"
            return solve_const_a(self, A_bm, q_eval, settings, warm=warm)
        self.last_path = "per_instance"
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
        rc = _lib.lib().ce_solve(self._h, B, A_bm.data_ptr(), 1, self.nnz_aug, q_eval.data_ptr(), q_eval.stride(0),
                                 q_eval.stride(1), C.byref(settings), x.data_ptr(), y.data_ptr(), s.data_ptr(),
                                 iters.data_ptr(), status.data_ptr(), resid.data_ptr(), self._stream())
        _lib.check(rc, "ce_solve")
        return x, y, s, iters, status, resid

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
        return is_constant_A(A_bm, self.nnzA)

    def vjp(self, A_bm, x, y, s, dx, dy, batch_minor_out: bool = False):
        """Returns dA (nnz_aug, B), dq (n+1, B), adj_status (B,).  dA is a transposed view of a batch-major buffer (the
        engine-native layout, no extra pass) unless batch_minor_out: then it is (nnz_aug, B) contiguous -- the layout of a
        reference-style A_eval, so that autograd can accumulate it into the leaf without a strided copy (one engine layout pass)."""
        B = A_bm.shape[0]
        dev = self.device
        if getattr(self, "last_path", None) == "const_a" and (self.launch_info()["bwd_mode"] in (1, 2) or __import__("os").environ.get("CE_CONST_A") == "1"):
            from cvxpylayers_amd.interfaces.const_a import vjp_const_a      # shared A: batched LSQR with GEMMs over the batch
            return vjp_const_a(self, A_bm, x, y, s, dx, dy, batch_minor_out=batch_minor_out)
        dq = torch.empty((self.n + 1, B), dtype=torch.float64, device=dev)
        adj = torch.empty((B,), dtype=torch.int32, device=dev)
        if batch_minor_out:
            dA = torch.empty((self.nnz_aug, B), dtype=torch.float64, device=dev)
            sk, sb = B, 1
        else:
            dA = torch.empty((B, self.nnz_aug), dtype=torch.float64, device=dev)
            sk, sb = 1, self.nnz_aug
        rc = _lib.lib().ce_vjp(self._h, B, A_bm.data_ptr(), 1, self.nnz_aug, None, 0, 0, x.data_ptr(), y.data_ptr(),
                               s.data_ptr(), dx.data_ptr(), dy.data_ptr(), dA.data_ptr(), sk, sb,
                               dq.data_ptr(), B, 1, adj.data_ptr(), self._stream())
        _lib.check(rc, "ce_vjp")
        return (dA if batch_minor_out else dA.t()), dq, adj

    # introspection (bench / tests)
    def set_profiling(self, on: bool):
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


class MI355_ctx:
    """Built once per layer from CVXPY's ParamConeProg; same constructor signature as DIFFCP_ctx
    (diffcp_if.py:105-120): constraint_structure = (indices, indptr, (m, n+1)) is the CSC structure of the
    augmented matrix [A_cvx | b_cvx]."""

    def __init__(self, objective_structure, constraint_structure, dims, lower_bounds=None, upper_bounds=None, options=None):
        con_indices, con_ptr, (m, np1) = constraint_structure
        self.A_structure = (np.asarray(con_indices), np.asarray(con_ptr))
        self.A_shape = (int(m), int(np1))
        self.b_idx = np.asarray(con_indices)[con_ptr[-2]:con_ptr[-1]]
        self.dims = dims
        self.cone_dict = dims_to_solver_dict(dims)
        self.options = options or {}
        self.default_device = torch.device("cuda", 0)
        self._engines: dict[int, ConeEngine] = {}

    def engine(self, device: torch.device) -> ConeEngine:
        idx = device.index or 0
        if idx not in self._engines:
            self._engines[idx] = ConeEngine(self.A_structure[0], self.A_structure[1], self.A_shape[1] - 1, self.A_shape[0],
                                            self.cone_dict, torch.device("cuda", idx))
        return self._engines[idx]


def _warn_flagged_adjoint(eng):
    """Deferred check of the previous backward's per-instance flags (degenerate active set: more active rows than the direct
    solve holds / singular pivot, or LSQR iteration limit): their gradients are zero or inexact.  Done at the next call so that
    the backward path itself never synchronises the host."""
    pend = getattr(eng, "_pending_adj", None)
    if pend is not None:
        eng._pending_adj = None
        adj, bs = pend
        nbad = int((adj != 0).sum())
        if nbad:
            warnings.warn(f"MI355 adjoint: {nbad} of {bs} instances of the previous backward pass were flagged (degenerate active "
                          "set or iteration limit); their gradients are unreliable")


def _detect_batch_size(con_values) -> tuple[int, bool]:
    """diffcp_if.py:34-43"""
    if con_values.dim() == 1:
        return 1, True
    return con_values.shape[1], False


class _CvxpyLayer(torch.autograd.Function):
    """Same calling convention as diffcp_if._CvxpyLayer (diffcp_if.py:327-403)."""

    @staticmethod
    def forward(P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad=True, warm_start=None):
        ctx = cl_ctx.solver_ctx if hasattr(cl_ctx, "solver_ctx") else cl_ctx
        if P_eval is not None:
            raise NotImplementedError("MI355 solver: quadratic objectives (P) are not supported yet")
        batch_size, originally_unbatched = _detect_batch_size(A_eval)
        if originally_unbatched:
            A_eval = A_eval.unsqueeze(1)
            q_eval = q_eval.unsqueeze(1)
        in_device = A_eval.device
        dev = in_device if in_device.type == "cuda" else ctx.default_device
        if not torch.cuda.is_available():
            raise RuntimeError("MI355 solver needs a ROCm GPU; there is no CPU fallback (use solver='DIFFCP' on CPU)")
        eng = ctx.engine(dev)
        _warn_flagged_adjoint(eng)
        merged_args = {**ctx.options}
        if solver_args:
            merged_args.update(solver_args)
        settings = make_settings(merged_args)
        with torch.cuda.device(dev):
            A_dev = A_eval.detach().to(device=dev, dtype=torch.float64)
            q_dev = q_eval.detach().to(device=dev, dtype=torch.float64)
            batch_minor_in = A_dev.dim() == 2 and A_dev.is_contiguous() and A_dev.shape[1] > 1
            A_bm = eng.to_batch_major(A_dev)
            warm = None
            if warm_start is True:                                   # re-use the previous solution of this layer (same batch size)
                prev = getattr(eng, "_last_solution", None)
                if prev is not None and prev[0].shape[0] == A_bm.shape[0]:
                    warm = prev
            elif warm_start not in (None, False):
                warm = tuple(t if t.dim() == 2 else t.unsqueeze(0) for t in warm_start)     # (x, y, s) tensors
            x, y, s, iters, status, resid = eng.solve(A_bm, q_dev, settings, warm=warm)
            eng._last_solution = (x.detach(), y.detach(), s)
            st = status.cpu()
        if bool((st < 0).any()) and merged_args.get("raise_on_error", True):
            bad = int((st < 0).nonzero()[0])
            raise SolverError(f"Solver mi355 returned status {STATUS_NAMES.get(int(st[bad]), int(st[bad]))} "
                              f"for instance {bad} ({int((st < 0).sum())} of {batch_size} instances failed)")
        if bool((st == 2).any()):
            warnings.warn("Solved/Inaccurate.")
        primal = x.to(in_device)
        dual = y.to(in_device)
        info = dict(iters=iters, status=status, resid=resid)
        # x / y are handed back as `primal` / `dual` (same objects when the input lives on the engine's device), and autograd
        # attaches this node to them: keeping the SAME objects on the node would form a reference cycle (node -> saved ->
        # primal -> grad_fn -> node) that only the cyclic GC breaks, i.e. 167 MB buffers pile up for many steps and the
        # caching allocator falls back to hipMalloc (3 ms each).  Detached aliases share the storage without the cycle.
        saved = (eng, A_bm, x.detach(), y.detach(), s, batch_minor_in) if needs_grad else None
        return primal, dual, info, (saved, batch_size, originally_unbatched, in_device)

    @staticmethod
    def setup_context(ctx, inputs, outputs):
        _, _, info, backward_data = outputs
        ctx.info = info
        ctx.backward_data = backward_data

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dprimal, ddual, _info, _data):
        saved, batch_size, originally_unbatched, in_device = ctx.backward_data
        if saved is None:
            raise RuntimeError("backward called on a layer evaluated with needs_grad=False")
        eng, A_bm, x, y, s, batch_minor_in = saved
        with torch.cuda.device(eng.device):
            dx = dprimal.to(device=eng.device, dtype=torch.float64).contiguous()
            dy = ddual.to(device=eng.device, dtype=torch.float64).contiguous()
            dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy, batch_minor_out=batch_minor_in)
        ctx.adj_status = adj
        eng._pending_adj = (adj, batch_size)     # inspected at the next call (no host sync on the backward path)
        dA = dA.to(in_device)
        dq = dq.to(in_device)
        if originally_unbatched:
            dq = dq.squeeze(1)
            dA = dA.squeeze(1)
        return None, dq, dA, None, None, None, None
