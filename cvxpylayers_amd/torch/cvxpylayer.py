"""PyTorch frontend for the MI355 engine, mirroring cvxpylayers.torch.CvxpyLayer
(reference: src/cvxpylayers/torch/cvxpylayer.py:285-490) for the one solver key this repository provides.

What is the same as the reference (argument meaning, shapes, errors):
  * CvxpyLayer(...).forward(*params, solver_args=None): one tensor per parameter, each with the parameter's shape or
    with one leading batch axis; batched and unbatched parameters may be mixed (unbatched ones are broadcast); all batched
    parameters must share the batch size; wrong counts / shapes raise ValueError with the reference's messages
    (utils/parse_args.py:94-143).
  * parameters are flattened in Fortran order, stacked in the canonical column order, followed by a constant 1
    (torch/cvxpylayer.py:84-141), and pushed through the affine parameter maps  A_eval = A_map p,  q_eval = q_map p  (:433-451).
  * the plugin is called as _CvxpyLayer.apply(P_eval, q_eval, A_eval, ctx, solver_args, needs_grad, warm_start) (:475-483) and
    variables are recovered by slicing primal / dual, Fortran reshape, svec -> symmetric unpacking (:225-282, :144-222).
  * a batch of one is not the same as unbatched: outputs keep the leading axis exactly when an input had it.

What is different (MI355X-first):
  * the parameter maps are evaluated ON THE DEVICE by ce_parammap_apply, batch-major: p is (B, Ptot+1) row-major -- the
    natural layout of the flattened parameters -- and A_eval comes out as (B, nnz_aug) row-major, which is the engine's native
    layout.  The reference's p_stack transpose, its (nnz_aug, B) result and the engine's layout pass all disappear; the plugin
    receives the transposed VIEW (nnz_aug, B) so the boundary convention is unchanged.
  * canonicalisation (host, once) is decoupled: the layer is built from a `CanonTemplate`.  With CVXPY installed,
    `CanonTemplate.from_cvxpy(problem, parameters, variables)` produces it exactly as utils/parse_args.py:388-514 does
    (canonicalising as "DIFFCP"); without CVXPY (this build image) templates are built by hand / by affine probing
    (cvxpylayers_amd.torch.templates).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Any, Sequence

import numpy as np
import scipy.sparse as sp
import torch

from cvxpylayers_amd import _lib
from cvxpylayers_amd.interfaces import get_solver_ctx, get_torch_cvxpylayer


@dataclass
class VariableRecovery:
    """How one requested variable is cut out of the solver's primal / dual vector (utils/parse_args.py:56-67)."""
    primal: slice | None
    dual: slice | None
    shape: tuple
    source: str = "primal"            # "primal" | "dual"
    unpack_fn: str = "reshape"        # "reshape" | "svec_primal" | "svec_dual"


@dataclass
class CanonTemplate:
    """Everything CVXPY's canonicalisation yields for a DPP problem (host side, built once):
    param_shapes   shapes of the parameters in USER order
    col_offsets    first column of each parameter (user order) in the canonical parameter vector; the constant 1 is last
    A_map          scipy CSR (nnz_aug, Ptot+1): values of the CSC data of [A_cvx | b_cvx] as an affine function of p
    q_map          scipy CSR (n+1, Ptot+1)
    A_structure    (indices, indptr, (m, n+1))  CSC structure of [A_cvx | b_cvx]
    cone_dims      {"z","l","q","ep","s","p"}
    var_recover    one VariableRecovery per requested variable
    """
    param_shapes: list
    col_offsets: list
    A_map: sp.csr_array
    q_map: sp.csr_array
    A_structure: tuple
    cone_dims: dict
    var_recover: list
    P_map: Any = None              # (nnz_P, Ptot+1) map of the quadratic objective 1/2 x^T P x (plugins in SUPPORTS_QUAD_OBJ), or None
    P_structure: Any = None        # (indices, indptr, (n, n)) CSC structure of P (param_prob.reduced_P.problem_data_index)
    gp: bool = False
    gp_log_mask: tuple | None = None

    @property
    def n_params_total(self) -> int:
        return int(self.A_map.shape[1]) - 1

    @staticmethod
    def from_cvxpy(problem, parameters, variables, gp: bool = False):
        """Canonicalise with CVXPY (as "DIFFCP", i.e. SCS cone form and CSC structure).  Needs cvxpy + cvxpylayers."""
        try:
            import cvxpylayers.utils.parse_args as pa  # type: ignore
        except Exception as e:  # pragma: no cover - not installable in the build image
            raise ImportError("CanonTemplate.from_cvxpy needs cvxpy and cvxpylayers; build a template by hand "
                              "(cvxpylayers_amd.torch.templates) when they are not installed") from e
        ctx = pa.parse_args(problem, variables, parameters, "DIFFCP", gp=gp, verbose=False, canon_backend=None, solver_args={})
        sizes = [int(np.prod(p.shape)) if p.shape else 1 for p in parameters]
        order = ctx.user_order_to_col_order
        offs_by_col = np.concatenate([[0], np.cumsum([sizes[list(order).index(k)] for k in range(len(parameters))])])
        col_offsets = [int(offs_by_col[order[i]]) for i in range(len(parameters))]
        rec = [VariableRecovery(v.primal, v.dual, tuple(v.shape), v.source, v.unpack_fn) for v in ctx.var_recover]
        sc = ctx.solver_ctx
        return CanonTemplate([tuple(p.shape) for p in parameters], col_offsets, ctx.reduced_A.reduced_mat.tocsr(), ctx.q.tocsr(),
                             (sc.A_structure[0], sc.A_structure[1], sc.A_shape), dict(sc.dims) if isinstance(sc.dims, dict) else sc.dims,
                             rec, gp=gp, gp_log_mask=ctx.gp_log_mask)


def _reshape_fortran(x: torch.Tensor, shape: tuple) -> torch.Tensor:
    """Column-major reshape (torch/cvxpylayer.py:40-55)."""
    if x.dim() == 0:
        return x.reshape(shape)
    xt = x.permute(*reversed(range(x.dim())))
    return xt.reshape(*reversed(shape)).permute(*reversed(range(len(shape))))


class _DeviceCSR:
    """A scipy CSR matrix and its transpose as int32/fp64 device arrays (lazily per device)."""

    def __init__(self, mat: sp.csr_array):
        self.mat = sp.csr_array(mat).astype(np.float64)
        self.mat.sort_indices()
        self.matT = sp.csr_array(self.mat.T.tocsr())
        self.matT.sort_indices()
        self._dev: dict = {}

    def on(self, device: torch.device):
        key = (device.type, device.index)
        if key not in self._dev:
            def up(m):
                return (torch.from_numpy(m.indptr.astype(np.int32)).to(device), torch.from_numpy(m.indices.astype(np.int32)).to(device),
                        torch.from_numpy(m.data.astype(np.float64)).to(device), int(m.shape[0]))
            self._dev[key] = (up(self.mat), up(self.matT))
        return self._dev[key]


def _spmm_bm(arrs, P: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """out (B, rows) = P (B, cols) . map^T on the device through the C ABI (ce_parammap_apply2); with `out` given, accumulate."""
    indptr, indices, vals, rows = arrs
    B = P.shape[0]
    acc = out is not None
    if out is None:
        out = torch.empty((B, rows), dtype=torch.float64, device=P.device)
    rc = _lib.lib().ce_parammap_apply2(P.device.index or 0, B, rows, P.shape[1], int(acc), indptr.data_ptr(), indices.data_ptr(), vals.data_ptr(),
                                       P.data_ptr(), P.stride(0), out.data_ptr(), out.stride(0),
                                       C.c_void_p(torch.cuda.current_stream(P.device).cuda_stream))
    _lib.check(rc, "ce_parammap_apply2")
    return out


class _ParamMap1(torch.autograd.Function):
    """One map (the quadratic objective's P_eval = P_map p), batch-major, transpose as backward."""

    @staticmethod
    def forward(ctx, csr: _DeviceCSR, p_bm: torch.Tensor):
        fwd, bwd = csr.on(p_bm.device)
        ctx.bwd = bwd
        return _spmm_bm(fwd, p_bm.contiguous())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return None, _spmm_bm(ctx.bwd, g.contiguous())


class _ParamMapApply(torch.autograd.Function):
    """Batch-major evaluation of both parameter maps, (A_bm, q_bm) = (p A_map^T, p q_map^T), with the transposed maps as backward
    accumulated into one p gradient (reference: _ScipySparseMatmul applied once per map, :12-37, and autograd's add)."""

    @staticmethod
    def forward(ctx, A_csr: _DeviceCSR, q_csr: _DeviceCSR, p_bm: torch.Tensor):
        p_bm = p_bm.contiguous()
        fA, bA = A_csr.on(p_bm.device)
        fq, bq = q_csr.on(p_bm.device)
        ctx.bwd = (bA, bq)
        return _spmm_bm(fA, p_bm), _spmm_bm(fq, p_bm)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gA, gq):
        bA, bq = ctx.bwd
        g = _spmm_bm(bA, gA.contiguous())
        _spmm_bm(bq, gq.contiguous(), out=g)
        return None, None, g


def _svec_to_symmetric(svec, k, batch, rows, cols, scale=None):
    rows_t = torch.as_tensor(rows, dtype=torch.long, device=svec.device)
    cols_t = torch.as_tensor(cols, dtype=torch.long, device=svec.device)
    data = svec * torch.as_tensor(scale, dtype=svec.dtype, device=svec.device) if scale is not None else svec
    out = torch.zeros(batch + (k, k), dtype=svec.dtype, device=svec.device)
    out[..., rows_t, cols_t] = data
    out[..., cols_t, rows_t] = data
    return out


def _unpack_primal_svec(svec, k, batch):
    """Symmetric primal variable: upper triangle, row-major, unscaled (torch/cvxpylayer.py:183-198)."""
    rows, cols = np.triu_indices(k)
    return _svec_to_symmetric(svec, k, batch, rows, cols)


def _unpack_svec(svec, k, batch):
    """PSD dual: lower triangle, column-major, off-diagonals scaled by sqrt(2) (torch/cvxpylayer.py:201-222)."""
    rr, cc = np.tril_indices(k)
    order = np.lexsort((rr, cc))
    rows, cols = rr[order], cc[order]
    return _svec_to_symmetric(svec, k, batch, rows, cols, np.where(rows == cols, 1.0, 1.0 / np.sqrt(2.0)))


def _recovery_map(var_recover, n_src: int, source: str):
    """The variable recovery of one source (primal or dual) as a sparse map R (sum of output sizes x n_src): output element t of a
    variable, in the row-major order of its final shape, is scale * src[index].  Slices + Fortran reshapes give one entry per row;
    svec unpacking gives the 1 (primal) or 1/sqrt(2) off-diagonal (PSD dual) weights of torch/cvxpylayer.py:183-222.  Returns
    (csr, [(variable position, row offset, size)])."""
    rows, cols, vals, layout, off = [], [], [], [], 0
    for pos, var in enumerate(var_recover):
        if var.source != source:
            continue
        sl = var.primal if source == "primal" else var.dual
        src = np.arange(n_src)[sl]
        shape = tuple(var.shape)
        size = int(np.prod(shape)) if len(shape) else 1
        if var.unpack_fn == "reshape":
            # out.reshape(-1)[t] = data[f(t)] with data Fortran-ordered over `shape`
            idx = np.arange(size).reshape(shape, order="F").reshape(-1) if len(shape) > 1 else np.arange(size)
            rows.append(off + np.arange(size)); cols.append(src[idx]); vals.append(np.ones(size))
        elif var.unpack_fn in ("svec_primal", "svec_dual"):
            k = shape[0]
            if var.unpack_fn == "svec_primal":
                rr, cc = np.triu_indices(k)
                w = np.ones(rr.size)
            else:
                r0, c0 = np.tril_indices(k)
                order = np.lexsort((r0, c0))
                rr, cc = r0[order], c0[order]
                w = np.where(rr == cc, 1.0, 1.0 / np.sqrt(2.0))
            rows.append(off + rr * k + cc); cols.append(src); vals.append(w)
            offd = rr != cc
            rows.append(off + cc[offd] * k + rr[offd]); cols.append(src[offd]); vals.append(w[offd])
        else:
            raise ValueError(f"Unknown variable recovery type: {var.unpack_fn}")
        layout.append((pos, off, size))
        off += size
    if not layout:
        return None, []
    mat = sp.csr_array((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(off, n_src))
    return mat, layout


class _RecoverMap(torch.autograd.Function):
    """Variable recovery of one source as ONE launch of the sparse-map kernel (ce_parammap_apply2) writing every requested variable
    in its final layout, and its transpose as the backward (gradients of all variables gathered into d primal / d dual in one
    launch).  Replaces the per-variable slice / permute / index_put chain of _recover_results_torch, which stays as the checker."""

    @staticmethod
    def forward(ctx, csr: _DeviceCSR, layout, shapes, src: torch.Tensor):
        fwd, bwd = csr.on(src.device)
        ctx.bwd, ctx.layout, ctx.total, ctx.src_shape = bwd, layout, fwd[3], src.shape
        src2 = src.reshape(-1, src.shape[-1]).contiguous()
        with torch.cuda.device(src.device):
            rec = _spmm_bm(fwd, src2)
        lead = tuple(src.shape[:-1])
        return tuple(rec[:, off:off + size].reshape(lead + tuple(shape)) if len(layout) > 1 else rec.reshape(lead + tuple(shape))
                     for (_, off, size), shape in zip(layout, shapes))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        B = int(np.prod(ctx.src_shape[:-1])) if len(ctx.src_shape) > 1 else 1
        ref = next(g for g in grads if g is not None)
        parts = [g.reshape(B, size) if g is not None else torch.zeros((B, size), dtype=ref.dtype, device=ref.device)
                 for g, (_, _, size) in zip(grads, ctx.layout)]
        g_rec = parts[0].contiguous() if len(parts) == 1 else torch.cat(parts, dim=1)
        with torch.cuda.device(ref.device):
            d_src = _spmm_bm(ctx.bwd, g_rec)
        return None, None, None, d_src.reshape(ctx.src_shape)


class _Ctx:
    """The object handed to the plugin as cl_ctx (the plugin only needs .solver_ctx)."""

    def __init__(self, solver_ctx, solver):
        self.solver_ctx = solver_ctx
        self.solver = solver


class _ParamProbView:
    """Minimal stand-in for CVXPY's ParamConeProg as far as interfaces.get_solver_ctx reads it."""

    class _Red:
        def __init__(self, pdi, mat=None):
            self.problem_data_index = pdi
            self.reduced_mat = mat          # the parameter map (ParamConeProg.reduced_A.reduced_mat): lets the plugin decide structurally whether A is batch-invariant

    def __init__(self, a_structure, p_structure=None, a_map=None):
        self.reduced_A = self._Red(a_structure, a_map)
        self.reduced_P = self._Red(p_structure) if p_structure is not None else None


class CvxpyLayer(torch.nn.Module):
    """Differentiable cone-program layer on MI355X.

    CvxpyLayer(problem, parameters, variables, solver="MI355", gp=False, solver_args=None)   # needs cvxpy
    CvxpyLayer(template=CanonTemplate(...), solver_args=None)                                # cvxpy-free
    """

    def __init__(self, problem=None, parameters: Sequence | None = None, variables: Sequence | None = None, solver: str | None = None,
                 gp: bool = False, verbose: bool = False, canon_backend=None, solver_args: dict | None = None,
                 template: CanonTemplate | None = None):
        super().__init__()
        if template is None:
            if problem is None:
                raise ValueError("CvxpyLayer needs either a CVXPY problem or a CanonTemplate")
            template = CanonTemplate.from_cvxpy(problem, parameters, variables, gp=gp)
        solver = solver or "MI355"
        if solver != "MI355":
            raise RuntimeError("Unknown solver. Check if your solver is supported by CVXPYlayers")
        self.template = template
        self.solver = solver
        opts = dict(solver_args or {})
        solver_ctx = get_solver_ctx(solver, _ParamProbView(template.A_structure, template.P_structure if template.P_map is not None else None, template.A_map),
                                    template.cone_dims, {}, opts, verbose)
        self.ctx = _Ctx(solver_ctx, solver)
        # The maps address parameters Fortran-flattened (the reference's p_stack); the device copies are re-indexed once to the
        # parameters' native row-major layout so flattening a batched parameter is a contiguous copy, not a strided transpose.
        colmap = np.arange(template.n_params_total + 1)
        for shape, off in zip(template.param_shapes, template.col_offsets):
            size = int(np.prod(shape)) if len(shape) else 1
            if len(shape) > 1:
                colmap[off:off + size] = off + np.arange(size).reshape(shape).reshape(-1, order="F")
        inv = np.empty_like(colmap)
        inv[colmap] = np.arange(colmap.size)

        def recol(mat):
            # new column c holds the parameter entry that was at Fortran column f: new = old[:, f(c)], with f(c) = position of c in colmap
            return sp.csr_array(sp.csr_array(mat)[:, inv])
        self._A = _DeviceCSR(recol(template.A_map))
        self._q = _DeviceCSR(recol(template.q_map))
        self._P = _DeviceCSR(recol(template.P_map)) if template.P_map is not None else None
        self.batch_sizes: list | None = None
        # variable recovery as one sparse-map launch per source (and its transpose in backward); CE_FUSED_RECOVERY=0 keeps the
        # per-variable torch chain (the checker of tests/test_gpu_recovery.py)
        n_primal = int(template.q_map.shape[0]) - 1
        n_dual = int(template.A_structure[2][0])
        self._rec = {}
        for source, n_src in (("primal", n_primal), ("dual", n_dual)):
            mat, layout = _recovery_map(template.var_recover, n_src, source)
            if mat is not None:
                self._rec[source] = (_DeviceCSR(mat), layout)
        self.fused_recovery = os.environ.get("CE_FUSED_RECOVERY", "1") != "0"

    # ---- utils/parse_args.py:94-143
    def validate_params(self, values: list) -> tuple:
        shapes = self.template.param_shapes
        if len(values) != len(shapes):
            raise ValueError("A tensor must be provided for each CVXPY parameter; "
                             f"received {len(values)} tensors, expected {len(shapes)}")
        batch_sizes = []
        for i, (value, shape) in enumerate(zip(values, shapes)):
            shape = tuple(shape)
            if value.dim() == len(shape):
                if tuple(value.shape) != shape:
                    raise ValueError(f"Invalid parameter shape for parameter {i}. Expected: {shape}, Got: {tuple(value.shape)}")
                batch_sizes.append(0)
            elif value.dim() == len(shape) + 1:
                if tuple(value.shape[1:]) != shape:
                    raise ValueError(f"Invalid parameter shape for parameter {i}. Expected batched shape: "
                                     f"(batch_size, {', '.join(map(str, shape))}), Got: {tuple(value.shape)}")
                batch_sizes.append(int(value.shape[0]))
            else:
                raise ValueError(f"Invalid parameter dimensionality for parameter {i}. Expected {len(shape)} or "
                                 f"{len(shape) + 1} dimensions, Got: {value.dim()} dimensions")
        nonzero = [b for b in batch_sizes if b > 0]
        self.batch_sizes = batch_sizes
        if nonzero:
            if not all(b == nonzero[0] for b in nonzero):
                raise ValueError("Inconsistent batch sizes. Expected all batched parameters to have the same batch size, "
                                 f"but got: {batch_sizes}")
            return (nonzero[0],)
        return ()

    def _flatten_params(self, params, batch) -> torch.Tensor:
        """(B, Ptot+1) row-major parameter matrix: flattened parameters at their canonical column ranges, then the constant 1
        (the batch-major transpose of the reference's p_stack, torch/cvxpylayer.py:84-141)."""
        B = batch[0] if batch else 1
        dev = params[0].device
        tot = self.template.n_params_total
        flats = []
        for i, value in enumerate(params):
            v = value.to(torch.float64)
            if self.batch_sizes[i] == 0:
                v = v.unsqueeze(0).expand((B,) + tuple(v.shape))
            flats.append(v.reshape(B, -1))                       # row-major; the device maps were re-indexed to match (__init__)
        order = sorted(range(len(flats)), key=lambda i: self.template.col_offsets[i])
        pos, tiled = 0, True
        for i in order:
            tiled = tiled and self.template.col_offsets[i] == pos
            pos += flats[i].shape[1]
        if tiled and pos == tot:
            # one concatenation: its backward hands each parameter a view of the gradient (in-place slice writes would make autograd
            # clone the whole (B, Ptot+1) gradient once per parameter)
            return torch.cat([flats[i] for i in order] + [torch.ones((B, 1), dtype=torch.float64, device=dev)], dim=1)
        p = torch.zeros((B, tot + 1), dtype=torch.float64, device=dev)
        p[:, tot] = 1.0
        for i, flat in enumerate(flats):
            off = self.template.col_offsets[i]
            p[:, off:off + flat.shape[1]] = flat
        return p

    def forward(self, *params: torch.Tensor, solver_args: dict | None = None, warm_start: bool = False):
        solver_args = solver_args or {}
        batch = self.validate_params(list(params))
        if self.template.gp and self.template.gp_log_mask is not None:
            params = tuple(torch.log(p) if lg else p for p, lg in zip(params, self.template.gp_log_mask))
        if params[0].device.type != "cuda":
            raise RuntimeError("MI355 solver needs parameters on a ROCm device; there is no CPU fallback (use solver='DIFFCP' on CPU)")
        with torch.cuda.device(params[0].device):
            p_bm = self._flatten_params(params, batch)
            A_bm, q_bm = _ParamMapApply.apply(self._A, self._q, p_bm)   # (B, nnz_aug) engine-native, (B, n+1)
            P_eval = _ParamMap1.apply(self._P, p_bm).t() if self._P is not None else None
        A_eval, q_eval = A_bm.t(), q_bm.t()                     # the reference's (nnz_aug, B) / (n+1, B), as views
        if not batch:
            A_eval, q_eval = A_eval.squeeze(1), q_eval.squeeze(1)
            P_eval = P_eval.squeeze(1) if P_eval is not None else None
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        layer_cls = get_torch_cvxpylayer(self.solver)
        # warm_start=True: the plugin starts from this layer's previous solution when the batch size matches (the reference keeps
        # the same cache for its MOREAU plugin, torch/cvxpylayer.py:464-487)
        primal, dual, info, _ = layer_cls.apply(P_eval, q_eval, A_eval, self.ctx, solver_args, needs_grad, True if warm_start else None)
        self.info = info
        if self.fused_recovery:
            return self._recover_results(primal, dual, batch)
        return self._recover_results_torch(primal, dual, batch)

    def _recover_results(self, primal, dual, batch):
        """torch/cvxpylayer.py:225-282 as one map launch per source (see _RecoverMap)."""
        out = [None] * len(self.template.var_recover)
        for source, src in (("primal", primal), ("dual", dual)):
            if source not in self._rec:
                continue
            csr, layout = self._rec[source]
            shapes = [tuple(self.template.var_recover[pos].shape) for pos, _, _ in layout]
            res = _RecoverMap.apply(csr, layout, shapes, src)
            for (pos, _, _), r in zip(layout, res):
                r = r.reshape(batch + tuple(self.template.var_recover[pos].shape))
                if self.template.gp and source == "primal":
                    r = torch.exp(r)
                out[pos] = r
        return tuple(out)

    # ---- torch/cvxpylayer.py:225-282, per variable with torch ops (checker of the fused path)
    def _recover_results_torch(self, primal, dual, batch):
        internal = tuple(primal.shape[:-1])
        out = []
        for var in self.template.var_recover:
            data = primal[..., var.primal] if var.source == "primal" else dual[..., var.dual]
            if var.unpack_fn == "svec_primal":
                res = _unpack_primal_svec(data, var.shape[0], internal)
            elif var.unpack_fn == "svec_dual":
                res = _unpack_svec(data, var.shape[0], internal)
            elif var.unpack_fn == "reshape":
                res = _reshape_fortran(data, internal + tuple(var.shape))
            else:
                raise ValueError(f"Unknown variable recovery type: {var.unpack_fn}")
            res = res.reshape(batch + tuple(var.shape))
            if self.template.gp and var.source == "primal":
                res = torch.exp(res)
            out.append(res)
        return tuple(out)
