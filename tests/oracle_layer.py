"""A CanonTemplate layer whose solver is the CPU oracle -- TEST INFRASTRUCTURE (imports oracle/): the CPU-side checker of the
notebook fixtures (tests/test_notebook_golden.py) wherever the HIP engine cannot run.

Follows the reference's glue step by step (torch/cvxpylayer.py:84-141 Fortran flattening + canonical column order, :433-451
parameter maps, diffcp_if.py:46-70 (A, b, c) out of A_eval / q_eval, :73-96 gradient packing, :111-117 broadcast-gradient sums)
with scipy / numpy only, so it also runs where /root/reference is absent.  Returns (primal (B, n), dual (B, m)) as torch CPU
tensors that are differentiable w.r.t. the parameters; the caller slices its variables out of them."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from cvxpylayers_amd import problems as P
from oracle import oracle


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, *params):
        tpl = layer.template
        batch = [p.shape[0] if p.dim() == len(s) + 1 else 0 for p, s in zip(params, tpl.param_shapes)]
        B = max(batch + [1])
        pm = np.zeros((tpl.n_params_total + 1, B)); pm[-1] = 1.0
        for p, s, off, bs in zip(params, tpl.param_shapes, tpl.col_offsets, batch):
            v = p.detach().double().numpy()
            v = v if bs else np.broadcast_to(v, (B,) + tuple(s))
            size = int(np.prod(s)) if len(s) else 1
            pm[off:off + size] = np.stack([v[i].reshape(-1, order="F") for i in range(B)], axis=1)
        A_eval = sp.csr_array(tpl.A_map) @ pm; q_eval = sp.csr_array(tpl.q_map) @ pm
        A, b, c = layer.cone.dense_from_values(A_eval, q_eval)
        r = oracle.solve_batch(A, b, c, tpl.cone_dims, **layer.solver_args)
        if (r["status"] < 0).any():
            raise RuntimeError("oracle: " + oracle.STATUS_NAMES[int(r["status"][r["status"] < 0][0])])
        ctx.layer, ctx.batch, ctx.saved = layer, batch, (A, b, c, r)
        layer.info = r
        return torch.from_numpy(r["x"].copy()), torch.from_numpy(r["y"].copy())

    @staticmethod
    def backward(ctx, dx, dy):
        layer, tpl = ctx.layer, ctx.layer.template
        A, b, c, r = ctx.saved
        g = oracle.adjoint_batch(A, b, c, tpl.cone_dims, r["x"], r["y"], r["s"], dx.numpy(), dy.numpy(), mode=layer.adj_mode)
        dA_eval, dq_eval = layer.cone.values_from_dense(g["dA"], g["db"], g["dc"])        # [-dA.data, db[b_idx]], [dc, 0]
        dp = sp.csr_array(tpl.A_map).T @ dA_eval + sp.csr_array(tpl.q_map).T @ dq_eval   # (Ptot+1, B)
        out = [None]
        for s, off, bs in zip(tpl.param_shapes, tpl.col_offsets, ctx.batch):
            size = int(np.prod(s)) if len(s) else 1
            gi = np.stack([dp[off:off + size, i].reshape(tuple(s), order="F") for i in range(dp.shape[1])])
            out.append(torch.from_numpy(gi if bs else gi.sum(0)))
        return tuple(out)


class OracleLayer:
    def __init__(self, template, adj_mode="dense", **solver_args):
        self.template = template
        idx, ptr, (m, n1) = template.A_structure
        self.cone = P.ConeTemplate(n=n1 - 1, m=m, indices=np.asarray(idx), indptr=np.asarray(ptr), cones=dict(template.cone_dims))
        self.solver_args = dict(solver_args)
        self.adj_mode = adj_mode
        self.info = None

    def __call__(self, *params):
        return _Fn.apply(self, *params)


class OraclePlugin(torch.autograd.Function):
    """The plugin calling convention of the reference (`apply(P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad, warm_start) -> (primal, dual,
    aux, data)`, 7-tuple backward; diffcp_if.py:327-403) served by the CPU oracle: what the multi-rank CPU tests shard instead of a stand-in.
    cl_ctx: a CanonTemplate (its A_structure / cone_dims are what a solver ctx carries).  Inputs batch-minor (n+1, B) / (nnz_aug, B)."""

    @staticmethod
    def forward(ctx, P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad=True, warm_start=None):
        tpl = cl_ctx
        idx, ptr, (m, n1) = tpl.A_structure
        cone = P.ConeTemplate(n=n1 - 1, m=m, indices=np.asarray(idx), indptr=np.asarray(ptr), cones=dict(tpl.cone_dims))
        A, b, c = cone.dense_from_values(A_eval.detach().numpy(), q_eval.detach().numpy())
        args = {"eps": 1e-10, "max_iters": 200000, **(solver_args or {})}
        args.pop("mode", None)
        r = oracle.solve_batch(A, b, c, tpl.cone_dims, **args)
        ctx.cone, ctx.saved, ctx.cones = cone, (A, b, c, r), tpl.cone_dims
        return torch.from_numpy(r["x"].copy()), torch.from_numpy(r["y"].copy()), {"iters": torch.from_numpy(r["iters"].copy()), "status": torch.from_numpy(r["status"].copy())}, None

    @staticmethod
    def backward(ctx, dprimal, ddual, _i, _d):
        A, b, c, r = ctx.saved
        g = oracle.adjoint_batch(A, b, c, ctx.cones, r["x"], r["y"], r["s"], dprimal.contiguous().numpy(), ddual.contiguous().numpy(), mode="dense")
        dA_eval, dq_eval = ctx.cone.values_from_dense(g["dA"], g["db"], g["dc"])
        return None, torch.from_numpy(dq_eval), torch.from_numpy(dA_eval), None, None, None, None
