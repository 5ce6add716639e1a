"""Executes the REFERENCE's own Python glue for the hot path, unchanged, with only its two un-installable third-party imports stubbed.

TEST INFRASTRUCTURE (imports oracle/): only tests/ and tests/golden/make_refglue.py use this module.  It needs
/root/reference (this build container); on the GPU box the fixtures it produced (tests/golden/refglue_*.npz) are replayed.

What runs as reference code, from the files where they lie under /root/reference/src/cvxpylayers (never copied):
    torch/cvxpylayer.py     CvxpyLayer.forward :377-490, _flatten_and_batch_params :84-141, _ScipySparseMatmul :12-37,
                            _recover_results :225-282, _unpack_primal_svec / _unpack_svec :183-222, _reshape_fortran :40-55
    utils/parse_args.py     LayersContext.validate_params :94-143, VariableRecovery :56-67
    interfaces/__init__.py  get_torch_cvxpylayer :76-101 (and get_solver_ctx :13-73 for the integration-hunk test)
    interfaces/diffcp_if.py DIFFCP_ctx :99-120, _CvxpyLayer.forward / backward :327-403, _build_diffcp_matrices :46-70,
                            _compute_gradients :73-96, _detect_batch_size :34-43
What is stubbed (the reference's own tests use the same sys.modules technique, tests/test_diffcp_optional_deps.py:17-23):
    cvxpy.*   -- canonicalisation is not on this path: CvxpyLayer.__init__ (parse_args) is bypassed and the LayersContext is
                 filled from a hand-canonicalised template; cvxpy's dims_to_solver_dict is restated (cone dict keys).
    diffcp    -- solve_and_derivative_batch / solve_only_batch / the adjoint closure call the CPU oracle (oracle/cone_oracle.c),
                 with diffcp's calling convention: lists of scipy CSC matrices / vectors in, lists out, dA with A's pattern.
So the boundary conventions (sign of A, position of b inside A_eval, gradient packing [-dA.data, db[b_idx]] / [dc, 0], batch axis
handling, Fortran flattening, column order, svec unpacking) are no longer restated: they are whatever the reference code does.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
from contextlib import contextmanager
from types import SimpleNamespace
from unittest import mock

import numpy as np
import scipy.sparse as sp

REF_SRC = "/root/reference/src"
_PKG = os.path.join(REF_SRC, "cvxpylayers")


def available() -> bool:
    return os.path.isfile(os.path.join(_PKG, "interfaces", "diffcp_if.py"))


# ------------------------------------------------------------------------------------------------ stubs
def _dims_to_solver_dict(dims):
    """cvxpy.reductions.solvers.conic_solvers.scs_conif.dims_to_solver_dict restated: ConeDims -> SCS/diffcp cone dict."""
    if isinstance(dims, dict):
        return dict(dims)
    return {"z": int(dims.zero), "l": int(dims.nonneg), "q": [int(v) for v in dims.soc], "ep": int(dims.exp),
            "s": [int(v) for v in dims.psd], "p": list(getattr(dims, "p3d", []))}


class _DiffcpSolverError(Exception):
    pass


def _make_diffcp_stub():
    from oracle import oracle
    m = types.ModuleType("diffcp")
    m.SolverError = _DiffcpSolverError
    m.calls = []          # (name, kwargs) of every call, for tests that check which options reach the solver

    def _opts(kw):
        kw = {k: v for k, v in kw.items() if k not in ("n_jobs_forward", "n_jobs_backward", "warm_starts", "solve_method", "verbose")}
        kw.pop("mode", None)
        return kw

    def _solve(As, bs, cs, cone_dicts, kw):
        A = np.stack([a.toarray() for a in As]); b = np.stack(bs); c = np.stack(cs)
        cones = cone_dicts[0]
        warm = kw.get("warm_starts")
        wt = None
        if warm is not None:
            wt = tuple(np.stack([np.asarray(w[k]) for w in warm]) for k in range(3))
        r = oracle.solve_batch(A, b, c, cones, warm=wt, **_opts(kw))
        if (r["status"] < 0).any():
            bad = int(np.nonzero(r["status"] < 0)[0][0])
            raise _DiffcpSolverError(f"Solver scs returned status {oracle.STATUS_NAMES[int(r['status'][bad])]}")
        return A, b, c, cones, r

    def solve_only_batch(As, bs, cs, cone_dicts, **kw):
        m.calls.append(("solve_only_batch", dict(kw)))
        _, _, _, _, r = _solve(As, bs, cs, cone_dicts, kw)
        return list(r["x"]), list(r["y"]), list(r["s"])

    def solve_and_derivative_batch(As, bs, cs, cone_dicts, **kw):
        m.calls.append(("solve_and_derivative_batch", dict(kw)))
        A, b, c, cones, r = _solve(As, bs, cs, cone_dicts, kw)
        mode = kw.get("mode", "lsqr")

        def D_batch(*a, **k):
            raise NotImplementedError("forward-mode derivative is not on the cvxpylayers path")

        def DT_batch(dxs, dys, dss, **k):
            g = oracle.adjoint_batch(A, b, c, cones, r["x"], r["y"], r["s"], np.stack(dxs), np.stack(dys), np.stack(dss),
                                     mode=("dense" if mode == "dense" else "lsqr"))
            dAs = []
            for i, Ai in enumerate(As):        # diffcp returns dA on A's sparsity pattern (CSC, same index arrays)
                cols = np.repeat(np.arange(Ai.shape[1]), np.diff(Ai.indptr))
                dAs.append(sp.csc_matrix((g["dA"][i][Ai.indices, cols], Ai.indices.copy(), Ai.indptr.copy()), shape=Ai.shape))
            return dAs, list(g["db"]), list(g["dc"])
        return list(r["x"]), list(r["y"]), list(r["s"]), D_batch, DT_batch

    m.solve_only_batch = solve_only_batch
    m.solve_and_derivative_batch = solve_and_derivative_batch
    return m


_CVXPY_STUBS = ["cvxpy", "cvxpy.constraints", "cvxpy.utilities", "cvxpy.utilities.scopes", "cvxpy.reductions", "cvxpy.reductions.dcp2cone",
                "cvxpy.reductions.dcp2cone.cone_matrix_stuffing", "cvxpy.reductions.solvers", "cvxpy.reductions.solvers.conic_solvers",
                "cvxpy.reductions.solvers.conic_solvers.scs_conif"]


def _quad_obj_set():
    """SUPPORTS_QUAD_OBJ as the reference defines it (_quad_form_dpp.py:32); the rest of that module patches CVXPY internals
    (canonicalisation side, not on this path) and is not executed."""
    import re
    src = open(os.path.join(_PKG, "_quad_form_dpp.py")).read()
    mt = re.search(r"SUPPORTS_QUAD_OBJ\s*=\s*frozenset\(\{([^}]*)\}\)", src)
    return frozenset(s.strip().strip('"\'') for s in mt.group(1).split(",") if s.strip())


@contextmanager
def reference_modules():
    """Installs the stubs, imports the reference modules from /root/reference/src unchanged, yields a namespace with
    .cvxpylayer (torch/cvxpylayer.py), .diffcp_if, .interfaces, .parse_args, .diffcp (the stub); restores sys.modules on exit."""
    if not available():
        raise RuntimeError("/root/reference is not present")
    saved = {k: v for k, v in sys.modules.items() if k == "diffcp" or k.split(".")[0] in ("cvxpy", "cvxpylayers", "jax", "mlx")}
    for k in saved:
        del sys.modules[k]
    try:
        for name in _CVXPY_STUBS:
            sys.modules[name] = mock.MagicMock(name=name)
        sys.modules["cvxpy.reductions.solvers.conic_solvers.scs_conif"].dims_to_solver_dict = _dims_to_solver_dict
        sys.modules["jax"] = None; sys.modules["jax.numpy"] = None; sys.modules["mlx"] = None; sys.modules["mlx.core"] = None   # optional frontends absent
        sys.modules["diffcp"] = _make_diffcp_stub()
        pkg = types.ModuleType("cvxpylayers"); pkg.__path__ = [_PKG]; pkg.__version__ = "reference (stubbed cvxpy / diffcp)"
        sys.modules["cvxpylayers"] = pkg
        qd = types.ModuleType("cvxpylayers._quad_form_dpp"); qd.SUPPORTS_QUAD_OBJ = _quad_obj_set()
        sys.modules["cvxpylayers._quad_form_dpp"] = qd
        ns = SimpleNamespace(
            diffcp_if=importlib.import_module("cvxpylayers.interfaces.diffcp_if"),
            interfaces=importlib.import_module("cvxpylayers.interfaces"),
            parse_args=importlib.import_module("cvxpylayers.utils.parse_args"),
            cvxpylayer=importlib.import_module("cvxpylayers.torch.cvxpylayer"),
            diffcp=sys.modules["diffcp"])
        for mod in (ns.diffcp_if, ns.interfaces, ns.parse_args, ns.cvxpylayer):
            assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REF_SRC)), mod.__file__
        yield ns
    finally:
        for k in [k for k in sys.modules if k == "diffcp" or k.split(".")[0] in ("cvxpy", "cvxpylayers", "jax", "mlx")]:
            del sys.modules[k]
        sys.modules.update(saved)


# ------------------------------------------------------------------------------------------------ reference layer on a template
def reference_layer(ns, template, solver="DIFFCP", solver_args=None):
    """A reference `cvxpylayers.torch.CvxpyLayer` whose canonicalisation products come from `template` (CanonTemplate) instead of
    CVXPY: __init__ (parse_args) is bypassed, everything forward() touches is set exactly as __init__ would (:346-375)."""
    import torch
    pa, cl = ns.parse_args, ns.cvxpylayer
    idx, ptr, shape = template.A_structure
    P_idx = template.P_structure if template.P_map is not None else None
    param_prob = SimpleNamespace(reduced_A=SimpleNamespace(problem_data_index=(np.asarray(idx), np.asarray(ptr), tuple(shape)), reduced_mat=template.A_map),
                                 reduced_P=SimpleNamespace(problem_data_index=P_idx, reduced_mat=template.P_map))
    solver_ctx = ns.interfaces.get_solver_ctx(solver, param_prob, dict(template.cone_dims), {}, dict(solver_args or {}))
    order = np.argsort(np.argsort(template.col_offsets))          # rank of each user parameter in canonical column order
    ctx = pa.LayersContext(
        parameters=[SimpleNamespace(shape=tuple(s)) for s in template.param_shapes], reduced_P=param_prob.reduced_P, q=template.q_map,
        reduced_A=param_prob.reduced_A, cone_dims=dict(template.cone_dims), solver_ctx=solver_ctx, solver=solver,
        var_recover=[pa.VariableRecovery(primal=v.primal, dual=v.dual, shape=tuple(v.shape), is_symmetric=(v.unpack_fn == "svec_primal"),
                                         is_psd_dual=(v.unpack_fn == "svec_dual" and v.source == "dual"), source=v.source, unpack_fn=v.unpack_fn)
                     for v in template.var_recover],
        user_order_to_col_order=tuple(int(o) for o in order), gp=template.gp, gp_log_mask=template.gp_log_mask)
    layer = cl.CvxpyLayer.__new__(cl.CvxpyLayer)
    torch.nn.Module.__init__(layer)
    layer.ctx = ctx
    layer.P = None; layer._P_scipy = None
    if template.P_map is not None:
        layer._P_scipy = sp.csr_array(template.P_map)
    layer._q_scipy = sp.csr_array(template.q_map)
    layer._A_scipy = sp.csr_array(template.A_map)
    layer._warm_start_cache = None
    return layer


@contextmanager
def record_boundary(ns, solver="DIFFCP"):
    """Wraps the plugin class the reference frontend obtains from get_torch_cvxpylayer so that the tensors crossing the plugin
    boundary are kept: rec.q_eval / A_eval (inputs), rec.primal / dual (outputs); after backward() their .grad fields hold
    dq_eval / dA_eval and dprimal / ddual."""
    rec = SimpleNamespace()
    inner = ns.interfaces.get_torch_cvxpylayer(solver)

    class Recorder:
        @staticmethod
        def apply(P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad, warm_start):
            if needs_grad:
                q_eval.retain_grad(); A_eval.retain_grad()
            rec.q_eval, rec.A_eval = q_eval, A_eval
            primal, dual, a, b = inner.apply(P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad, warm_start)
            if needs_grad:
                primal.retain_grad(); dual.retain_grad()
            rec.primal, rec.dual = primal, dual
            return primal, dual, a, b
    orig = ns.interfaces.get_torch_cvxpylayer
    ns.interfaces.get_torch_cvxpylayer = lambda s: Recorder if s == solver else orig(s)
    try:
        yield rec
    finally:
        ns.interfaces.get_torch_cvxpylayer = orig


def run_case(ns, case, solver_args):
    """Forward + backward of one tests/ref_cases.py case through the reference frontend and plugin.  Returns a dict of numpy arrays."""
    import torch
    layer = reference_layer(ns, case["template"])
    params = [torch.tensor(np.asarray(p), dtype=torch.float64, requires_grad=True) for p in case["params"]]
    with record_boundary(ns) as rec:
        outs = layer(*params, solver_args=dict(solver_args))
        loss = sum((o * torch.as_tensor(w)).sum() for o, w in zip(outs, case["weights"]))
        loss.backward()
    out = {}
    for k, p in enumerate(params):
        out[f"param{k}"] = p.detach().numpy(); out[f"grad{k}"] = p.grad.numpy()
    for k, (o, w) in enumerate(zip(outs, case["weights"])):
        out[f"out{k}"] = o.detach().numpy(); out[f"weight{k}"] = np.asarray(w)
    for name in ("q_eval", "A_eval", "primal", "dual"):
        t = getattr(rec, name)
        out[name] = t.detach().numpy()
        out["d" + name] = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
    return out
