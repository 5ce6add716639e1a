"""Boundary pinned on EXECUTED reference code (CPU; needs /root/reference, i.e. the build container -- skipped elsewhere).

tests/ref_glue.py runs the reference's frontend (torch/cvxpylayer.py) and DIFFCP plugin (interfaces/diffcp_if.py) unchanged, with
cvxpy stubbed (canonicalisation is bypassed: hand-canonicalised templates) and diffcp's two calls served by the CPU oracle.
Checked here:
  * the committed fixtures tests/golden/refglue_*.npz are what that run produces (so the GPU replay, tests/test_gpu_refglue.py,
    compares the HIP path with outputs of the reference's own glue);
  * that stack reproduces the closed forms the reference's tests assert (values and gradients), i.e. reference glue + oracle
    is a faithful stand-in for reference glue + diffcp on these problems;
  * the repository's restated conventions equal what the reference functions do: (A, b, c) cut out of A_eval / q_eval
    (_build_diffcp_matrices), gradient packing (_compute_gradients), parameter flattening + column order
    (_flatten_and_batch_params vs the frontend's row-major re-indexed maps), variable recovery (_recover_results), error
    messages of validate_params;
  * the integration hunks of INTEGRATION.md apply to the reference sources and route solver="MI355" to this repository's plugin.
"""
import os
import re

import numpy as np
import pytest
import torch

import ref_cases
import ref_glue

pytestmark = pytest.mark.skipif(not ref_glue.available(), reason="/root/reference is not present (GPU box): fixtures are replayed instead")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ns():
    with ref_glue.reference_modules() as n:
        yield n


@pytest.mark.parametrize("name", list(ref_cases.CASES))
def test_fixtures_are_outputs_of_the_reference_glue(ns, name):
    got = ref_glue.run_case(ns, ref_cases.CASES[name](), {**ref_cases.SOLVER_ARGS, "mode": "dense"})
    want = np.load(os.path.join(GOLD, f"refglue_{name}.npz"))
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-9, atol=1e-11, err_msg=f"{name}:{k}")


def test_reference_glue_plus_oracle_reproduces_the_reference_tests_closed_forms(ns):
    # tests/test_torch.py:90-118 (ridge value + gradient, atol 1e-6 at eps 1e-10) through the reference's own frontend + plugin
    f = np.load(os.path.join(GOLD, "refglue_ridge_batched_matrix_param.npz"))
    F = torch.tensor(f["param0"], requires_grad=True); g = torch.tensor(f["param1"], requires_grad=True)
    n = F.shape[2]
    x = torch.linalg.solve(F.transpose(1, 2) @ F + torch.eye(n, dtype=torch.float64), (F.transpose(1, 2) @ g[:, :, None]))[:, :, 0]
    (x * torch.tensor(f["weight0"])).sum().backward()
    np.testing.assert_allclose(f["out0"], x.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(f["grad0"], F.grad.numpy(), atol=1e-6)
    np.testing.assert_allclose(f["grad1"], g.grad.numpy(), atol=1e-6)
    # broadcast parameter: the gradient is the SUM over the batch (tests/test_torch.py:355-384)
    f = np.load(os.path.join(GOLD, "refglue_ridge_mixed.npz"))
    F = torch.tensor(f["param0"], requires_grad=True); g = torch.tensor(f["param1"], requires_grad=True)
    x = torch.linalg.solve(F.t() @ F + torch.eye(F.shape[1], dtype=torch.float64), F.t() @ g.t()).t()
    (x * torch.tensor(f["weight0"])).sum().backward()
    np.testing.assert_allclose(f["out0"], x.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(f["grad0"], F.grad.numpy(), atol=1e-6)
    # SDP: X = v v^T, PSD dual = C - lmin I (tests/test_dual_variables.py:523-550), gradient through eigh (tests/test_torch.py:233-248)
    f = np.load(os.path.join(GOLD, "refglue_sdp_sym_primal_psd_dual.npz"))
    C = torch.tensor(f["param0"], requires_grad=True)
    w, V = torch.linalg.eigh(0.5 * (C + C.transpose(1, 2)))
    X = V[:, :, 0, None] * V[:, None, :, 0]
    Z = 0.5 * (C + C.transpose(1, 2)) - w[:, 0, None, None] * torch.eye(C.shape[1], dtype=torch.float64)
    ((X * torch.tensor(f["weight0"])).sum() + (Z * torch.tensor(f["weight1"])).sum()).backward()
    np.testing.assert_allclose(f["out0"], X.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(f["out1"], Z.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(f["grad0"], C.grad.numpy(), atol=1e-5)


def test_build_diffcp_matrices_is_what_the_restated_conventions_say(ns):
    """diffcp_if.py:46-70 executed on seeded boundary values vs cvxpylayers_amd.problems.ConeTemplate.dense_from_values (the
    convention every oracle comparison in this repository goes through): A = -A_aug[:, :-1], b = A_aug[:, -1], c = q[:-1]."""
    from cvxpylayers_amd import problems as P
    rng = np.random.default_rng(0)
    n, cones = 7, {"z": 2, "l": 3, "q": [4]}
    pattern = rng.random((9, n)) < 0.6; pattern[0, :] = True
    bpat = rng.random(9) < 0.7; bpat[0] = True
    tpl = P.dense_template(n, cones, pattern=pattern, b_pattern=bpat)
    B = 3
    A_eval = rng.standard_normal((tpl.nnz_aug, B)); q_eval = rng.standard_normal((n + 1, B))
    As, bs, cs, b_idxs = ns.diffcp_if._build_diffcp_matrices(torch.tensor(A_eval), torch.tensor(q_eval), (tpl.indices, tpl.indptr), (tpl.m, n + 1), tpl.b_idx, B)
    A, b, c = tpl.dense_from_values(A_eval, q_eval)
    for i in range(B):
        np.testing.assert_array_equal(As[i].toarray(), A[i]); np.testing.assert_array_equal(bs[i], b[i]); np.testing.assert_array_equal(cs[i], c[i])
        np.testing.assert_array_equal(b_idxs[i], tpl.b_idx)
    # _compute_gradients (:73-96) with a known adjoint: dA_eval = [-dA.data (CSC order of A), db[b_idx]], dq_eval = [dc, 0]
    dA_d = rng.standard_normal((B, tpl.m, n)); db = rng.standard_normal((B, tpl.m)); dc = rng.standard_normal((B, n))

    def adj(dxs, dys, dss):
        assert all(np.all(d == 0) for d in dss)                       # ds = 0 (diffcp_if.py:84)
        out = []
        for i in range(B):
            cols = np.repeat(np.arange(n), np.diff(As[i].indptr))
            out.append(type(As[i])((dA_d[i][As[i].indices, cols], As[i].indices, As[i].indptr), shape=As[i].shape))
        return out, list(db), list(dc)
    dq_b, dA_b = ns.diffcp_if._compute_gradients(adj, np.zeros((B, n)), np.zeros((B, tpl.m)), bs, b_idxs, B)
    cols_aug = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    for i in range(B):
        aug = np.concatenate([-dA_d[i], db[i][:, None]], axis=1)       # d/d[A_cvx | b_cvx] with A = -A_cvx
        np.testing.assert_allclose(dA_b[i], aug[tpl.indices, cols_aug])
        np.testing.assert_allclose(dq_b[i], np.concatenate([dc[i], [0.0]]))


def _repo_layer(template):
    from cvxpylayers_amd.torch import CvxpyLayer
    return CvxpyLayer(template=template)


@pytest.mark.parametrize("name", ["ridge_mixed", "ridge_batched_matrix_param", "matrix_variable", "metric_shape"])
def test_flattening_and_maps_agree_with_the_reference_frontend(ns, name):
    """reference: p_stack = _flatten_and_batch_params (Fortran, canonical column order, (Ptot+1, B)); A_eval = A_map @ p_stack.
    repository: p_bm (B, Ptot+1) row-major against maps whose columns were re-indexed once.  Same A_eval / q_eval."""
    case = ref_cases.CASES[name]()
    tpl = case["template"]
    rl = ref_glue.reference_layer(ns, tpl)
    params = [torch.tensor(np.asarray(p)) for p in case["params"]]
    batch = rl.ctx.validate_params(list(params))
    p_stack = ns.cvxpylayer._flatten_and_batch_params(tuple(params), rl.ctx, batch).numpy()
    mine = _repo_layer(tpl)
    assert mine.validate_params(list(params)) == batch and mine.batch_sizes == rl.ctx.batch_sizes
    p_bm = mine._flatten_params(params, batch).numpy()
    np.testing.assert_allclose(mine._A.mat @ p_bm.T, tpl.A_map @ p_stack, rtol=0, atol=1e-13)
    np.testing.assert_allclose(mine._q.mat @ p_bm.T, tpl.q_map @ p_stack, rtol=0, atol=1e-13)
    f = np.load(os.path.join(GOLD, f"refglue_{name}.npz"))               # and these are the tensors the reference plugin received
    np.testing.assert_allclose(tpl.A_map @ p_stack, f["A_eval"], atol=1e-13)


@pytest.mark.parametrize("name", ["sdp_sym_primal_psd_dual", "matrix_variable", "metric_shape", "ridge_unbatched"])
def test_variable_recovery_agrees_with_the_reference(ns, name):
    case = ref_cases.CASES[name]()
    tpl = case["template"]
    rl = ref_glue.reference_layer(ns, tpl)
    mine = _repo_layer(tpl)
    rng = np.random.default_rng(1)
    n = tpl.A_structure[2][1] - 1; m = tpl.A_structure[2][0]
    for batch in ((), (4,)):
        B = batch[0] if batch else 1
        primal = torch.tensor(rng.standard_normal((B, n))); dual = torch.tensor(rng.standard_normal((B, m)))
        want = ns.cvxpylayer._recover_results(primal, dual, rl.ctx, batch)
        got = mine._recover_results_torch(primal, dual, batch)       # the per-variable chain (checker of the one-launch device path)
        assert len(want) == len(got)
        for a, b in zip(want, got):
            assert a.shape == b.shape and torch.equal(a, b)
        # the sparse recovery map the device path launches (ce_parammap_apply2), applied here with scipy: same numbers, same layout
        for source, src in (("primal", primal), ("dual", dual)):
            if source not in mine._rec:
                continue
            csr, layout = mine._rec[source]
            rec = (csr.mat @ src.numpy().T).T
            for pos, off, size in layout:
                w = want[pos].numpy()
                assert np.array_equal(rec[:, off:off + size].reshape(w.shape), w)


def test_validate_params_messages_are_the_references(ns):
    tpl = ref_cases.CASES["ridge_mixed"]()["template"]
    rl = ref_glue.reference_layer(ns, tpl); mine = _repo_layer(tpl)
    mF, n = tpl.param_shapes[0]
    bad = [[torch.zeros(mF, n)], [torch.zeros(mF, n), torch.zeros(mF + 1)], [torch.zeros(2, 2, mF, n), torch.zeros(mF)],
           [torch.zeros(2, mF, n), torch.zeros(3, mF)], [torch.zeros(2, mF, n + 1), torch.zeros(2, mF)]]
    for vals in bad:
        with pytest.raises(ValueError) as e_ref:
            rl.ctx.validate_params(list(vals))
        with pytest.raises(ValueError) as e_mine:
            mine.validate_params(list(vals))
        norm = lambda s: re.sub(r"torch\.Size\(\[([^\]]*)\]\)", r"(\1)", str(s)).replace(",)", ")")
        assert norm(e_ref.value) == norm(e_mine.value)


def test_solver_args_reach_diffcp_merged_and_unmutated(ns):
    """diffcp_if.py:356-359: merged = {**ctx.options, **solver_args}; neither dict is mutated (tests/test_parse_args.py:224-248)."""
    case = ref_cases.CASES["ridge_unbatched"]()
    layer_opts = {"eps": 1e-6, "max_iters": 1234}
    rl = ref_glue.reference_layer(ns, case["template"], solver_args=layer_opts)
    call = {"eps": 1e-9}
    ns.diffcp.calls.clear()
    rl(*[torch.tensor(np.asarray(p)) for p in case["params"]], solver_args=call)
    name, kw = ns.diffcp.calls[-1]
    assert name == "solve_only_batch" and kw["eps"] == 1e-9 and kw["max_iters"] == 1234          # no grad needed -> solve_only_batch
    assert layer_opts == {"eps": 1e-6, "max_iters": 1234} and call == {"eps": 1e-9}
    # the repository's plugin merges the same way
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    s = make_settings({**layer_opts, **call})
    assert s.eps_abs == 1e-9 and s.max_iters == 1234


def test_integration_hunks_apply_to_the_reference_sources_and_route_to_the_plugin(ns):
    """the hunks (tests/integration_hunks.py holds them in machine-applicable form, INTEGRATION.md as a
    diff) are applied to the reference's source text in memory; the patched registry returns this repository's context / plugin
    for solver="MI355", and the patched reference frontend reaches the plugin (which refuses a CPU-only host, loudly)."""
    import integration_hunks as ih
    with ih.patched_reference_modules(ns) as patched:
        from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
        assert patched.interfaces.get_torch_cvxpylayer("MI355") is _CvxpyLayer
        assert patched.interfaces.get_torch_cvxpylayer("DIFFCP") is ns.diffcp_if._CvxpyLayer
        case = ref_cases.CASES["ridge_unbatched"]()
        layer = ref_glue.reference_layer(patched, case["template"], solver="MI355", solver_args={"eps": 1e-8})
        assert isinstance(layer.ctx.solver_ctx, MI355_ctx) and layer.ctx.solver_ctx.options == {"eps": 1e-8}
        params = [torch.tensor(np.asarray(p), requires_grad=True) for p in case["params"]]
        if torch.cuda.is_available():
            pytest.skip("GPU present: covered by the -m gpu tests")
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            layer(*params)
        # warm-start hunk: the reference frontend refuses warm_start for every solver but MOREAU (torch/cvxpylayer.py:425-429);
        # patched, "MI355" is let through and the plugin is asked to reuse its previous solution
        with pytest.raises(ValueError, match="only supported"):
            ref_glue.reference_layer(ns, case["template"])(*params, warm_start=True)
        seen = {}

        class Spy:
            @staticmethod
            def apply(P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad, warm_start):
                seen["ws"] = warm_start
                raise KeyboardInterrupt
        orig = patched.interfaces.get_torch_cvxpylayer
        patched.interfaces.get_torch_cvxpylayer = lambda s: Spy
        try:
            with pytest.raises(KeyboardInterrupt):
                layer(*params, warm_start=True)
        finally:
            patched.interfaces.get_torch_cvxpylayer = orig
        assert seen["ws"] is True
