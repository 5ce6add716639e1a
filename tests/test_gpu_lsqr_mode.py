"""solver_args mode="lsqr" on per-instance-A templates: diffcp's own adjoint method (diffcp_if.py:86 -> adj_batch, default mode "lsqr") instead of the engine's
direct elimination.  ce_vjp_lsqr runs Paige & Saunders' LSQR on the full (n + m + 1) system M^T r = dz, one workgroup per instance, with the instance's own A.
Checked against the oracle's LSQR mode (oracle/cone_oracle.c lsqr_MT) at matched tolerances:
  * a regular system: LSQR (tight) = the direct elimination = the oracle, to 1e-6; at diffcp's 1e-8 rule the iteration counts are the oracle's;
  * a RANK-DEFICIENT system (a duplicated equality row): LSQR returns the minimum-norm solution -- db equal on the two copies, as diffcp does -- where the direct
    elimination returns a basic one; the engine's LSQR still matches the oracle's to 1e-6;
  * through the plugin: `_CvxpyLayer.apply(..., solver_args={"mode": "lsqr"})` reaches it, silently (no "ignored" warning);
  * the DEFAULT path (round 6): the elimination kernel lists the instances whose system it found rank deficient and the LSQR kernel re-solves exactly those behind it
    on the device (ce_vjp with q_vals) -- default solver_args give diffcp's minimum-norm element on the degenerate instances, the regular instances of the same
    batch keep the elimination's answer bit for bit, and info["adjoint"] says which is which."""
import warnings

import numpy as np
import pytest
import torch

from cvxpylayers_amd import problems as P
from kit import TIGHT_LSQR
from test_gpu_parity import gpu_solve

pytestmark = pytest.mark.gpu


def _want(tpl, g, n):
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    B = g["dA"].shape[0]
    want = np.empty((tpl.nnz_aug, B))
    for k in range(tpl.nnz_aug):
        i, j = tpl.indices[k], cols[k]
        want[k] = -g["dA"][:, i, j] if j < n else g["db"][:, i]          # boundary convention (diffcp_if.py:91-92)
    return want


def _setup(n, cones, B, seed, eps, dup_rows=None, max_iters=200000):
    from oracle import oracle
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=seed)
    if dup_rows is not None:
        src, dst = dup_rows
        A[:, dst, :] = A[:, src, :]; b[:, dst] = b[:, src]
    ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=max_iters)
    assert (ref["status"] == 1).all()
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, A, b, c, eps=eps, max_iters=max_iters)
    assert (status == 1).all()
    rng = np.random.default_rng(seed + 100)
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    pt = tuple(torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))          # differentiate at the oracle's point
    _, q_eval = tpl.values_from_dense(A, b, c)
    return oracle, tpl, (A, b, c), ref, eng, A_bm, pt, dx, dy, torch.from_numpy(q_eval).cuda()


def test_lsqr_mode_equals_direct_and_oracle_on_a_regular_system():
    cfg = P.CONFIGS["M"]; n, cones = cfg["n"], cfg["cones"]
    oracle, tpl, (A, b, c), ref, eng, A_bm, (xr, yr, sr), dx, dy, q_t = _setup(n, cones, 48, 0, 1e-9)
    dxt, dyt = torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda()
    dA_d, dq_d, _ = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance")
    dA_l, dq_l, adj = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_lsqr", lsqr=TIGHT_LSQR, q_eval=q_t)
    torch.cuda.synchronize()
    assert (adj.cpu().numpy() == 0).all()
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    want = _want(tpl, g, n)
    scale = 1 + np.abs(want).max()
    assert np.abs(dA_l.cpu().numpy() - want).max() < 1e-6 * scale
    assert np.abs(dA_l.cpu().numpy() - dA_d.cpu().numpy()).max() < 1e-5 * scale          # regular system: one solution
    assert np.abs(dq_l.cpu().numpy()[:n] - g["dc"].T).max() < 1e-6 * (1 + np.abs(g["dc"]).max())
    # diffcp's stopping rule (1e-8 / 1e-8 / 1e8 / 2 N): the iteration counts are the oracle's
    dA_r, dq_r, adj_r = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_lsqr", q_eval=q_t)
    torch.cuda.synchronize()
    g_r = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr")
    it_e, it_o = eng.last_lsqr_iters.cpu().numpy(), g_r["lsqr_iters"]
    assert np.abs(it_e - it_o).max() <= 3 + 0.05 * it_o.max(), (it_e, it_o)
    assert np.abs(dA_r.cpu().numpy() - _want(tpl, g_r, n)).max() < 1e-5 * scale


def test_lsqr_mode_returns_diffcps_minimum_norm_solution_on_a_rank_deficient_system():
    n, cones = 8, {"z": 4, "l": 6, "q": [4]}
    oracle, tpl, (A, b, c), ref, eng, A_bm, (xr, yr, sr), dx, dy, q_t = _setup(n, cones, 6, 2, 1e-10, dup_rows=(0, 1))
    dxt, dyt = torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda()
    dA_l, dq_l, adj = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_lsqr", lsqr=TIGHT_LSQR, q_eval=q_t)
    dA_d, dq_d, _ = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance")
    torch.cuda.synchronize()
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    want = _want(tpl, g, n)
    scale = 1 + np.abs(want).max()
    got = dA_l.cpu().numpy()
    assert np.abs(got - want).max() < 1e-6 * scale, np.abs(got - want).max() / scale
    assert np.abs(dq_l.cpu().numpy()[:n] - g["dc"].T).max() < 1e-6 * (1 + np.abs(g["dc"]).max())
    # the two copies of the equality receive the same db from the minimum-norm solution (b entries: the last column of the value order) ...
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    kb = {int(tpl.indices[k]): k for k in range(tpl.nnz_aug) if cols[k] == n}
    assert np.abs(got[kb[0]] - got[kb[1]]).max() < 1e-8 * scale
    # ... and not from the direct elimination's basic solution: the two methods differ here, which is why the mode exists
    direct = dA_d.cpu().numpy()
    assert np.abs(direct[kb[0]] - direct[kb[1]]).max() > 1e-3
    assert np.isfinite(direct).all()


def test_mode_lsqr_through_the_plugin_is_silent_and_reaches_the_lsqr_kernel():
    from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
    n, cones = 8, {"z": 4, "l": 6, "q": [4]}
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, 6, seed=2)
    A[:, 1, :] = A[:, 0, :]; b[:, 1] = b[:, 0]
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-10, "max_iters": 200000})
    grads = {}
    for mode in ("lsqr", "dense"):
        A_t = torch.from_numpy(A_eval).cuda().requires_grad_(); q_t = torch.from_numpy(q_eval).cuda().requires_grad_()
        with warnings.catch_warnings():
            warnings.simplefilter("error")          # an "accepted and ignored" warning would fail the call
            primal, dual, info, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, {"mode": mode, "lsqr_atol": 1e-12, "lsqr_btol": 1e-12, "lsqr_iter_lim": 20000}, True, None)
            (primal * torch.arange(1, n + 1, device="cuda", dtype=torch.float64)).sum().backward()
        grads[mode] = A_t.grad.cpu().numpy()
    eng = ctx.engine(torch.device("cuda", 0))
    assert eng.last_lsqr_iters is not None and int(eng.last_lsqr_iters.max()) > 0
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    kb = {int(tpl.indices[k]): k for k in range(tpl.nnz_aug) if cols[k] == n}
    assert np.abs(grads["lsqr"][kb[0]] - grads["lsqr"][kb[1]]).max() < 1e-7 * (1 + np.abs(grads["lsqr"]).max())
    assert np.abs(grads["dense"][kb[0]] - grads["dense"][kb[1]]).max() > 1e-3


def test_default_path_resolves_exactly_the_rank_deficient_instances_by_lsqr():
    from oracle import oracle
    n, cones, B = 8, {"z": 4, "l": 6, "q": [4]}, 12
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=5)
    deg = np.arange(B) % 3 == 0                      # every third instance gets a duplicated equality row
    A[deg, 1, :] = A[deg, 0, :]; b[deg, 1] = b[deg, 0]
    ref = oracle.solve_batch(A, b, c, cones, eps=1e-10, max_iters=200000)
    assert (ref["status"] == 1).all()
    eng, A_bm, *_ = gpu_solve(tpl, A, b, c, eps=1e-10, max_iters=200000)
    rng = np.random.default_rng(7)
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dxt, dyt = torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda()
    _, q_eval = tpl.values_from_dense(A, b, c); q_t = torch.from_numpy(q_eval).cuda()
    dA_basic, dq_basic, adj_basic = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance")                                  # no q_eval: the elimination alone
    dA_def, dq_def, adj_def = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance", lsqr=TIGHT_LSQR, q_eval=q_t)           # the default route of the plugin
    dA_def2, _, adj_def2 = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance", lsqr=TIGHT_LSQR, q_eval=q_t)              # (the device-side list is reset between calls)
    dA_l, dq_l, _ = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_lsqr", lsqr=TIGHT_LSQR, q_eval=q_t)
    torch.cuda.synchronize()
    a0, a1 = adj_basic.cpu().numpy(), adj_def.cpu().numpy()
    assert ((a0 & 4) != 0).tolist() == deg.tolist(), a0                  # the elimination flags exactly the duplicated-row instances ...
    assert (a1[deg] == 12).all() and (a1[~deg] == 0).all(), a1           # ... and exactly those are re-solved (4 | 8), converged (bit 0 clear)
    assert (adj_def2.cpu().numpy() == a1).all()
    got, basic, lsq = dA_def.cpu().numpy(), dA_basic.cpu().numpy(), dA_l.cpu().numpy()
    sc0 = 1 + np.abs(basic).max()
    assert np.abs(got[:, ~deg] - basic[:, ~deg]).max() < 1e-9 * sc0      # regular instances: the elimination's answer (search-free kernel here, pivoting kernel there: one solution)
    assert np.array_equal(got[:, deg], lsq[:, deg])                      # degenerate instances: the LSQR kernel's answer
    assert np.array_equal(dA_def2.cpu().numpy(), got)
    assert np.array_equal(dq_def.cpu().numpy()[:, deg], dq_l.cpu().numpy()[:, deg])
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    want = _want(tpl, g, n)
    assert np.abs(got - want).max() < 1e-6 * (1 + np.abs(want).max())  # every instance = the oracle's LSQR mode = diffcp's element
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    kb = {int(tpl.indices[k]): k for k in range(tpl.nnz_aug) if cols[k] == n}
    assert np.abs(got[kb[0]][deg] - got[kb[1]][deg]).max() < 1e-8 * (1 + np.abs(want).max())
    assert np.abs(basic[kb[0]][deg] - basic[kb[1]][deg]).max() > 1e-3


def test_default_solver_args_through_the_plugin_give_diffcps_element_and_report_it():
    from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer, adjoint_report
    n, cones = 8, {"z": 4, "l": 6, "q": [4]}
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, 6, seed=2)
    A[:3, 1, :] = A[:3, 0, :]; b[:3, 1] = b[:3, 0]
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-10, "max_iters": 200000})
    grads, reports = {}, {}
    for name, args in (("default", {}), ("lsqr", {"mode": "lsqr"}), ("dense", {"mode": "dense"})):
        A_t = torch.from_numpy(A_eval).cuda().requires_grad_(); q_t = torch.from_numpy(q_eval).cuda().requires_grad_()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            primal, dual, info, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, args, True, None)
            (primal * torch.arange(1, n + 1, device="cuda", dtype=torch.float64)).sum().backward()
        grads[name] = A_t.grad.cpu().numpy(); reports[name] = adjoint_report(info)
    assert reports["default"] == dict(rank_deficient=3, lsqr_resolved=3, lsqr_iteration_limit=0, no_gradient=0, backward_ran=True), reports
    assert reports["dense"]["rank_deficient"] == 3 and reports["dense"]["lsqr_resolved"] == 0
    scale = 1 + np.abs(grads["lsqr"]).max()
    assert np.abs(grads["default"][:, :3] - grads["lsqr"][:, :3]).max() < 1e-12 * scale          # the same kernel, the same rule
    assert np.abs(grads["default"][:, 3:] - grads["dense"][:, 3:]).max() < 1e-9 * scale          # (two elimination kernels, one solution)
    assert np.abs(grads["default"][:, 3:] - grads["lsqr"][:, 3:]).max() < 1e-5 * scale           # regular instances: elimination = LSQR at diffcp's rule
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    kb = {int(tpl.indices[k]): k for k in range(tpl.nnz_aug) if cols[k] == n}
    assert np.abs(grads["default"][kb[0]][:3] - grads["default"][kb[1]][:3]).max() < 1e-6 * scale
    assert np.abs(grads["dense"][kb[0]][:3] - grads["dense"][kb[1]][:3]).max() > 1e-3


def test_lsmr_mode_matches_the_oracles_lsmr_and_the_direct_elimination():
    """solver_args mode="lsmr" (diffcp's third adjoint mode): k_sa_lsqr<..., LSMR> -- the LSQR kernel's bidiagonalisation under Fong & Saunders' recurrences and stopping
    tests (ce_set_lsqr_variant) -- against the oracle's LSMR (oracle/cone_oracle.c lsmr_core, pinned on scipy.sparse.linalg.lsmr by the CPU suite): same gradients at a
    tight rule, the elimination's gradients on a regular system, the oracle's iteration counts at diffcp's rule; on a shared-A template too; through the plugin, silently."""
    cfg = P.CONFIGS["M"]; n, cones = cfg["n"], cfg["cones"]
    oracle, tpl, (A, b, c), ref, eng, A_bm, (xr, yr, sr), dx, dy, q_t = _setup(n, cones, 24, 1, 1e-9)
    dxt, dyt = torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda()
    dA_d, dq_d, _ = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance")
    dA_m, dq_m, adj = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_lsqr", lsqr=TIGHT_LSQR + ("full", "lsmr"), q_eval=q_t)
    torch.cuda.synchronize()
    assert (adj.cpu().numpy() == 0).all()
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsmr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    want = _want(tpl, g, n)
    scale = 1 + np.abs(want).max()
    assert np.abs(dA_m.cpu().numpy() - want).max() < 1e-6 * scale
    assert np.abs(dA_m.cpu().numpy() - dA_d.cpu().numpy()).max() < 1e-5 * scale          # regular system: one solution
    assert np.abs(dq_m.cpu().numpy()[:n] - g["dc"].T).max() < 1e-6 * (1 + np.abs(g["dc"]).max())
    # diffcp's tolerances: the iteration counts are the oracle's LSMR's (and not LSQR's: the two stop at different iterations)
    dA_r, dq_r, adj_r = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_lsqr", lsqr=(1e-8, 1e-8, 2 * (tpl.n + tpl.m + 1), "full", "lsmr"), q_eval=q_t)
    torch.cuda.synchronize()
    g_r = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsmr")
    it_e, it_o = eng.last_lsqr_iters.cpu().numpy(), g_r["lsqr_iters"]
    assert np.abs(it_e - it_o).max() <= 3 + 0.05 * it_o.max(), (it_e, it_o)
    assert np.abs(dA_r.cpu().numpy() - _want(tpl, g_r, n)).max() < 1e-5 * scale
    # the variant is per call: the next LSQR call is LSQR again
    dA_l, _, _ = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_lsqr", q_eval=q_t)
    g_l = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr")
    assert np.abs(eng.last_lsqr_iters.cpu().numpy() - g_l["lsqr_iters"]).max() <= 3 + 0.05 * g_l["lsqr_iters"].max()


def test_mode_lsmr_through_the_plugin_is_silent_and_differs_from_nothing_but_the_solver():
    from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 16
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=5)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    outs = {}
    for mode in ("lsqr", "lsmr"):
        ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-9, "max_iters": 100000, "mode": mode, "lsqr_atol": 1e-12, "lsqr_btol": 1e-12, "lsqr_iter_lim": 20000})
        A_t = torch.from_numpy(A_eval).cuda().requires_grad_(); q_t = torch.from_numpy(q_eval).cuda().requires_grad_()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            p, d, info, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, {}, True, None)
            (p * torch.arange(1, p.numel() + 1, device=p.device, dtype=p.dtype).reshape(p.shape) / p.numel()).sum().backward()
        outs[mode] = (A_t.grad.cpu().numpy().copy(), q_t.grad.cpu().numpy().copy())
    for a, b_ in zip(outs["lsqr"], outs["lsmr"]):
        assert np.abs(a - b_).max() < 1e-6 * (1 + np.abs(a).max())          # regular systems: both converge to the one solution
        assert np.abs(a - b_).max() > 0                                      # ... along different iterates


def test_lsmr_on_a_shared_A_template_matches_the_oracle(monkeypatch):
    """the shared-A adjoint (ce_vjp_shared_a: the split products of k_sa_lsqr<RP>) under LSMR, against the oracle's LSMR at a tight rule and at diffcp's tolerances"""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    B = 6
    A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=3, nw=60, kf=9)
    Ab = np.broadcast_to(A, (B,) + A.shape).copy(); bb = np.broadcast_to(b, (B,) + b.shape).copy()
    ref = oracle.solve_batch(Ab, bb, c, cones, eps=1e-8, max_iters=200000)
    assert (ref["status"] == 1).all()
    monkeypatch.setenv("CE_CONST_A", "1")
    A_eval, q_eval = tpl.values_from_dense(Ab, bb, c)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dx = np.random.default_rng(2).standard_normal((B, tpl.n)); dy = np.zeros_like(ref["y"])
    dA, dq, adj = eng.vjp(A_bm, xo, yo, so, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), path="const_a", lsqr=TIGHT_LSQR + ("full", "lsmr"), q_eval=q_t)
    torch.cuda.synchronize()
    g = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsmr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    want = _want(tpl, g, tpl.n)
    assert np.abs(dA.cpu().numpy() - want).max() < 1e-6 * (1 + np.abs(want).max())
    assert np.abs(dq.cpu().numpy()[:tpl.n] - g["dc"].T).max() < 1e-6 * (1 + np.abs(g["dc"]).max())
    # iteration counts at diffcp's tolerances (at 1e-12 both sides stop on rounding-level quantities, a few dozen iterations apart)
    dA2, dq2, adj2 = eng.vjp(A_bm, xo, yo, so, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), path="const_a", lsqr=(1e-8, 1e-8, 2 * (tpl.n + tpl.m + 1), "full", "lsmr"), q_eval=q_t)
    torch.cuda.synchronize()
    g2 = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsmr")
    it_e, it_o = eng.last_lsqr_iters.cpu().numpy(), g2["lsqr_iters"]
    assert np.abs(it_e - it_o).max() <= 3 + 0.12 * it_o.max(), (it_e, it_o)          # (this template: the engine's LSQR and LSMR both stop 5-10 % before the oracle's at 1e-8: scripts/probes/lsmr_counts_check.py)
    # the recurrence itself: after THREE iterations (no stopping test involved) the iterates are the oracle's to rounding
    from cvxpylayers_amd.interfaces.const_a import vjp_const_a
    from cvxpylayers_amd import _lib
    _lib.lib().ce_set_lsqr_variant(eng._h, 1)
    try:
        dA3, dq3, _ = vjp_const_a(eng, A_bm, xo, yo, so, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), atol=0.0, btol=0.0, iter_lim=3, q_eval=q_t, conlim=0.0)
    finally:
        _lib.lib().ce_set_lsqr_variant(eng._h, 0)
    g3 = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsmr", lsqr_atol=0.0, lsqr_btol=0.0, lsqr_iter_lim=3, lsqr_conlim=0.0)
    assert (eng.last_lsqr_iters.cpu().numpy() == 3).all() and (g3["lsqr_iters"] == 3).all()
    assert np.abs(dq3.cpu().numpy()[:tpl.n] - g3["dc"].T).max() < 1e-12 * (1 + np.abs(g3["dc"]).max())
