"""GPU parity tests proper: the HIP engine (through the C ABI, via the plugin's ConeEngine) against the CPU
oracle on the same seeded inputs.  Tolerances (fp64): solutions within 1e-6*(1+|x|_inf) at eps=1e-8
(BASELINE.md parity gate); gradients within 1e-5 relative to the oracle's dense adjoint."""
import os

import numpy as np
import pytest
import torch

import kit
from cvxpylayers_amd import problems as P
from kit import TIGHT_LSQR, lsqr_own_movement, assert_lsqr_agreement_per_instance

pytestmark = pytest.mark.gpu


def _engine_for(tpl):
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine
    return ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, tpl.cones, torch.device("cuda", 0))


def gpu_solve(tpl, A, b, c, batch_minor=True, **args):
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    eng = _engine_for(tpl)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_t = torch.from_numpy(A_eval).cuda()
    q_t = torch.from_numpy(q_eval).cuda()
    if not batch_minor:
        A_t = A_t.t().contiguous().t()
    A_bm = eng.to_batch_major(A_t)
    args.setdefault("acceleration_lookback", 0)      # the oracle's default: plain iteration (the accelerated runs pass lookback 1 to both)
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(args))
    torch.cuda.synchronize()
    return eng, A_bm, x, y, s, iters.cpu().numpy(), status.cpu().numpy(), resid.cpu().numpy()


def run_parity(n, cones, B, seed, eps, pattern=None, max_iters=20000):
    from oracle import oracle
    tpl = P.dense_template(n, cones, pattern=pattern)
    A, b, c = P.generate(n, cones, B, seed=seed, pattern=pattern)
    ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=max_iters)
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, A, b, c, eps=eps, max_iters=max_iters)
    assert (ref["status"] == 1).all(), "test family must converge so that gradients are defined"
    assert (status == ref["status"]).all(), (status, ref["status"])
    xs, ys, ss = x.cpu().numpy(), y.cpu().numpy(), s.cpu().numpy()
    tol = max(1e-6, 20 * eps)
    for got, want in ((xs, ref["x"]), (ys, ref["y"]), (ss, ref["s"])):
        err = np.abs(got - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
        assert err.max() < tol, err.max()
    # same algorithm, same arithmetic up to summation order: iteration counts agree (a borderline check may shift by one interval)
    assert np.abs(iters - ref["iters"]).max() <= 25, (iters, ref["iters"])
    # backward
    rng = np.random.default_rng(seed + 100)
    dx = rng.standard_normal(xs.shape); dy = rng.standard_normal(ys.shape)
    # At a tightly converged point diffcp's dense and LSQR modes agree and the dense one is exact: compare to 1e-5.
    # At a loosely converged point (eps >= 1e-6) M is only nearly singular and the dense elimination amplifies the
    # inconsistency (errors of 0.1-0.6 against the true gradient), so the reference-default LSQR
    # mode is the comparator there, at the accuracy LSQR itself has.
    tight = eps <= 1e-7
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="dense" if tight else "lsqr")
    gtol = 1e-5 if tight else 2e-3
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))   # differentiate at the oracle's point
    dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), lsqr=TIGHT_LSQR)
    torch.cuda.synchronize()
    assert (adj.cpu().numpy() == 0).all()
    dA = dA.cpu().numpy(); dq = dq.cpu().numpy()
    # boundary convention: dA_eval = [-dA.data, db[b_idx]], dq_eval = [dc, 0]  (diffcp_if.py:91-92)
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    want = np.empty_like(dA)
    for k in range(tpl.nnz_aug):
        i, j = tpl.indices[k], cols[k]
        want[k] = -g["dA"][:, i, j] if j < n else g["db"][:, i]
    scale = 1 + np.abs(want).max()
    assert np.abs(dA - want).max() < gtol * scale, np.abs(dA - want).max() / scale
    assert np.abs(dq[:n] - g["dc"].T).max() < gtol * (1 + np.abs(g["dc"]).max())
    assert np.abs(dq[n]).max() == 0
    return eng


@pytest.mark.parametrize("eps", [1e-4, 1e-8])
def test_metric_config_parity(eps):
    cfg = P.CONFIGS["M"]
    run_parity(cfg["n"], cfg["cones"], 64, seed=0, eps=eps)


def test_nonneg_only_parity():
    # pure LPs converge slowly under operator splitting (tens of thousands of iterations): small case, many iterations
    run_parity(20, {"z": 0, "l": 40, "q": []}, 8, seed=1, eps=1e-8, max_iters=400000)


def test_mixed_zero_nonneg_soc_small():
    run_parity(10, {"z": 3, "l": 8, "q": [5, 4]}, 16, seed=3, eps=1e-9)


def test_sparse_pattern_and_ragged_soc():
    cones = {"z": 2, "l": 5, "q": [3, 1, 6]}
    rng = np.random.default_rng(7)
    m = P.cone_rows(cones)
    pattern = rng.random((m, 12)) < 0.6
    pattern[np.arange(m), rng.integers(0, 12, m)] = True
    pattern[rng.integers(0, m, 12), np.arange(12)] = True
    run_parity(12, cones, 8, seed=4, eps=1e-9, pattern=pattern)


def test_socp_c3_global_residency():
    cfg = P.CONFIGS["C3"]
    run_parity(cfg["n"], cfg["cones"], 8, seed=2, eps=1e-8)


def test_large_per_instance_template_all_global_residency():
    """n = 120, m = 186 with per-instance A: neither A nor G fits LDS (size-generic kernels, residency mode 2): the coalesced
    global-memory products, the MFMA formation of S from the global A and the blocked Gauss-Jordan, against the oracle (solution
    and adjoint)."""
    cones = {"z": 0, "l": 60, "q": [21] * 6}
    eng = run_parity(120, cones, 6, seed=7, eps=1e-8)
    info = eng.launch_info()
    assert info["fwd_mode"] == 2 and info["bwd_mode"] == 2, info


def test_layouts_agree():
    cfg = P.CONFIGS["M"]
    tpl = P.dense_template(cfg["n"], cfg["cones"])
    A, b, c = P.generate(cfg["n"], cfg["cones"], 8, seed=5)
    r1 = gpu_solve(tpl, A, b, c, batch_minor=True, eps=1e-6)
    r2 = gpu_solve(tpl, A, b, c, batch_minor=False, eps=1e-6)
    assert torch.equal(r1[2], r2[2]) and torch.equal(r1[3], r2[3])


def test_known_answers_on_gpu():
    for (A, b, c, cones, xstar), nx in ((kit.box_qp(np.array([2.0, 0.5, -1.0])), 3),
                                        (kit.simplex_lp(np.array([1.0, 2.0])), 2),
                                        (kit.soc_lin(np.array([1.0, 0.5, -0.5]), 2.0), 3)):
        tpl = P.dense_template(A.shape[1], cones)
        *_, x, y, s, iters, status, resid = gpu_solve(tpl, A[None], b[None], c[None], eps=1e-10)
        assert status[0] == 1
        np.testing.assert_allclose(x.cpu().numpy()[0, :nx], xstar, atol=1e-5)


def test_infeasible_unbounded_status_on_gpu():
    for builder, want in ((kit.infeasible, -2), (kit.unbounded, -1)):
        A, b, c, cones = builder()
        tpl = P.dense_template(A.shape[1], cones)
        *_, iters, status, resid = gpu_solve(tpl, A[None], b[None], c[None], eps=1e-6)
        assert status[0] == want


def _forward_parity(n, cones, B, seed, eps, max_iters=100000):
    from oracle import oracle
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=seed)
    ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=max_iters)
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, A, b, c, eps=eps, max_iters=max_iters)
    assert (ref["status"] == 1).all() and (status == 1).all(), (status, ref["status"])
    tol = max(1e-6, 20 * eps)
    for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
        err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
        assert err.max() < tol, err.max()
    return eng, A_bm, ref


def test_psd_forward_parity_small():
    _forward_parity(6, {"z": 2, "l": 3, "q": [4], "s": [3]}, 8, seed=5, eps=1e-9)


def test_psd_forward_parity_two_cones_odd_order():
    _forward_parity(12, {"z": 1, "l": 2, "q": [], "s": [5, 4]}, 6, seed=8, eps=1e-9)


def test_psd_forward_parity_c4_lite():
    _forward_parity(36, {"z": 6, "l": 0, "q": [], "s": [8]}, 4, seed=6, eps=1e-9)


def test_sdp_min_eigenvector_known_answer_on_gpu():
    # min tr(C X) s.t. tr X = 1, X PSD -> X = v v^T, dual = C - lmin I  (reference tests/test_dual_variables.py:523-550)
    Cm = np.array([[1.0, 0.5], [0.5, 2.0]])
    A, b, c, cones, X, Z = kit.sdp_min_eig(Cm)
    tpl = P.dense_template(A.shape[1], cones)
    *_, x, y, s, iters, status, resid = gpu_solve(tpl, A[None], b[None], c[None], eps=1e-10, max_iters=100000)
    assert status[0] == 1
    np.testing.assert_allclose(P.svec_to_sym(x.cpu().numpy()[0], 2), X, atol=1e-5)
    np.testing.assert_allclose(P.svec_to_sym(y.cpu().numpy()[0][1:], 2), Z, atol=1e-5)


def test_psd_forward_and_adjoint_parity():
    run_parity(6, {"z": 2, "l": 3, "q": [4], "s": [3]}, 8, seed=5, eps=1e-9, max_iters=100000)


def test_psd_two_cones_adjoint_parity():
    run_parity(12, {"z": 1, "l": 2, "q": [], "s": [5, 4]}, 6, seed=8, eps=1e-9, max_iters=100000)


def test_constant_A_gemm_path_matches_oracle(monkeypatch):
    """A batch-invariant (only b, c vary): the batch-GEMM forward (interfaces/const_a.py) against the oracle, forced on a small
    mixed zero / nonneg / SOC instance; same iteration counts up to one check interval."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    monkeypatch.setenv("CE_CONST_A", "1")
    n, cones, B = 12, {"z": 2, "l": 6, "q": [4, 5]}, 24
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=21, batched=("b", "c"))
    for eps in (1e-4, 1e-9):
        ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=100000)
        eng = _engine_for(tpl)
        A_eval, q_eval = tpl.values_from_dense(A, b, c)
        A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda())
        x, y, s, iters, status, resid = eng.solve(A_bm, torch.from_numpy(q_eval).cuda(), make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=100000)))
        assert eng.last_path == "const_a"
        assert (status.cpu().numpy() == ref["status"]).all() and (ref["status"] == 1).all()
        tol = max(1e-6, 20 * eps)
        for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
            err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
            assert err.max() < tol, err.max()
        assert np.abs(iters.cpu().numpy() - ref["iters"]).max() <= 25
        if eps < 1e-6:       # batched-LSQR adjoint of the constant-A path against the oracle's dense adjoint
            rng = np.random.default_rng(5)
            dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
            g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="dense")
            xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
            dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), lsqr=TIGHT_LSQR)
            assert (adj.cpu().numpy() == 0).all()
            dA = dA.cpu().numpy(); dq = dq.cpu().numpy()
            cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
            want = np.empty_like(dA)
            for k in range(tpl.nnz_aug):
                i, j = tpl.indices[k], cols[k]
                want[k] = -g["dA"][:, i, j] if j < n else g["db"][:, i]
            assert np.abs(dA - want).max() < 1e-5 * (1 + np.abs(want).max()), np.abs(dA - want).max()
            assert np.abs(dq[:n] - g["dc"].T).max() < 1e-5 * (1 + np.abs(g["dc"]).max())


def test_torch_lsqr_fallback_solves_diffcps_full_system_like_the_kernel(monkeypatch):
    """the batched torch LSQR of const_a.py (taken when the one-kernel LSQR does not apply) used to drop the tau row / column and the conlim and
    machine-precision stopping tests, so the gradient depended on which path ran.  Both paths now solve diffcp's full (n + m + 1) system under the same rule:
    the fallback (CE_SA_KERNEL=0) equals the kernel and the oracle's LSQR mode, and a degenerate instance (duplicated equality row: rank-deficient system)
    gets the same minimum-norm element from all three."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    monkeypatch.setenv("CE_CONST_A", "1")
    n, cones, B = 12, {"z": 3, "l": 6, "q": [4, 5]}, 16
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=23, batched=("b", "c"))
    A[:, 1, :] = A[:, 0, :]; b[:, 1] = b[:, 0]                       # redundant equality: M^T is rank deficient on every instance
    ref = oracle.solve_batch(A, b, c, cones, eps=1e-9, max_iters=200000)
    assert (ref["status"] == 1).all()
    eng = _engine_for(tpl)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda()); q_t = torch.from_numpy(q_eval).cuda()
    rng = np.random.default_rng(6)
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    args = (A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda())
    out = {}
    for name, env in (("kernel", "1"), ("torch", "0")):
        monkeypatch.setenv("CE_SA_KERNEL", env)
        for rule, lsqr in (("tight", TIGHT_LSQR), ("diffcp", None)):
            dA, dq, adj = eng.vjp(*args, path="const_a", lsqr=lsqr, q_eval=q_t)
            torch.cuda.synchronize()
            assert (adj.cpu().numpy() == 0).all()
            out[name, rule] = (dA.cpu().numpy().copy(), dq.cpu().numpy().copy())
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    want = np.empty((tpl.nnz_aug, B))
    for k in range(tpl.nnz_aug):
        i, j = tpl.indices[k], cols[k]
        want[k] = -g["dA"][:, i, j] if j < n else g["db"][:, i]
    sc = 1 + np.abs(want).max()
    for name in ("kernel", "torch"):
        dA, dq = out[name, "tight"]
        assert np.abs(dA - want).max() < 1e-6 * sc, (name, np.abs(dA - want).max() / sc)
        assert np.abs(dq[:n] - g["dc"].T).max() < 1e-6 * (1 + np.abs(g["dc"]).max()), name
    # diffcp's own rule (1e-8 / 1e-8 / conlim 1e8 / 2 N): the two paths stop at the same point of the same recurrence
    assert np.abs(out["torch", "diffcp"][0] - out["kernel", "diffcp"][0]).max() < 1e-7 * sc
    kb = {int(tpl.indices[k]): k for k in range(tpl.nnz_aug) if cols[k] == n}
    assert np.abs(out["torch", "tight"][0][kb[0]] - out["torch", "tight"][0][kb[1]]).max() < 1e-8 * sc          # minimum norm: the two copies share db


def test_constant_A_path_is_selected_for_large_shared_templates():
    """Portfolio-shaped template (1^T w = 1, w >= 0, ||F^T w|| <= t; only the returns vary) too large for the LDS-resident
    kernels: the engine picks the batch-GEMM forward and the batched-LSQR adjoint by itself; both agree with the oracle."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    nw, kf, B = 140, 12, 6
    rng = np.random.default_rng(0)
    F = rng.standard_normal((nw, kf)) / np.sqrt(kf) * 0.3
    n = nw + 1; cones = {"z": 1, "l": nw, "q": [kf + 1]}; m = P.cone_rows(cones)
    A = np.zeros((m, n)); b = np.zeros(m)
    A[0, :nw] = 1.0; b[0] = 1.0
    A[1:1 + nw, :nw] = -np.eye(nw)
    A[1 + nw, nw] = -1.0
    A[2 + nw:, :nw] = -F.T
    tpl = P.dense_template(n, cones, pattern=(A != 0), b_pattern=(b != 0))
    mu = 0.05 + 0.1 * rng.random((B, nw))
    c = np.concatenate([-mu, np.ones((B, 1))], axis=1)
    Ab = np.broadcast_to(A, (B, m, n)).copy(); bb = np.broadcast_to(b, (B, m)).copy()
    eps = 1e-8
    ref = oracle.solve_batch(Ab, bb, c, cones, eps=eps, max_iters=200000)
    assert (ref["status"] == 1).all()
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, Ab, bb, c, eps=eps, max_iters=200000)
    assert eng.last_path == "const_a" and (status == 1).all()
    assert np.abs(x.cpu().numpy() - ref["x"]).max() < 1e-6 and np.abs(y.cpu().numpy() - ref["y"]).max() < 1e-6
    dx = rng.standard_normal(ref["x"].shape)
    # the comparator is the oracle's LSQR mode (diffcp's default and its semantics) at the same stopping rule: both sides return the minimum-norm solution of the
    # full (n + m + 1) adjoint system -- at the vertices of this LP-like program that system is rank deficient and the dense elimination picks another element
    g = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, np.zeros_like(ref["y"]), mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.zeros_like(yr), lsqr=TIGHT_LSQR)
    assert (adj.cpu().numpy() == 0).all()
    el = np.abs(dq.cpu().numpy()[:n].T - g["dc"]).max(axis=1) / (1 + np.abs(g["dc"]).max(axis=1))
    own = lsqr_own_movement(lambda **kw: oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, np.zeros_like(ref["y"]), mode="lsqr", **kw), g,
                            lambda g1, g2: np.abs(g1["dc"] - g2["dc"]).max(axis=1) / (1 + np.abs(g2["dc"]).max(axis=1)))
    assert np.median(el) < 1e-9, el
    assert_lsqr_agreement_per_instance(el, own)          # (rounding-level agreement; on ill-conditioned instances: within the oracle's own movement under a tighter rule)


def test_constant_A_path_with_psd_cone(monkeypatch):
    """PSD blocks on the batch-GEMM path (projection and its derivative by batched symmetric eigendecompositions)."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    monkeypatch.setenv("CE_CONST_A", "1")
    n, cones, B = 10, {"z": 2, "l": 3, "q": [4], "s": [4]}, 12
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=31, batched=("b", "c"))
    eps = 1e-9
    ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=200000)
    assert (ref["status"] == 1).all()
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, A, b, c, eps=eps, max_iters=200000)
    assert eng.last_path == "const_a" and (status == 1).all()
    for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
        assert np.abs(got.cpu().numpy() - want).max() < 1e-6 * (1 + np.abs(want).max())
    assert np.abs(iters - ref["iters"]).max() <= 25
    rng = np.random.default_rng(2)
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="dense")
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), lsqr=TIGHT_LSQR)
    assert (adj.cpu().numpy() == 0).all()
    assert np.abs(dq.cpu().numpy()[:n] - g["dc"].T).max() < 1e-5 * (1 + np.abs(g["dc"]).max())


def test_box_qp_config2_epigraph_form():
    """BASELINE config 2 (box-constrained QP, n=50) in the SOC-epigraph form DIFFCP canonicalises it to; the optimum is also
    checked against projected gradient descent on the QP itself."""
    from oracle import oracle
    A, b, c, cones = P.box_qp_batch(50, 12, seed=3)
    n = A.shape[2]
    tpl = P.dense_template(n, cones, pattern=(A[0] != 0))
    ref = oracle.solve_batch(A, b, c, cones, eps=1e-9, max_iters=100000)
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, A, b, c, eps=1e-9, max_iters=100000)
    assert (status == 1).all() and (ref["status"] == 1).all()
    assert np.abs(x.cpu().numpy() - ref["x"]).max() < 1e-6 and np.abs(iters - ref["iters"]).max() <= 25
    F = -A[0, 102:, :50] / 2.0; g = -b[:, 102:] / 2.0; lo = -b[:, :50]; hi = b[:, 50:100]
    xq = np.clip(np.zeros((12, 50)), lo, hi)
    Lc = np.linalg.norm(F, 2) ** 2
    for _ in range(20000):
        xq = np.clip(xq - (xq @ F.T - g) @ F / Lc, lo, hi)
    assert np.abs(x.cpu().numpy()[:, :50] - xq).max() < 1e-5


# ------------------------------------------------------------------ exponential cones (SCS row order z,l,q,s,ep)
def test_exp_cone_forward_and_adjoint_parity():
    run_parity(8, {"z": 2, "l": 4, "q": [4], "s": [], "ep": 3}, 16, seed=1, eps=1e-9, max_iters=200000)


def test_exp_cone_with_psd_block_parity():
    run_parity(9, {"z": 1, "l": 2, "q": [3], "s": [3], "ep": 2}, 8, seed=4, eps=1e-9, max_iters=200000)


def test_entropy_and_logistic_known_answers_on_gpu():
    # entropy maximisation -> uniform distribution; logistic regression against a smooth solver (tests/test_torch.py:158-230 layer)
    from scipy.optimize import minimize
    A, b, c, cones, xstar = kit.entropy_max(6)
    tpl = P.dense_template(A.shape[1], cones)
    *_, x, y, s, iters, status, resid = gpu_solve(tpl, A[None], b[None], c[None], eps=1e-10, max_iters=100000)
    assert status[0] == 1
    np.testing.assert_allclose(x.cpu().numpy()[0][:6], xstar, atol=1e-6)
    rng = np.random.default_rng(1)
    N, d, lam = 12, 3, 0.5
    X = rng.standard_normal((N, d)); lab = np.sign(X @ np.array([1.0, -2.0, 0.5]) + 0.3 * rng.standard_normal(N))
    A, b, c, cones = kit.logistic_regression(X, lab, lam)
    tpl = P.dense_template(A.shape[1], cones)
    *_, x, y, s, iters, status, resid = gpu_solve(tpl, A[None], b[None], c[None], eps=1e-9, max_iters=200000)
    assert status[0] == 1
    ref = minimize(lambda w: np.logaddexp(0.0, -lab * (X @ w)).sum() + lam * np.linalg.norm(w), np.ones(d), method="BFGS", options=dict(gtol=1e-10))
    np.testing.assert_allclose(c @ x.cpu().numpy()[0], ref.fun, atol=1e-6)
    np.testing.assert_allclose(x.cpu().numpy()[0][:d], ref.x, atol=2e-4)


def test_warm_start_matches_oracle_warm_start():
    """ce_settings.warm_start: the initial point (x, y, s) goes in through the output buffers; iteration counts and solutions follow
    the oracle's warm start (SCS: u = (x, y, 1), v = (0, s, 0)); a point with NaNs falls back to a cold start per instance."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 48
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=2)
    r0 = oracle.solve_batch(A, b, c, cones, eps=1e-6, max_iters=20000)
    rng = np.random.default_rng(9)
    b2 = b * (1 + 1e-3 * rng.standard_normal(b.shape)); c2 = c * (1 + 1e-3 * rng.standard_normal(c.shape))
    warm = [r0["x"].copy(), r0["y"].copy(), r0["s"].copy()]
    warm[0][5] = np.nan                                   # instance 5 must start cold
    rw = oracle.solve_batch(A, b2, c2, cones, eps=1e-6, max_iters=20000, warm=warm)
    rc = oracle.solve_batch(A, b2, c2, cones, eps=1e-6, max_iters=20000)
    assert rw["iters"][5] == rc["iters"][5] and rw["iters"].mean() < 0.7 * rc["iters"].mean()
    eng = _engine_for(tpl)
    A_eval, q_eval = tpl.values_from_dense(A, b2, c2)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda())
    wt = tuple(torch.from_numpy(w).cuda() for w in warm)
    x, y, s, iters, status, resid = eng.solve(A_bm, torch.from_numpy(q_eval).cuda(), make_settings(dict(acceleration_lookback=0, eps=1e-6, max_iters=20000)), warm=wt)
    assert (status.cpu().numpy() == 1).all()
    assert np.abs(iters.cpu().numpy() - rw["iters"]).max() <= 25, (iters.cpu().numpy(), rw["iters"])
    for got, want in ((x, rw["x"]), (y, rw["y"]), (s, rw["s"])):
        err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
        assert err.max() < 2e-5, err.max()


# ------------------------------------------------------------------ 3-d power cones (after the exponential cones; negative exponent = dual cone)
def test_power_cone_forward_and_adjoint_parity():
    run_parity(8, {"z": 1, "l": 3, "q": [3], "s": [], "ep": 1, "p": [0.3, -0.6, 0.5]}, 16, seed=3, eps=1e-9, max_iters=200000)


def test_geometric_mean_known_answer_on_gpu():
    A, b, c, cones, xs = kit.geo_mean_max(np.array([1.0, 2.0]), 2.0, 0.25)
    tpl = P.dense_template(A.shape[1], cones)
    *_, x, y, s, iters, status, resid = gpu_solve(tpl, A[None], b[None], c[None], eps=1e-10, max_iters=100000)
    assert status[0] == 1
    np.testing.assert_allclose(x.cpu().numpy()[0][:2], xs, atol=1e-6)


# ------------------------------------------------------------------ PSD / exponential cones on the 512-thread kernel variants (50 < n <= 98)
def test_psd_order_12_on_the_512_thread_variants():
    eng, A_bm, ref = _forward_parity(78, {"z": 6, "l": 0, "q": [], "s": [12]}, 3, seed=6, eps=1e-9)
    assert eng.launch_info()["fwd_mode"] == 4 and eng.launch_info()["bwd_mode"] == 3
    run_parity(78, {"z": 6, "l": 0, "q": [], "s": [12]}, 3, seed=6, eps=1e-9, max_iters=200000)


def test_exp_cones_on_the_512_thread_variants():
    run_parity(60, {"z": 4, "l": 12, "q": [4], "s": [], "ep": 28}, 4, seed=2, eps=1e-9, max_iters=200000)


def test_constant_A_path_with_exp_and_power_cones(monkeypatch):
    """Exponential / power triples on the batch-GEMM path: ce_ca_triples in the forward, ce_ca_triple_jac + batched LSQR in the adjoint."""
    from oracle import oracle
    monkeypatch.setenv("CE_CONST_A", "1")
    n, cones, B = 10, {"z": 2, "l": 3, "q": [4], "s": [3], "ep": 2, "p": [0.4, -0.7]}, 12
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=33, batched=("b", "c"))
    eps = 1e-9
    ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=200000)
    assert (ref["status"] == 1).all()
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, A, b, c, eps=eps, max_iters=200000)
    assert eng.last_path == "const_a" and (status == 1).all()
    for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
        assert np.abs(got.cpu().numpy() - want).max() < 1e-6 * (1 + np.abs(want).max())
    assert np.abs(iters - ref["iters"]).max() <= 25
    rng = np.random.default_rng(2)
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="dense")
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), lsqr=TIGHT_LSQR)
    assert (adj.cpu().numpy() == 0).all()
    assert np.abs(dq.cpu().numpy()[:n] - g["dc"].T).max() < 1e-5 * (1 + np.abs(g["dc"]).max())


def test_shared_A_kernels_with_exp_and_power_cones(monkeypatch):
    """Exponential / power triples inside the persistent shared-A kernels (k_sa_fwd: in-place projection with the previous root as warm
    start; k_sa_lsqr: symmetrised 3 x 3 derivative per triple).  Template in the shape canonicalisation produces: bounds on every variable
    (single-entry rows), a dense equality row, triples made of a dense row, an EMPTY row (constant 1) and a single-entry row."""
    from oracle import oracle
    monkeypatch.setenv("CE_CONST_A", "1")
    n = 9
    cones = {"z": 1, "l": n, "q": [3], "s": [], "ep": 2, "p": [0.35]}
    m = P.cone_rows(cones)
    rng = np.random.default_rng(5)
    A = np.zeros((m, n)); r = 0
    A[r, :6] = rng.standard_normal(6); r += 1                          # equality: dense row
    A[r:r + n, :] = -np.eye(n); r += n                                  # x >= lo
    A[r, 6] = -1.0; A[r + 1, 0] = -1.0; A[r + 2, 1] = -1.0; r += 3      # SOC(3) of single-entry rows
    for t in range(3):                                                  # two exponential triples, one power triple
        A[r, :6] = -rng.standard_normal(6) * 0.5                        # (x.w, 1, z)-style: dense row, empty row, epigraph variable
        A[r + 2, 6 + (t % 3)] = -1.0
        r += 3
    assert r == m
    B = 10
    x0 = rng.standard_normal((B, n)) * 0.3
    s0 = np.zeros((B, m)); s0[:, 1:1 + n] = 1.0 + rng.random((B, n)); s0[:, 1 + n] = 2.0
    for t in range(2): s0[:, 4 + n + 3 * t:7 + n + 3 * t] = np.array([-0.3, 1.0, 1.5])       # interior of the exponential cone: y exp(x / y) < z
    s0[:, 10 + n:13 + n] = np.array([1.2, 1.1, 0.4])                                        # interior of the power cone
    b = x0 @ A.T + s0
    # a strictly feasible dual point per instance makes the program solvable: c = -A^T y0, y0 in the interior of K*
    y0 = np.zeros((B, m)); y0[:, 0] = rng.standard_normal(B); y0[:, 1:1 + n] = 0.5 + rng.random((B, n)); y0[:, 1 + n:4 + n] = np.array([2.0, 0.5, -0.5])
    for t in range(2): y0[:, 4 + n + 3 * t:7 + n + 3 * t] = np.array([-1.0, 0.5, 1.0])       # -u exp(v / u) = exp(-0.5) <= e w
    y0[:, 10 + n:13 + n] = np.array([1.0, 1.0, 0.3])
    y0 *= (0.5 + rng.random((B, 1)))
    c = -(y0 @ A)
    tpl = P.dense_template(n, cones, pattern=(A != 0))
    Ab = np.broadcast_to(A, (B,) + A.shape).copy()
    ref = oracle.solve_batch(Ab, b, c, cones, eps=1e-9, max_iters=200000)
    ok = ref["status"] == 1
    assert ok.mean() > 0.5, ref["status"]
    eng, A_bm, x, y, s, iters, status, resid = gpu_solve(tpl, Ab, b, c, eps=1e-9, max_iters=200000)
    assert eng.last_path == "const_a"
    if os.environ.get("CE_SA_FWD") != "0":          # (the toggle exists for A/B runs of the batch-GEMM path)
        assert eng.last_const_a_kernel == "k_sa_fwd"
    assert (status[ok] == 1).all(), status
    for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
        assert np.abs(got.cpu().numpy()[ok] - want[ok]).max() < 1e-6 * (1 + np.abs(want[ok]).max())
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    g = oracle.adjoint_batch(Ab, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="dense")
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), lsqr=TIGHT_LSQR)
    if os.environ.get("CE_SA_KERNEL") != "0":
        assert eng.last_lsqr_iters is not None
    good = ok & (adj.cpu().numpy() == 0)
    assert good.mean() > 0.5
    assert np.abs(dq.cpu().numpy()[:n].T[good] - g["dc"][good]).max() < 1e-5 * (1 + np.abs(g["dc"][good]).max())


def test_anderson_acceleration_gives_up_like_the_oracle_on_linear_programs():
    """The give-up rule (AA_MAX_REJECT safeguard rejections) in k_fwd2 and in the oracle: on slowly converging random LPs both switch the
    acceleration off for the same instances, so iteration counts stay within a check interval or two and nobody runs into the limit."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    cfg = P.CONFIGS["C2"]; n, cones, B = cfg["n"], cfg["cones"], 32
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=0)
    eng = _engine_for(tpl)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda()); q_t = torch.from_numpy(q_eval).cuda()
    ref = oracle.solve_batch(A, b, c, cones, eps=1e-4, max_iters=20000, acceleration_lookback=1)
    plain = oracle.solve_batch(A, b, c, cones, eps=1e-4, max_iters=20000)
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-4, max_iters=20000, acceleration_lookback=1)))
    it = iters.cpu().numpy()
    assert (status.cpu().numpy() == 1).all() and (ref["status"] == 1).all()
    assert it.mean() < 1.25 * plain["iters"].mean(), (it.mean(), plain["iters"].mean())
    # (after the acceleration is off both sides run the plain iteration from slightly different points: counts agree loosely)
    assert np.median(np.abs(it - ref["iters"]) / ref["iters"]) < 0.2, (it, ref["iters"])


def test_anderson_acceleration_matches_the_oracle_with_memory_one():
    """acceleration_lookback > 0: k_fwd2 applies type-I Anderson acceleration with a one-pair history every acceleration_interval
    iterations; the oracle with aa_mem = 1 is the same algorithm (iteration counts within one check interval, same solutions,
    fewer iterations than the plain iteration)."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 64
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=0)
    eng = _engine_for(tpl)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda()); q_t = torch.from_numpy(q_eval).cuda()
    for eps in (1e-4, 1e-8):
        ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=20000, acceleration_lookback=1)
        plain = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=20000)
        x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(eps=eps, max_iters=20000, acceleration_lookback=1)))
        assert (status.cpu().numpy() == 1).all() and (ref["status"] == 1).all()
        tol = max(1e-6, 20 * eps)
        for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
            err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
            assert err.max() < tol, err.max()
        it = iters.cpu().numpy()
        assert np.mean(np.abs(it - ref["iters"]) <= 25) > 0.9, (it, ref["iters"])       # a borderline safeguard decision may shift an instance
        assert it.mean() < plain["iters"].mean()


@pytest.mark.parametrize("cfg", ["C4", "C5lite"])
def test_anderson_acceleration_in_the_shared_A_kernel_matches_the_oracle(cfg):
    """k_sa_fwd (BASELINE configs 4 / 5: A shared by the batch) runs the same one-pair Anderson acceleration as k_fwd2, with its history in
    global memory: same solutions as the oracle with aa_mem = 1, iteration counts within a check interval, fewer iterations than plain."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    if cfg == "C4":
        B = 24
        A1, b, c, cones, tpl = P.sdp_c4_batch(B, seed=3)
    else:
        B = 16
        A1, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=3, nw=120, kf=12)
    m, n = A1.shape
    A = np.broadcast_to(A1, (B, m, n)).copy()
    b = np.broadcast_to(b, (B, m)).copy() if b.ndim == 1 else b
    eng = _engine_for(tpl)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda()); q_t = torch.from_numpy(q_eval).cuda()
    import os
    os.environ["CE_CONST_A"] = "1"
    try:
        for eps in ((1e-4, 1e-7) if cfg == "C4" else (1e-4, 1e-5)):      # (the portfolio needs ~9000 iterations at 1e-7: too close to max_iters for a status assertion)
            ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=20000, acceleration_lookback=1)
            plain = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=20000)
            x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(eps=eps, max_iters=20000, acceleration_lookback=1)))
            assert eng.last_path == "const_a" and eng.last_const_a_kernel == "k_sa_fwd" and eng.last_acceleration
            assert (status.cpu().numpy() == 1).all() and (ref["status"] == 1).all()
            tol = max(1e-6, 20 * eps)
            for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
                err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
                assert err.max() < tol, (cfg, eps, err.max())
            it = iters.cpu().numpy()
            assert np.mean(np.abs(it - ref["iters"]) <= 25) >= 0.85, (it, ref["iters"])
            # (on these two shapes the acceleration does not pay -- profiles/r02/aa_memory.json: 114.8 -> 112.5 / 415.6 -> 442.2 iterations at eps 1e-4, with
            #  memory 10 as with memory 1 -- so there is no "fewer iterations than plain" assertion here; the point is that every path runs the SAME algorithm
            #  for the same solver_args, and that the accelerated trajectory is the oracle's accelerated trajectory, not the plain one)
            assert not np.array_equal(ref["iters"], plain["iters"]) or eps > 1e-5
    finally:
        os.environ.pop("CE_CONST_A", None)


def test_rescale_by_neumann_series_gives_the_iterates_of_a_refactorisation(monkeypatch):
    """k_fwd2 updates G = (rho I + A^T Dy A)^-1 by a Neumann series when the adaptive scale changes (ce_forward_v2.h, refactor()): same solutions and the
    SAME iteration counts as the build that refactors (CE_F2_NEUMANN=0), on long runs where every instance rescales (LPs: hundreds to thousands of iterations)."""
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    n, cones, B = 30, {"z": 2, "l": 40, "q": [6, 5]}, 48
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=5)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("CE_F2_NEUMANN", flag)            # read at ce_create
        eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, tpl.cones, torch.device("cuda", 0))
        A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda())
        x, y, s, iters, status, resid = eng.solve(A_bm, torch.from_numpy(q_eval).cuda(), make_settings(dict(eps=1e-9, max_iters=100000, acceleration_lookback=0)))
        out[flag] = (x.cpu().numpy(), iters.cpu().numpy(), status.cpu().numpy())
    assert (out["1"][2] == 1).all() and (out["0"][2] == 1).all()
    assert out["1"][1].max() > 150                                             # past RESCALING_MIN_ITERS: rescales happened
    assert np.array_equal(out["1"][1], out["0"][1]), (out["1"][1], out["0"][1])
    assert np.abs(out["1"][0] - out["0"][0]).max() < 1e-9


def test_blocked_generic_backward_is_rank_revealing(monkeypatch):
    """n = 120 with per-instance A and K in global memory (the blocked Gauss-Jordan of the size-generic backward kernel) on a template with a REDUNDANT
    equality row (row 1 = row 0, consistent right-hand side): sixteen pivots per pass cannot skip a column, so the kernel rebuilds K and hands over to
    the rank-revealing unblocked elimination.  Same gradients as the unblocked path run on its own (CE_GEN_BLOCKED=0), status bit 2 (= 4: rank deficient,
    not a failure), finite everywhere; on the same template WITHOUT the redundancy the blocked path itself answers (status 0)."""
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    cones = {"z": 2, "l": 58, "q": [21] * 6}
    n, B = 120, 4
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=11)
    outs = {}
    for redundant in (True, False):
        A2, b2 = A.copy(), b.copy()
        if redundant:
            A2[:, 1, :] = A2[:, 0, :]; b2[:, 1] = b2[:, 0]
        A_eval, q_eval = tpl.values_from_dense(A2, b2, c)
        for blocked in ("1", "0"):
            monkeypatch.setenv("CE_GEN_BLOCKED", blocked)
            eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
            assert eng.launch_info()["bwd_mode"] == 2
            A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
            x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-9, max_iters=100000, acceleration_lookback=0)))
            assert (st == 1).all()
            g = torch.Generator(device="cpu").manual_seed(5)
            dx = torch.randn(x.shape, generator=g, dtype=torch.float64).cuda(); dy = torch.randn(y.shape, generator=g, dtype=torch.float64).cuda()
            dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy)
            torch.cuda.synchronize()
            outs[(redundant, blocked)] = (dA.cpu().numpy().copy(), dq.cpu().numpy().copy(), adj.cpu().numpy().copy())
    for redundant in (True, False):
        (dA1, dq1, adj1), (dA0, dq0, adj0) = outs[(redundant, "1")], outs[(redundant, "0")]
        assert np.isfinite(dA1).all() and np.isfinite(dq1).all()
        want = 4 if redundant else 0
        assert (adj1 == want).all() and (adj0 == want).all(), (redundant, adj1, adj0)
        sc = 1 + np.abs(dA0).max()
        assert np.abs(dA1 - dA0).max() < (1e-12 if redundant else 1e-6) * sc and np.abs(dq1 - dq0).max() < (1e-12 if redundant else 1e-6) * (1 + np.abs(dq0).max())


@pytest.mark.parametrize("cones,n", [({"z": 0, "l": 30, "q": [5], "s": [], "ep": 50}, 60),            # m = 185 > 160: 50 exponential cones beyond k_fwd2
                                     ({"z": 5, "l": 40, "q": [], "s": [14]}, 80),                     # m = 150 > 120 at n = 80 > 62: a 14 x 14 PSD block beyond k_fwd2
                                     ({"z": 2, "l": 100, "q": [6, 4], "s": [6], "ep": 12, "p": [0.3, -0.6]}, 72)])      # every cone type at once, m = 175
def test_every_cone_type_on_the_size_generic_forward_kernel(cones, n):
    """PSD / exponential / power cones with per-instance A beyond the sizes of k_fwd2 (n <= 62 with m <= 160, n <= 104 with m <= 120): the size-generic forward kernel projects them
    (workgroup-parallel Jacobi in LDS, one thread per triple) with block-averaged equilibration, the 512-thread register-tiled backward kernel
    differentiates them; both against the oracle."""
    eng = run_parity(n, cones, 4, seed=21, eps=1e-9, max_iters=200000)
    info = eng.launch_info()
    assert info["fwd_mode"] in (0, 1, 2) and info["bwd_mode"] == 3, info


@pytest.mark.parametrize("cones,n", [({"z": 5, "l": 40, "q": [], "s": [14]}, 100),                                     # m = 150: dense A (120 KB) + the PSD scratch exceed the register-tiled adjoint's LDS
                                     ({"z": 0, "l": 30, "q": [7], "s": [], "ep": 60}, 120),                            # n = 120, m = 217: sixty exponential cones
                                     ({"z": 3, "l": 80, "q": [9, 5], "s": [8, 5], "ep": 10, "p": [0.25, -0.5]}, 118)])  # every cone type, two PSD blocks, m = 184
def test_every_cone_type_on_the_size_generic_kernels_forward_and_backward(cones, n):
    """Beyond the register-tiled adjoint as well (n > 112, or A + the PSD scratch beyond LDS): the size-generic backward kernel rotates PSD blocks and
    triples into the eigenbasis of their projection's derivative (the construction of k_backward_rt<PSD>, with A possibly in global memory) -- no
    CE_E_UNSUPPORTED left for per-instance templates with these cones."""
    eng = run_parity(n, cones, 3, seed=23, eps=1e-9, max_iters=200000)
    info = eng.launch_info()
    assert info["fwd_mode"] in (0, 1, 2) and info["bwd_mode"] in (0, 1, 2), info


def test_rank_tolerance_is_one_constant_across_the_adjoint_kernels(monkeypatch):
    """A template with a REDUNDANT equality row (row 1 = row 0, consistent right-hand side) on the register-tiled adjoint (the default for this size) and on
    the size-generic kernel (CE_FORCE_GENERIC=1) with its unblocked and its blocked elimination: the same rank tolerance (CE_RANK_TOL = 1e-11 max|K|, the
    oracle's) decides what a vanishing pivot is in all of them, so the three return the same basic solution and the same flag (4: rank deficient)."""
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    n, B = 24, 8
    A0, b0, c = P.generate(n, {"z": 2, "l": 20, "q": [6, 5]}, B, seed=21)
    A = np.concatenate([A0[:, :1, :], A0], axis=1); b = np.concatenate([b0[:, :1], b0], axis=1)      # equality row 0 twice: consistent, and (x0, (0, y0)) stays a certificate
    cones = {"z": 3, "l": 20, "q": [6, 5]}
    tpl = P.dense_template(n, cones)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    outs = {}
    for key, env in (("rt", {}), ("generic_unblocked", {"CE_FORCE_GENERIC": "1", "CE_GEN_BLOCKED": "0"}), ("generic_blocked", {"CE_FORCE_GENERIC": "1", "CE_GEN_BLOCKED": "1"})):
        for k in ("CE_FORCE_GENERIC", "CE_GEN_BLOCKED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
        assert (eng.launch_info()["bwd_mode"] == 3) == (key == "rt"), (key, eng.launch_info())
        A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
        if key == "rt":
            x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-8, max_iters=200000, acceleration_lookback=0)))
            assert ((st == 1) | (st == 2)).all(), st          # (only a common point at which the three eliminations are compared is needed)
            sol = (x, y, s)
        g = torch.Generator(device="cpu").manual_seed(5)
        dx = torch.randn(sol[0].shape, generator=g, dtype=torch.float64).cuda(); dy = torch.randn(sol[1].shape, generator=g, dtype=torch.float64).cuda()
        dA, dq, adj = eng.vjp(A_bm, *sol, dx, dy)          # (the SAME solution for the three: only the eliminations differ)
        torch.cuda.synchronize()
        outs[key] = (dA.cpu().numpy().copy(), dq.cpu().numpy().copy(), adj.cpu().numpy().copy())
    ref = outs["rt"]
    assert (ref[2] == 4).all(), ref[2]
    for key in ("generic_unblocked", "generic_blocked"):
        dA, dq, adj = outs[key]
        # (an instance with more active rows than variables exceeds the size-generic kernel's system size n + min(m, n) and is flagged 2 there -- zero gradient --
        # while the register tile still holds it: compared are the instances both kernels solve)
        ok = adj == 4
        assert ok.sum() >= len(adj) - 2 and set(adj[~ok].tolist()) <= {2}, (key, adj)
        assert np.isfinite(dA).all() and np.isfinite(dq).all()
        # dA, db, dc are the same for every solution of the consistent singular system only up to the free variable's choice: both kernels set it to zero
        assert np.abs(dq[:, ok] - ref[1][:, ok]).max() < 1e-8 * (1 + np.abs(ref[1]).max()), (key, np.abs(dq[:, ok] - ref[1][:, ok]).max())
        assert np.abs(dA[:, ok] - ref[0][:, ok]).max() < 1e-8 * (1 + np.abs(ref[0]).max()), (key, np.abs(dA[:, ok] - ref[0][:, ok]).max())


def test_longest_first_dispatch_is_a_scheduling_hint_only():
    """ce_set_dispatch_history: once two consecutive solves of one batch size have shown the same iteration counts (the history is predictive), the next solve
    dispatches its workgroups by the recorded counts, longest first; on unrelated batches the index order is kept.  Every instance is computed by the same code
    whatever workgroup index it gets: solutions, iteration counts and statuses are BIT-identical with the hint on and off, applied or not."""
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 777
    tpl = P.dense_template(n, cones)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
    st = lambda: make_settings(dict(eps=1e-4, max_iters=10000, acceleration_lookback=10))
    outs = []
    for seed in (0, 1):
        A, b, c = P.generate(n, cones, B, seed=seed)
        A_eval, q_eval = tpl.values_from_dense(A, b, c)
        A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
        eng.set_dispatch_history(False)
        ref = [t.clone() for t in eng.solve(A_bm, q_t, st())]
        eng.set_dispatch_history(True)
        first = [t.clone() for t in eng.solve(A_bm, q_t, st())]          # seed 0: no order yet; seed 1: seed 0's order exists but did not predict this batch: not applied
        second = [t.clone() for t in eng.solve(A_bm, q_t, st())]         # an order exists, its predictive flag is still 0
        third = [t.clone() for t in eng.solve(A_bm, q_t, st())]          # first -> second agreed everywhere: dispatched longest-first by `second`'s counts
        for got in (first, second, third):
            for a_, b_ in zip(got, ref):
                assert torch.equal(a_, b_)
        assert len(torch.unique(ref[3])) > 1                              # (the batch does have instances of different length)
        outs.append(ref)
    # smaller batch afterwards: the recorded order (777 entries) must not be applied
    A, b, c = P.generate(n, cones, 100, seed=2)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    x1 = eng.solve(A_bm, q_t, st())[0].clone()
    eng.set_dispatch_history(False)
    assert torch.equal(x1, eng.solve(A_bm, q_t, st())[0])


def test_two_tile_adjoint_plan_gives_the_single_tile_gradients(monkeypatch):
    """k_backward_rt's tile holds the template's worst case (every row active); the two-tile plan serves the batch on a smaller tile first -- sized by the
    largest system of the previous call -- and re-runs the instances that do not fit it on the worst-case tile (ce_vjp_qp, cone_engine.hip).  Same elimination, same pivots, the same fused multiply-adds per
    entry whatever the tile: the gradients are BIT-identical to the single-tile plan -- by default (few or no retries at the metric shape) and with a first
    tile forced too small (CE_BWD_FAST_VARIANT=0: 63 unknowns, about half of the instances are retried)."""
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 512
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=3)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    g = torch.Generator(device="cpu").manual_seed(1)
    outs = {}
    sol = None
    for key, env in (("single", {"CE_BWD_TWO_TILE": "0"}), ("two", {}), ("forced", {"CE_BWD_FAST_VARIANT": "0"})):
        for k in ("CE_BWD_TWO_TILE", "CE_BWD_FAST_VARIANT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
        if sol is None:
            x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-8, max_iters=20000, acceleration_lookback=0)))
            assert (st == 1).all()
            sol = (x, y, s)
            dx = torch.randn(x.shape, generator=g, dtype=torch.float64).cuda(); dy = torch.randn(y.shape, generator=g, dtype=torch.float64).cuda()
        dA, dq, adj = eng.vjp(A_bm, *sol, dx, dy)
        torch.cuda.synchronize()
        if key != "single":           # the first call of an engine has no history (worst-case tile): the SECOND one runs the plan
            dA, dq, adj = eng.vjp(A_bm, *sol, dx, dy)
            torch.cuda.synchronize()
        outs[key] = (dA.clone(), dq.clone(), adj.clone())
    assert int((outs["single"][2] != 0).sum()) == 0
    v = sol[1] - sol[2]
    nk = n + (v[:, :cones["l"]] > 0).sum(dim=1) + len(cones["q"])           # (n + active bounds + one row per boundary cone: what most instances have)
    assert int((nk > 63).sum()) > B // 8                                      # the forced plan really does retry a good part of the batch
    for key in ("two", "forced"):
        for a_, b_ in zip(outs[key], outs["single"]):
            assert torch.equal(a_, b_), key


@pytest.mark.parametrize("kernel", ["generic", "rt"])
def test_anderson_acceleration_in_the_fallback_forward_kernels_matches_the_oracle(monkeypatch, kernel):
    """k_forward (size-generic: templates beyond the register-tiled kernels) and k_forward_rt (first-generation register-tiled kernel), forced here with CE_FORCE_GENERIC /
    CE_FWD=rt, run the one-pair Anderson acceleration of k_fwd2 / k_sa_fwd with their history in global memory: the oracle with aa_mem = 1 is the same algorithm (iteration
    counts within a check interval, same solutions, fewer iterations than the plain iteration), and a positive acceleration_lookback is honoured without a warning."""
    import warnings
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    if kernel == "generic": monkeypatch.setenv("CE_FORCE_GENERIC", "1")
    else: monkeypatch.setenv("CE_FWD", "rt")
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 32
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=2)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
    assert (eng.launch_info()["fwd_mode"] <= 2) if kernel == "generic" else (eng.launch_info()["fwd_mode"] == 3), eng.launch_info()
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda()); q_t = torch.from_numpy(q_eval).cuda()
    for eps in (1e-4, 1e-8):
        ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=20000, acceleration_lookback=1)
        plain = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=20000)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(eps=eps, max_iters=20000, acceleration_lookback=1)))
        assert eng.last_acceleration
        assert (status.cpu().numpy() == 1).all() and (ref["status"] == 1).all()
        tol = max(1e-6, 20 * eps)
        for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
            err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
            assert err.max() < tol, err.max()
        it = iters.cpu().numpy()
        assert np.mean(np.abs(it - ref["iters"]) <= 25) > 0.9, (it, ref["iters"])       # a borderline safeguard decision may shift an instance
        assert it.mean() < plain["iters"].mean()
