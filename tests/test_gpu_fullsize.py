"""Full-size checks at BASELINE.json's metric configuration (B=4096, n=50, m=100, SOC) through size-independent
properties -- the oracle would need minutes for these batches, so only a random subset is compared with it:
  * KKT conditions of every returned (x, y, s): primal / dual residuals, duality gap, cone membership, complementarity;
  * the adjoint is the derivative of the solution map: directional finite differences of the GPU solve itself;
  * linearity of the VJP in (dx, dy); batch-order invariance (instances are independent);
  * edge cases: batch of one, batch sizes that are not multiples of anything, unbatched (1-D) inputs."""
import numpy as np
import pytest
import torch

from cvxpylayers_amd import problems as P

pytestmark = pytest.mark.gpu


def _setup(n, cones, B, seed):
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=seed)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, tpl.cones, torch.device("cuda", 0))
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous()
    return tpl, eng, A, b, c, A_bm, torch.from_numpy(q_eval).cuda()


def _soc_violation(v, cones):
    off = cones.get("z", 0) + cones.get("l", 0)
    worst = 0.0
    for d in cones.get("q", []):
        blk = v[:, off:off + d]
        worst = max(worst, float((np.linalg.norm(blk[:, 1:], axis=1) - blk[:, 0]).max()))
        off += d
    return worst


def test_metric_config_full_batch_kkt_and_subset_parity():
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    from oracle import oracle
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 4096
    tpl, eng, A, b, c, A_bm, q_t = _setup(n, cones, B, seed=0)
    eps = 1e-8
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=20000)))
    assert (status.cpu().numpy() == 1).all()
    x, y, s = x.cpu().numpy(), y.cpu().numpy(), s.cpu().numpy()
    # KKT: A x + s = b, A^T y + c = 0, c^T x + b^T y = 0, s in K, y in K*, s . y = 0
    rp = np.abs(np.einsum("bij,bj->bi", A, x) + s - b).max(axis=1) / (1 + np.abs(b).max(axis=1))
    rd = np.abs(np.einsum("bij,bi->bj", A, y) + c).max(axis=1) / (1 + np.abs(c).max(axis=1))
    gap = np.abs(np.einsum("bj,bj->b", c, x) + np.einsum("bi,bi->b", b, y)) / (1 + np.abs(np.einsum("bj,bj->b", c, x)))
    assert rp.max() < 50 * eps and rd.max() < 50 * eps and gap.max() < 50 * eps, (rp.max(), rd.max(), gap.max())
    l0, l1 = cones["z"], cones["z"] + cones["l"]
    assert s[:, l0:l1].min() > -1e-7 and y[:, l0:l1].min() > -1e-7
    assert _soc_violation(s, cones) < 1e-7 and _soc_violation(y, cones) < 1e-7
    assert np.abs(np.einsum("bi,bi->b", s, y)).max() < 1e-5
    # a random subset against the oracle
    idx = np.random.default_rng(1).choice(B, 48, replace=False)
    ref = oracle.solve_batch(A[idx], b[idx], c[idx], cones, eps=eps, max_iters=20000)
    assert np.abs(x[idx] - ref["x"]).max() < 1e-6 * (1 + np.abs(ref["x"]).max())
    assert np.abs(y[idx] - ref["y"]).max() < 1e-6 * (1 + np.abs(ref["y"]).max())


def test_adjoint_is_the_derivative_of_the_gpu_solution_map():
    """<dx, x(b + h db, c + h dc) - x(b - h db, c - h dc)> / 2h  ==  <db_grad, db> + <dc_grad, dc>  on the GPU path alone."""
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 64
    tpl, eng, A, b, c, A_bm, q_t = _setup(n, cones, B, seed=7)
    st = make_settings(dict(acceleration_lookback=0, eps=1e-11, max_iters=100000))
    x, y, s, *_ = eng.solve(A_bm, q_t, st)
    rng = np.random.default_rng(3)
    dx = torch.from_numpy(rng.standard_normal((B, n))).cuda()
    dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, torch.zeros_like(y))
    assert (adj.cpu().numpy() == 0).all()
    db_dir = rng.standard_normal(b.shape); dc_dir = rng.standard_normal(c.shape); h = 1e-5
    outs = []
    for sign in (+1, -1):
        A_eval, q_eval = tpl.values_from_dense(A, b + sign * h * db_dir, c + sign * h * dc_dir)
        xs, *_ = eng.solve(torch.from_numpy(A_eval).cuda().t().contiguous(), torch.from_numpy(q_eval).cuda(), st)
        outs.append(xs)
    fd = ((outs[0] - outs[1]) * dx).sum(dim=1).cpu().numpy() / (2 * h)
    dA_np = dA.cpu().numpy(); dq_np = dq.cpu().numpy()
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    db_grad = np.zeros_like(b)
    for k in np.nonzero(cols == n)[0]:
        db_grad[:, tpl.indices[k]] = dA_np[k]            # dA_eval's b part is +db (diffcp_if.py:91)
    an = (db_grad * db_dir).sum(axis=1) + (dq_np[:n].T * dc_dir).sum(axis=1)
    assert np.abs(fd - an).max() < 2e-4 * (1 + np.abs(an).max()), np.abs(fd - an).max()


def test_vjp_is_linear_and_batch_order_invariant():
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 37          # deliberately not a multiple of anything
    tpl, eng, A, b, c, A_bm, q_t = _setup(n, cones, B, seed=9)
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=1e-9)))
    rng = np.random.default_rng(0)
    d1 = [torch.from_numpy(rng.standard_normal(t.shape)).cuda() for t in (x, y)]
    d2 = [torch.from_numpy(rng.standard_normal(t.shape)).cuda() for t in (x, y)]
    g1 = eng.vjp(A_bm, x, y, s, *d1); g2 = eng.vjp(A_bm, x, y, s, *d2)
    g12 = eng.vjp(A_bm, x, y, s, 2.0 * d1[0] - 3.0 * d2[0], 2.0 * d1[1] - 3.0 * d2[1])
    assert torch.allclose(g12[0], 2.0 * g1[0] - 3.0 * g2[0], rtol=1e-9, atol=1e-9)
    assert torch.allclose(g12[1], 2.0 * g1[1] - 3.0 * g2[1], rtol=1e-9, atol=1e-9)
    perm = torch.randperm(B, device="cuda")
    xp, yp, sp, itp, stp, _ = eng.solve(A_bm[perm].contiguous(), q_t[:, perm].contiguous(), make_settings(dict(acceleration_lookback=0, eps=1e-9)))
    assert torch.equal(xp, x[perm]) and torch.equal(yp, y[perm]) and torch.equal(itp, iters[perm])


def test_batch_of_one_and_unbatched_through_the_plugin():
    from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
    cfg = P.CONFIGS["M"]; n, cones = cfg["n"], cfg["cones"]
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, 1, seed=4)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-9})
    A1 = torch.from_numpy(A_eval[:, 0]).cuda().requires_grad_(); q1 = torch.from_numpy(q_eval[:, 0]).cuda().requires_grad_()
    p1, d1, *_ = _CvxpyLayer.apply(None, q1, A1, ctx, {}, True, None)
    assert p1.shape == (1, n) and d1.shape == (1, tpl.m)          # the plugin always returns 2-D (torch/cvxpylayer.py:247-249)
    p1.sum().backward()
    assert A1.grad.shape == A1.shape and q1.grad.shape == q1.shape      # gradients squeezed for 1-D inputs (diffcp_if.py:399-401)
    A2 = torch.from_numpy(A_eval).cuda().requires_grad_(); q2 = torch.from_numpy(q_eval).cuda().requires_grad_()
    p2, d2, *_ = _CvxpyLayer.apply(None, q2, A2, ctx, {}, True, None)
    p2.sum().backward()
    assert A2.grad.shape == (tpl.nnz_aug, 1) and torch.allclose(p1, p2) and torch.allclose(A2.grad[:, 0], A1.grad)
