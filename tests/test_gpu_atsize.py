"""Every BASELINE.json configuration at its stated size, asserted against the oracle on the GPU:
    C3  SOCP n=100, 10 x SOC(11) + 10 nonneg rows (m=120), per-instance A                  B = 256
    C4  SDP, one 20 x 20 PSD cone, n=210, m=230, shared A (b, c per instance)               B = 64
    C5  portfolio n=501, m=552 (zero + 500 nonneg + SOC(51)), shared A (mu per instance)    B = 64
(config M and config 2 at size: tests/test_gpu_fullsize.py, tests/test_quad_objective.py.)  Both sides run the same settings;
tolerances: solutions 1e-6 (1 + |x|_inf), gradients 1e-5 relative, as SURVEY.md 8d states."""
import numpy as np
import pytest
import torch

from cvxpylayers_amd import problems as P
from kit import TIGHT_LSQR, lsqr_own_movement, assert_lsqr_agreement_per_instance

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))


def _solve_and_vjp(tpl, cones, A, b, c, eps, max_iters, dx=None):
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    B, n, m = c.shape[0], tpl.n, tpl.m
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    eng = ConeEngine(tpl.indices, tpl.indptr, n, m, cones, torch.device("cuda", 0))
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    x, y, s, it, status, res = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=max_iters)))
    path = eng.last_path
    dxt = torch.ones_like(x) if dx is None else torch.from_numpy(dx).cuda()
    dA, dq, adj = eng.vjp(A_bm, x, y, s, dxt, torch.zeros_like(y), path=path, lsqr=TIGHT_LSQR)
    return eng, path, (x.cpu().numpy(), y.cpu().numpy(), s.cpu().numpy()), it.cpu().numpy(), status.cpu().numpy(), dA.cpu().numpy(), dq.cpu().numpy(), adj.cpu().numpy()


def _db_from_dA(tpl, dA_np, B):
    cols = np.repeat(np.arange(tpl.n + 1), np.diff(tpl.indptr))
    db = np.zeros((B, tpl.m))
    for kk in np.nonzero(cols == tpl.n)[0]:
        db[:, tpl.indices[kk]] = dA_np[kk]
    return db


def test_C3_socp_n100_at_size():
    from oracle import oracle
    cfg = P.CONFIGS["C3"]; n, cones, B = cfg["n"], cfg["cones"], 256
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=3)
    eng, path, (x, y, s), it, status, dA, dq, adj = _solve_and_vjp(tpl, cones, A, b, c, 1e-8, 50000)
    assert path == "per_instance" and (status == 1).all() and (adj == 0).all()
    ref = oracle.solve_batch(A, b, c, cones, eps=1e-8, max_iters=50000)
    assert (ref["status"] == 1).all()
    assert _rel(x, ref["x"]) < 1e-6 and _rel(y, ref["y"]) < 1e-6 and _rel(s, ref["s"]) < 1e-6
    assert np.abs(it.astype(int) - ref["iters"]).max() <= 25                       # same algorithm: within one check interval
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], np.ones_like(ref["x"]), np.zeros_like(ref["y"]), mode="dense")
    assert _rel(dq[:n].T, g["dc"]) < 1e-5 and _rel(_db_from_dA(tpl, dA, B), g["db"]) < 1e-5
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    ka = np.nonzero(cols < n)[0]
    assert _rel(-dA[ka].T, g["dA"][:, tpl.indices[ka], cols[ka]]) < 1e-5            # dA_eval = -dA.data (diffcp_if.py:91)


def test_C4_sdp_20x20_at_size():
    from oracle import oracle
    B = 64
    A, b, c, cones, tpl = P.sdp_c4_batch(B, seed=0)
    Ab = np.broadcast_to(A, (B,) + A.shape).copy()
    eng, path, (x, y, s), it, status, dA, dq, adj = _solve_and_vjp(tpl, cones, Ab, b, c, 1e-8, 50000)
    assert (status == 1).all(), status
    ref = oracle.solve_batch(Ab, b, c, cones, eps=1e-8, max_iters=50000)
    assert (ref["status"] == 1).all()
    assert _rel(x, ref["x"]) < 1e-6 and _rel(y, ref["y"]) < 1e-6 and _rel(s, ref["s"]) < 1e-6
    # X = smat(x) is PSD and feasible: the property form of the same statement, independent of the oracle
    X = P.svec_to_sym(x, 20)
    assert np.linalg.eigvalsh(X).min() > -1e-6
    assert np.abs(x @ A[:20].T - b[:, :20]).max() < 1e-6 * (1 + np.abs(b).max())
    g = oracle.adjoint_batch(Ab, b, c, cones, ref["x"], ref["y"], ref["s"], np.ones_like(ref["x"]), np.zeros_like(ref["y"]), mode="dense")
    assert (adj == 0).all()
    assert _rel(dq[:tpl.n].T, g["dc"]) < 1e-5, _rel(dq[:tpl.n].T, g["dc"])
    assert _rel(_db_from_dA(tpl, dA, B), g["db"]) < 1e-5


def test_C5_portfolio_n501_at_size():
    from oracle import oracle
    B = 64
    A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=0)
    Ab = np.broadcast_to(A, (B,) + A.shape).copy(); bb = np.broadcast_to(b, (B,) + b.shape).copy()
    eps = 1e-6          # splitting converges slowly on this LP-like program (4 000 - 9 000 iterations at 1e-6, > 1e5 at 1e-8, on both sides)
    eng, path, (x, y, s), it, status, dA, dq, adj = _solve_and_vjp(tpl, cones, Ab, bb, c, eps, 100000)
    assert path == "const_a" and (status == 1).all(), (path, status)
    ref = oracle.solve_batch(Ab, bb, c, cones, eps=eps, max_iters=100000)
    assert (ref["status"] == 1).all()
    assert _rel(x, ref["x"]) < 1e-6 and _rel(y, ref["y"]) < 1e-6 and _rel(s, ref["s"]) < 1e-6
    assert np.abs(it.astype(int) - ref["iters"]).max() <= 25
    # feasibility / optimality properties (independent of the oracle): budget, nonnegativity, the risk cone, duality gap
    w, t = x[:, :500], x[:, 500]
    assert np.abs(w.sum(axis=1) - 1).max() < 1e-5 and w.min() > -1e-5
    assert (np.linalg.norm(w @ (-A[502:, :500].T), axis=1) - t).max() < 1e-5
    assert np.abs((c * x).sum(axis=1) + (bb * y).sum(axis=1)).max() < 1e-4
    # adjoint of dx = 1 at the oracle's own solution (so that only the adjoint solves are compared), against the oracle's LSQR mode (diffcp's default and its
    # semantics) at matched tolerances: both run LSQR on the full (n + m + 1) system and return ITS minimum-norm solution -- on every instance, degenerate
    # faces included (tests/test_gpu_baseline_batches.py has the longer story; the dense elimination agrees at vertices only, to the accuracy of the point).
    A_eval, q_eval = tpl.values_from_dense(Ab, bb, c)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous()
    xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA2, dq2, adj2 = eng.vjp(A_bm, xo, yo, so, torch.ones_like(xo), torch.zeros_like(yo), path="const_a", lsqr=TIGHT_LSQR, q_eval=torch.from_numpy(q_eval).cuda())
    ones, zeros = np.ones_like(ref["x"]), np.zeros_like(ref["y"])
    gl = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], ones, zeros, mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    assert int((adj2 != 0).sum().item()) == 0
    dc_gpu = dq2.cpu().numpy()[:tpl.n].T
    db_gpu = _db_from_dA(tpl, dA2.cpu().numpy(), B)
    brows = tpl.b_idx                                                 # the boundary carries db only for the structural entries of b

    def err_against(g):
        return np.maximum(np.abs(dc_gpu - g["dc"]).max(axis=1) / (1 + np.abs(g["dc"]).max(axis=1)),
                          np.abs(db_gpu[:, brows] - g["db"][:, brows]).max(axis=1) / (1 + np.abs(g["db"]).max(axis=1)))
    # Same recurrences on the same system: agreement to rounding (1e-15) on most instances.  On the ill-conditioned ones (near-degenerate faces: singular values of
    # M^T far below the rest) the components along the near-null directions never converge to working precision on EITHER side, and the two implementations'
    # summation orders separate them by up to 3e-3 -- the accuracy LSQR itself has there (the oracle run with atol = btol = 1e-13 moves by as much).
    el = err_against(gl)
    def dist(g1, g2):
        return np.maximum(np.abs(g1["dc"] - g2["dc"]).max(axis=1) / (1 + np.abs(g2["dc"]).max(axis=1)),
                          np.abs(g1["db"][:, brows] - g2["db"][:, brows]).max(axis=1) / (1 + np.abs(g2["db"]).max(axis=1)))
    own = lsqr_own_movement(lambda **kw: oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], ones, zeros, mode="lsqr", **kw), gl, dist)
    assert np.median(el) < 1e-9, el
    assert_lsqr_agreement_per_instance(el, own)          # per instance: 1e-5, or explained by the oracle's own movement under a tighter rule
    # the same recurrences and stopping tests: the iteration counts are the oracle's (a loose bound on the gradients must not hide a different stopping point)
    li = eng.last_lsqr_iters.cpu().numpy().astype(int)
    assert (np.abs(li - gl["lsqr_iters"]) <= 0.05 * gl["lsqr_iters"] + 3).mean() >= 0.9, (li, gl["lsqr_iters"])
    v = ref["y"] - ref["s"]
    n_act = (v[:, 1:501] > 0).sum(axis=1) + 1 + 51                  # active bounds + budget row + the SOC rows (dual in the interior)
    regular = n_act >= tpl.n
    gd = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], ones, zeros, mode="dense")
    assert regular.mean() > 0.8 and err_against(gd)[regular].max() < 5e-4, (regular.mean(), err_against(gd)[regular].max())


@pytest.mark.parametrize("ni", [2, 3])
def test_lsqr_adjoint_with_several_instances_per_workgroup_equals_the_one_instance_kernel(monkeypatch, ni):
    """k_sa_lsqr_mi (ce_shared_a_mi.h, opt-in with CE_SA_LSQR_NI: NI instances of a shared-A template per workgroup, one stream over A_d^T for all of them) runs
    k_sa_lsqr's recurrences on the same system: same gradients, same iteration counts (up to the order of its sums).  B = 7 leaves a sub-group without an instance
    in the last workgroup; a smaller portfolio template than config 5 keeps the oracle out of it (k_sa_lsqr itself is pinned on the oracle above)."""
    B = 7
    A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=3, nw=60, kf=9)
    Ab = np.broadcast_to(A, (B,) + A.shape).copy(); bb = np.broadcast_to(b, (B,) + b.shape).copy()
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    A_eval, q_eval = tpl.values_from_dense(Ab, bb, c)
    monkeypatch.setenv("CE_CONST_A", "1")          # (a template this small would take the per-instance kernels)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    x, y, s, it, status, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-7, max_iters=100000)))
    assert eng.last_path == "const_a" and (status.cpu().numpy() == 1).all()
    dx = torch.from_numpy(np.random.default_rng(1).standard_normal((B, tpl.n))).cuda(); dy = torch.zeros_like(y)
    monkeypatch.delenv("CE_SA_LSQR_NI", raising=False)
    # (at the tight rule: under diffcp's 1e-8 the two stop an iteration or two apart and differ by what LSQR still has to gain there, ~1e-6)
    dA1, dq1, adj1 = eng.vjp(A_bm, x, y, s, dx, dy, path="const_a", q_eval=q_t, lsqr=TIGHT_LSQR)
    it1 = eng.last_lsqr_iters.cpu().numpy().astype(int)
    monkeypatch.setenv("CE_SA_LSQR_NI", str(ni))
    dA2, dq2, adj2 = eng.vjp(A_bm, x, y, s, dx, dy, path="const_a", q_eval=q_t, lsqr=TIGHT_LSQR)
    it2 = eng.last_lsqr_iters.cpu().numpy().astype(int)
    assert (adj1.cpu().numpy() == adj2.cpu().numpy()).all()
    assert np.abs(it1 - it2).max() <= 2 + 0.05 * it1.max(), (it1, it2)
    for a1, a2 in ((dA1, dA2), (dq1, dq2)):
        a1, a2 = a1.cpu().numpy(), a2.cpu().numpy()
        assert np.abs(a1 - a2).max() <= 1e-11 * (1 + np.abs(a1).max()), np.abs(a1 - a2).max()
