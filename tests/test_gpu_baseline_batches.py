"""The BASELINE.json configurations at their STATED batch sizes, and the bench's exact solver settings against the oracle
(VERDICT round 3, items 1a / 1b):

  * config M with bench.py's solver_args (eps 1e-4, acceleration_lookback 10 -- SCS's default --, max_iters 10000) against the
    oracle run with the SAME arguments (a ten-pair history there, the engine's one-pair history here): status, (x, y, s),
    and the histogram of the per-instance iteration counts;
  * C3 at B = 4096, C4 at B = 1024, C5 at B = 16384: the KKT certificate of EVERY instance (primal / dual residual, gap, cone
    membership of s and y, complementarity), a random subset of 48 instances against the oracle (solution and adjoint), and the
    kernel the shared-A configurations actually ran on.
The oracle needs minutes for whole batches of these sizes, hence the subsets; the certificates are size-independent properties.
Tolerances: solutions 1e-6 (1 + |.|_inf) at eps 1e-8, KKT residuals 50 eps, gradients 1e-5 relative."""
import numpy as np
import pytest
import torch

from cvxpylayers_amd import problems as P
from kit import TIGHT_LSQR, lsqr_own_movement, assert_lsqr_agreement_per_instance

pytestmark = pytest.mark.gpu


def _rel_rows(a, b):
    return np.abs(a - b).max(axis=1) / (1.0 + np.abs(b).max(axis=1))


def _engine(tpl, cones):
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine
    return ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))


def _db_from_dA(tpl, dA_np, B):
    cols = np.repeat(np.arange(tpl.n + 1), np.diff(tpl.indptr))
    db = np.zeros((B, tpl.m))
    for kk in np.nonzero(cols == tpl.n)[0]:
        db[:, tpl.indices[kk]] = dA_np[kk]
    return db


def _cone_violation(v, cones, dual=False):
    """largest violation of v in K (or K*) over the batch: zero rows (primal: |v|, dual: free), nonneg rows, SOC blocks, PSD blocks"""
    off = 0; worst = 0.0
    z = cones.get("z", 0)
    if z and not dual:
        worst = max(worst, float(v[:, :z].abs().max()))
    off += z
    l = cones.get("l", 0)
    if l:
        worst = max(worst, float((-v[:, off:off + l]).max()))
    off += l
    for d in cones.get("q", []):
        blk = v[:, off:off + d]
        worst = max(worst, float((torch.linalg.vector_norm(blk[:, 1:], dim=1) - blk[:, 0]).max()))
        off += d
    for k in cones.get("s", []):
        d = k * (k + 1) // 2
        S = torch.from_numpy(P.svec_to_sym(v[:, off:off + d].cpu().numpy(), k)).to(v.device)
        worst = max(worst, float((-torch.linalg.eigvalsh(S)[:, 0]).max()))
        off += d
    return worst


def _kkt_all(A_t, b_t, c_t, x, y, s, cones, eps, shared, P_t=None):
    """KKT certificate of every instance, on the GPU (A_t: (m, n) shared or (B, m, n)); P_t: (n, n) shared quadratic objective 1/2 x^T P x"""
    Ax = x @ A_t.T if shared else torch.einsum("bij,bj->bi", A_t, x)
    ATy = y @ A_t if shared else torch.einsum("bij,bi->bj", A_t, y)
    Px = x @ P_t if P_t is not None else torch.zeros_like(x)
    rp = (Ax + s - b_t).abs().amax(dim=1) / (1 + b_t.abs().amax(dim=-1))
    rd = (Px + ATy + c_t).abs().amax(dim=1) / (1 + c_t.abs().amax(dim=1))
    ctx = (c_t * x).sum(dim=1); bty = (b_t * y).sum(dim=1)
    gap = ((Px * x).sum(dim=1) + ctx + bty).abs() / (1 + ctx.abs())
    assert float(rp.max()) < 50 * eps and float(rd.max()) < 50 * eps and float(gap.max()) < 50 * eps, (float(rp.max()), float(rd.max()), float(gap.max()))
    assert _cone_violation(s, cones) < 100 * eps and _cone_violation(y, cones, dual=True) < 100 * eps
    comp = (s * y).sum(dim=1).abs() / (1 + ctx.abs())
    assert float(comp.max()) < 100 * eps, float(comp.max())


# ------------------------------------------------------------------ 1a: the bench's exact settings
def test_metric_config_with_the_bench_settings_against_the_oracle_with_the_same_settings():
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    from oracle import oracle
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 256
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=0)                                   # bench.py's batch (first 256 instances of seed 0)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    eng = _engine(tpl, cones)
    args = dict(eps=1e-4, acceleration_lookback=10, max_iters=10000)           # bench.py solver_args
    x, y, s, iters, status, resid = eng.solve(torch.from_numpy(A_eval).cuda().t().contiguous(), torch.from_numpy(q_eval).cuda(), make_settings(dict(args)))
    assert eng.last_acceleration
    ref = oracle.solve_batch(A, b, c, cones, **args)                            # aa_mem = 10
    st = status.cpu().numpy(); it = iters.cpu().numpy().astype(int)
    assert (st == ref["status"]).all() and (st == 1).all()
    tol = 20 * 1e-4                                                             # both sides are eps-accurate solutions reached along different accelerated paths
    for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
        assert _rel_rows(got.cpu().numpy(), want).max() < tol, _rel_rows(got.cpu().numpy(), want).max()
    d = np.abs(it - ref["iters"])
    hist = np.bincount(d // 25, minlength=4)
    assert (d <= 25).mean() >= 0.95, hist                                       # the histogram, not the mean: at most one check interval apart
    assert abs(it.mean() - ref["iters"].mean()) < 2.5, (it.mean(), ref["iters"].mean())


# ------------------------------------------------------------------ config 2 at B = 4096, in its three forms
def test_C2N_native_box_qp_at_B4096():
    """BASELINE config 2 (i): the box QP with P = 2 F^T F inside the kernels (ce_solve_qp / ce_vjp_qp) at the stated batch: the KKT certificate of every
    instance, a 48-instance subset against the oracle's QP embedding (solution, iteration counts, dq / db / dP)."""
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    from oracle import oracle
    nx, B = 50, 4096
    An, bn, qn, Pm, pst, pv = P.native_box_qp_batch(nx, B, seed=0)
    cones = {"z": 0, "l": 2 * nx, "q": [], "s": []}
    tpl = P.dense_template(nx, cones, pattern=(An != 0))
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0), p_structure=pst[:2])
    assert eng.qp_native
    A1, _ = tpl.values_from_dense(An[None], bn[:1], qn[:1])
    cols = np.repeat(np.arange(nx + 1), np.diff(tpl.indptr))
    A_eval = np.repeat(A1, B, axis=1); A_eval[cols == nx] = bn[:, tpl.indices[cols == nx]].T          # A shared, b per instance
    q_eval = np.concatenate([qn.T, np.zeros((1, B))], axis=0)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    P_bm = torch.from_numpy(np.broadcast_to(pv, (B, len(pv))).copy()).cuda()
    eps = 1e-8
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=100000)), P_bm=P_bm)
    assert bool((status == 1).all()), torch.bincount(status + 10)
    _kkt_all(torch.from_numpy(An).cuda(), torch.from_numpy(bn).cuda(), torch.from_numpy(qn).cuda(), x, y, s, cones, eps, shared=True, P_t=torch.from_numpy(Pm).cuda())
    bn_t = torch.from_numpy(bn).cuda()
    assert float((x + bn_t[:, :nx]).min()) > -1e-6 and float((bn_t[:, nx:] - x).min()) > -1e-6          # the box itself: lo <= x <= hi  (b = (-lo, hi))
    dA, dq, adj, dP = eng.vjp(A_bm, x, y, s, torch.ones_like(x), torch.zeros_like(y), P_bm=P_bm)
    assert int(((adj & 3) != 0).sum()) == 0 and bool(torch.isfinite(dq).all()) and bool(torch.isfinite(dP).all())
    idx = np.random.default_rng(1).choice(B, 48, replace=False)
    Ab = np.broadcast_to(An, (48,) + An.shape).copy(); Pb = np.broadcast_to(Pm, (48, nx, nx)).copy()
    ref = oracle.solve_batch(Ab, bn[idx], qn[idx], cones, P=Pb, eps=eps, max_iters=100000)
    assert (ref["status"] == 1).all()
    xs, ys, ss = x.cpu().numpy()[idx], y.cpu().numpy()[idx], s.cpu().numpy()[idx]
    assert _rel_rows(xs, ref["x"]).max() < 1e-6 and _rel_rows(ys, ref["y"]).max() < 1e-6 and _rel_rows(ss, ref["s"]).max() < 1e-6
    assert np.abs(iters.cpu().numpy()[idx].astype(int) - ref["iters"]).max() <= 25
    # adjoint at the ORACLE's solutions (only the adjoint solves are compared)
    xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA2, dq2, adj2, dP2 = eng.vjp(A_bm[idx], xo, yo, so, torch.ones_like(xo), torch.zeros_like(yo), P_bm=P_bm[idx])
    assert int((adj2 != 0).sum()) == 0
    g = oracle.adjoint_batch(Ab, bn[idx], qn[idx], cones, ref["x"], ref["y"], ref["s"], np.ones_like(ref["x"]), np.zeros_like(ref["y"]), P=Pb, mode="dense")
    assert _rel_rows(dq2.cpu().numpy()[:nx].T, g["dc"]).max() < 1e-5
    assert _rel_rows(_db_from_dA(tpl, dA2.cpu().numpy(), 48), g["db"]).max() < 1e-5
    pc = np.repeat(np.arange(nx), np.diff(pst[1]))
    wantP = g["dP"][:, pst[0], pc] + np.where(pst[0] != pc, g["dP"][:, pc, pst[0]], 0.0)        # one stored entry stands for (i, j) and (j, i)
    assert _rel_rows(dP2.cpu().numpy(), wantP).max() < 1e-5


def test_C2Q_epigraph_box_qp_at_B4096():
    """BASELINE config 2 (ii): the same box QP in the SOC-epigraph form DIFFCP would see (n = 51, m = 152: 100 box rows + SOC(52)), B = 4096."""
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    from oracle import oracle
    nx, B = 50, 4096
    A, b, c, cones = P.box_qp_batch(nx, B, seed=0)
    tpl = P.dense_template(A.shape[2], cones, pattern=(A[0] != 0))
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    eng = _engine(tpl, cones)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    eps = 1e-8
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=100000)))
    assert eng.last_path == "per_instance" and bool((status == 1).all()), torch.bincount(status + 10)
    _kkt_all(torch.from_numpy(A[0]).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda(), x, y, s, cones, eps, shared=True)
    # the epigraph variable equals the quadratic it bounds, and both forms of config 2 have the same minimiser (same F, g, lo, hi as the native form)
    An, bn, qn, Pm, pst, pv = P.native_box_qp_batch(nx, B, seed=0)
    xq = x[:, :nx]; Pt = torch.from_numpy(Pm).cuda(); qt = torch.from_numpy(qn).cuda()
    obj_native = 0.5 * ((xq @ Pt) * xq).sum(dim=1) + (qt * xq).sum(dim=1)          # = |F x - g|^2 - |g|^2
    g_t = -0.5 * torch.from_numpy(b[:, 2 * nx + 2:]).cuda()                        # (box_qp_batch: the SOC rows carry b = -2 g)
    assert float((x[:, nx] - (obj_native + (g_t * g_t).sum(dim=1))).abs().max()) < 1e-5          # u = |F x - g|^2
    dA, dq, adj = eng.vjp(A_bm, x, y, s, torch.ones_like(x), torch.zeros_like(y), path="per_instance")
    assert int((adj != 0).sum()) == 0
    idx = np.random.default_rng(1).choice(B, 48, replace=False)
    ref = oracle.solve_batch(A[idx], b[idx], c[idx], cones, eps=eps, max_iters=100000)
    assert (ref["status"] == 1).all()
    xs, ys, ss = x.cpu().numpy()[idx], y.cpu().numpy()[idx], s.cpu().numpy()[idx]
    assert _rel_rows(xs, ref["x"]).max() < 1e-6 and _rel_rows(ys, ref["y"]).max() < 1e-6 and _rel_rows(ss, ref["s"]).max() < 1e-6
    assert np.abs(iters.cpu().numpy()[idx].astype(int) - ref["iters"]).max() <= 25
    xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA2, dq2, adj2 = eng.vjp(A_bm[idx], xo, yo, so, torch.ones_like(xo), torch.zeros_like(yo), path="per_instance")
    assert int((adj2 != 0).sum()) == 0
    g = oracle.adjoint_batch(A[idx], b[idx], c[idx], cones, ref["x"], ref["y"], ref["s"], np.ones_like(ref["x"]), np.zeros_like(ref["y"]), mode="dense")
    assert _rel_rows(dq2.cpu().numpy()[:tpl.n].T, g["dc"]).max() < 1e-5
    assert _rel_rows(_db_from_dA(tpl, dA2.cpu().numpy(), 48), g["db"]).max() < 1e-5


def test_C2_lp_form_at_B4096_statuses_are_the_oracles():
    """The nonneg-only random LP of problems.CONFIGS["C2"] at B = 4096 with scripts/bench_configs.py's settings (eps 1e-4, max_iters 20000).  The splitting
    crawls on a handful of these LPs: a few instances end "solved / inaccurate" at the iteration limit (profiles/r04/zzz_configs.json: solved 0.99902).  That
    is the ALGORITHM, not the engine: without acceleration the engine's status equals the oracle's on EVERY instance (same unsolved set, iteration counts a
    check interval apart); with the plugin's default acceleration the accelerated paths of the two implementations differ on such 10^4-iteration runs, so
    there the claim is: every instance either side leaves unsolved at 20000 iterations solves (status 1) on BOTH sides when the limit is raised tenfold."""
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    from oracle import oracle
    cfg = P.CONFIGS["C2"]; n, cones, B = cfg["n"], cfg["cones"], 4096
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=0)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    eng = _engine(tpl, cones)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    args = dict(eps=1e-4, max_iters=20000)
    # plain iteration: instance by instance the oracle's outcome
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, **args)))
    ref = oracle.solve_batch(A, b, c, cones, acceleration_lookback=0, **args)
    st = status.cpu().numpy(); it = iters.cpu().numpy().astype(int)
    assert (st == ref["status"]).all(), (np.nonzero(st != ref["status"])[0], st[st != ref["status"]], ref["status"][st != ref["status"]])
    assert set(np.unique(st)) <= {1, 2} and (st == 2).sum() == (ref["status"] == 2).sum() >= 1          # the unsolved ones exist on both sides, and are the same
    d = np.abs(it - ref["iters"])
    assert (d <= 25).mean() >= 0.99 and abs(it.mean() - ref["iters"].mean()) < 0.01 * ref["iters"].mean(), (np.bincount(np.minimum(d // 25, 8)), it.mean(), ref["iters"].mean())
    ok = (st == 1) & (it == ref["iters"])          # (instances that stopped at the same check: eps-accurate points of an LP a check apart can differ by more than 20 eps)
    assert ok.mean() > 0.98
    for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
        assert _rel_rows(got.cpu().numpy()[ok], want[ok]).max() < 20 * 1e-4
    # the plugin's default (one-pair Anderson acceleration) against the oracle with the same memory
    xa, ya, sa, iters_a, status_a, _ = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=1, **args)))
    refa = oracle.solve_batch(A, b, c, cones, acceleration_lookback=1, **args)
    sta = status_a.cpu().numpy()
    assert set(np.unique(sta)) <= {1, 2} and (sta == 2).sum() <= 16 and (refa["status"] == 2).sum() <= 16
    assert abs(iters_a.float().mean().item() - refa["iters"].mean()) < 0.02 * refa["iters"].mean()
    hard = np.nonzero((sta == 2) | (refa["status"] == 2))[0]
    if len(hard):
        long_args = dict(eps=1e-4, max_iters=200000, acceleration_lookback=1)
        xh, yh, sh, ih, sth, _ = eng.solve(A_bm[hard].contiguous(), q_t[:, hard].contiguous(), make_settings(dict(long_args)))
        refh = oracle.solve_batch(A[hard], b[hard], c[hard], cones, **long_args)
        assert bool((sth == 1).all()) and (refh["status"] == 1).all(), (sth.cpu().numpy(), refh["status"])
        assert _rel_rows(xh.cpu().numpy(), refh["x"]).max() < 50 * 1e-4


# ------------------------------------------------------------------ 1b: C3 / C4 / C5 at their stated batch
def test_C3_socp_n100_at_B4096():
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    from oracle import oracle
    cfg = P.CONFIGS["C3"]; n, cones, B = cfg["n"], cfg["cones"], 4096
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=3)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    eng = _engine(tpl, cones)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    eps = 1e-8
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=50000)))
    assert eng.last_path == "per_instance" and bool((status == 1).all())
    _kkt_all(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda(), x, y, s, cones, eps, shared=False)
    dA, dq, adj = eng.vjp(A_bm, x, y, s, torch.ones_like(x), torch.zeros_like(y), path="per_instance")
    assert int((adj != 0).sum()) == 0
    idx = np.random.default_rng(1).choice(B, 48, replace=False)
    ref = oracle.solve_batch(A[idx], b[idx], c[idx], cones, eps=eps, max_iters=50000)
    assert (ref["status"] == 1).all()
    xs, ys, ss = x.cpu().numpy()[idx], y.cpu().numpy()[idx], s.cpu().numpy()[idx]
    assert _rel_rows(xs, ref["x"]).max() < 1e-6 and _rel_rows(ys, ref["y"]).max() < 1e-6 and _rel_rows(ss, ref["s"]).max() < 1e-6
    assert np.abs(iters.cpu().numpy()[idx].astype(int) - ref["iters"]).max() <= 25
    g = oracle.adjoint_batch(A[idx], b[idx], c[idx], cones, ref["x"], ref["y"], ref["s"], np.ones_like(ref["x"]), np.zeros_like(ref["y"]), mode="dense")
    dq_np = dq.cpu().numpy(); dA_np = dA.cpu().numpy()
    assert _rel_rows(dq_np[:n].T[idx], g["dc"]).max() < 1e-5
    assert _rel_rows(_db_from_dA(tpl, dA_np, B)[idx], g["db"]).max() < 1e-5
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr)); ka = np.nonzero(cols < n)[0]
    assert _rel_rows(-dA_np[ka].T[idx], g["dA"][:, tpl.indices[ka], cols[ka]]).max() < 1e-5      # dA_eval = -dA.data (diffcp_if.py:91)


def test_C4_sdp_20x20_at_B1024():
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    from oracle import oracle
    B = 1024
    A, b, c, cones, tpl = P.sdp_c4_batch(B, seed=0)
    A_eval, q_eval = tpl.values_from_dense(np.broadcast_to(A, (B,) + A.shape).copy(), b, c)          # A shared, b and c per instance
    eng = _engine(tpl, cones)
    A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
    eps = 1e-8
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=50000)))
    assert eng.last_path == "const_a" and eng.last_const_a_kernel == "k_sa_fwd", (eng.last_path, getattr(eng, "last_const_a_kernel", None))
    assert bool((status == 1).all())
    _kkt_all(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c).cuda(), x, y, s, cones, eps, shared=True)
    dA, dq, adj = eng.vjp(A_bm, x, y, s, torch.ones_like(x), torch.zeros_like(y), path="const_a", lsqr=TIGHT_LSQR)
    assert int((adj != 0).sum()) == 0
    idx = np.random.default_rng(1).choice(B, 48, replace=False)
    Ab = np.broadcast_to(A, (48,) + A.shape).copy()
    ref = oracle.solve_batch(Ab, b[idx], c[idx], cones, eps=eps, max_iters=50000)
    assert (ref["status"] == 1).all()
    xs, ys, ss = x.cpu().numpy()[idx], y.cpu().numpy()[idx], s.cpu().numpy()[idx]
    assert _rel_rows(xs, ref["x"]).max() < 1e-6 and _rel_rows(ys, ref["y"]).max() < 1e-6 and _rel_rows(ss, ref["s"]).max() < 1e-6
    g = oracle.adjoint_batch(Ab, b[idx], c[idx], cones, ref["x"], ref["y"], ref["s"], np.ones_like(ref["x"]), np.zeros_like(ref["y"]), mode="dense")
    assert _rel_rows(dq.cpu().numpy()[:tpl.n].T[idx], g["dc"]).max() < 1e-5
    assert _rel_rows(_db_from_dA(tpl, dA.cpu().numpy(), B)[idx], g["db"]).max() < 1e-5


def test_C5_portfolio_n501_at_B16384():
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    from oracle import oracle
    B = 16384
    A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=0)
    A1, _ = tpl.values_from_dense(A[None], b[None], c[:1])           # A, b shared: tile one instance's values (the dense (B, m, n) array would be 36 GB)
    eng = _engine(tpl, cones)
    A_bm = torch.from_numpy(A1[:, 0]).cuda().repeat(B, 1).contiguous()
    q_t = torch.from_numpy(np.concatenate([c.T, np.zeros((1, B))], axis=0)).cuda()
    eps = 1e-6          # the splitting converges slowly on this LP-like program (4 000 - 9 000 iterations at 1e-6, > 1e5 at 1e-8, on both sides)
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=100000)))
    assert eng.last_path == "const_a" and eng.last_const_a_kernel == "k_sa_fwd", (eng.last_path, getattr(eng, "last_const_a_kernel", None))
    # (the splitting crawls on a handful of these LP-like instances: 2 of 16384 are still "solved / inaccurate" after 100 000 iterations -- on the
    # oracle as well; they are counted, not hidden, and everything below is asserted on the solved ones)
    ok = status == 1
    assert int(ok.sum()) >= B - 8 and bool(((status == 1) | (status == 2)).all()), torch.bincount(status + 10)
    A_t = torch.from_numpy(A).cuda(); b_t = torch.from_numpy(b).cuda(); c_t = torch.from_numpy(c).cuda()
    _kkt_all(A_t, b_t[None].expand(int(ok.sum()), -1), c_t[ok], x[ok], y[ok], s[ok], cones, eps, shared=True)
    # the portfolio's own statement of feasibility: budget, no short positions, the risk cone
    w, t = x[ok][:, :500], x[ok][:, 500]
    assert float((w.sum(dim=1) - 1).abs().max()) < 1e-4 and float(w.min()) > -1e-4
    assert float((torch.linalg.vector_norm(w @ (-A_t[502:, :500].T), dim=1) - t).max()) < 1e-4
    idx = np.sort(np.random.default_rng(1).choice(np.nonzero(ok.cpu().numpy())[0], 48, replace=False))
    Ab = np.broadcast_to(A, (48,) + A.shape).copy(); bb = np.broadcast_to(b, (48,) + b.shape).copy()
    ref = oracle.solve_batch(Ab, bb, c[idx], cones, eps=eps, max_iters=100000)
    assert (ref["status"] == 1).all()
    xs, ys, ss = x.cpu().numpy()[idx], y.cpu().numpy()[idx], s.cpu().numpy()[idx]
    assert _rel_rows(xs, ref["x"]).max() < 1e-6 and _rel_rows(ys, ref["y"]).max() < 1e-6 and _rel_rows(ss, ref["s"]).max() < 1e-6
    assert np.abs(iters.cpu().numpy()[idx].astype(int) - ref["iters"]).max() <= 25
    # adjoint of dx = 1 over the WHOLE batch at the engine's own solutions: finite and flag-free everywhere ...
    dA, dq, adj = eng.vjp(A_bm, x, y, s, torch.ones_like(x), torch.zeros_like(y), path="const_a", lsqr=TIGHT_LSQR)
    assert int((adj[ok] != 0).sum()) == 0 and bool(torch.isfinite(dq[:, ok]).all())
    # ... and, on the subset, at the ORACLE's solutions (so that only the adjoint solves are compared), against the oracle's LSQR mode -- diffcp's default and
    # its semantics -- at matched tolerances.  Since round 5 the kernel runs LSQR on diffcp's FULL (n + m + 1) system (tau row and column): the two are then
    # the minimum-norm solutions of the SAME system and agree to LSQR's accuracy on every instance, the ones on a degenerate face included (as many active
    # rows as variables minus one: the system is singular there, and the r_tau = 0 reduction of rounds 1-4 returned a different element: 1e-3 .. 5e-3 apart).
    # The oracle's DENSE elimination pins free variables to zero instead: it agrees at vertices only, and only as far as the eps-accurate point makes M singular
    # along z (2e-5 .. 8e-5 at eps 1e-6), so it is compared loosely and on the regular instances only.
    xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA2, dq2, adj2 = eng.vjp(A_bm[idx], xo, yo, so, torch.ones_like(xo), torch.zeros_like(yo), path="const_a", lsqr=TIGHT_LSQR, q_eval=q_t[:, idx].contiguous())
    assert int((adj2 != 0).sum()) == 0
    dc_gpu = dq2.cpu().numpy()[:tpl.n].T
    db_gpu = _db_from_dA(tpl, dA2.cpu().numpy(), 48)
    brows = tpl.b_idx

    def err_against(gg):
        return np.maximum(np.abs(dc_gpu - gg["dc"]).max(axis=1) / (1 + np.abs(gg["dc"]).max(axis=1)),
                          np.abs(db_gpu[:, brows] - gg["db"][:, brows]).max(axis=1) / (1 + np.abs(gg["db"]).max(axis=1)))
    ones, zeros = np.ones_like(ref["x"]), np.zeros_like(ref["y"])
    gl = oracle.adjoint_batch(Ab, bb, c[idx], cones, ref["x"], ref["y"], ref["s"], ones, zeros, mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    v = ref["y"] - ref["s"]
    n_act = (v[:, 1:501] > 0).sum(axis=1) + 1 + 51                  # active bounds + budget row + the SOC rows (dual in the interior)
    regular = n_act >= tpl.n
    el = err_against(gl)
    # Same recurrences on the same system: rounding-level agreement (1e-15) on most instances; on ill-conditioned ones (near-degenerate faces) the components along
    # near-null directions converge on neither side and the implementations' summation orders separate them -- LSQR's own accuracy there (tests/test_gpu_atsize.py)
    # PER-INSTANCE rule (round 6; was "95 % below 5e-3"): every instance is within 1e-5 of the oracle, or no further from it than 3 x what the oracle's OWN answer
    # moves when its stopping rule is tightened once more (kit.assert_lsqr_agreement_per_instance)
    def dist(g1, g2):
        return np.maximum(np.abs(g1["dc"] - g2["dc"]).max(axis=1) / (1 + np.abs(g2["dc"]).max(axis=1)),
                          np.abs(g1["db"][:, brows] - g2["db"][:, brows]).max(axis=1) / (1 + np.abs(g2["db"]).max(axis=1)))
    own = lsqr_own_movement(lambda **kw: oracle.adjoint_batch(Ab, bb, c[idx], cones, ref["x"], ref["y"], ref["s"], ones, zeros, mode="lsqr", **kw), gl, dist)
    assert np.median(el) < 1e-9, el
    assert_lsqr_agreement_per_instance(el, own)
    gd = oracle.adjoint_batch(Ab, bb, c[idx], cones, ref["x"], ref["y"], ref["s"], ones, zeros, mode="dense")
    assert err_against(gd)[regular].max() < 5e-4, (regular.mean(), err_against(gd)[regular].max())
    # diffcp's own stopping rule (the plugin's default: atol = btol = 1e-8, 2 (n + m + 1) iterations) on both sides: the same element to LSQR's accuracy at that rule
    dA3, dq3, adj3 = eng.vjp(A_bm[idx], xo, yo, so, torch.ones_like(xo), torch.zeros_like(yo), path="const_a", q_eval=q_t[:, idx].contiguous())
    gdef = oracle.adjoint_batch(Ab, bb, c[idx], cones, ref["x"], ref["y"], ref["s"], ones, zeros, mode="lsqr")
    assert int((adj3 != 0).sum()) == 0
    e3 = np.abs(dq3.cpu().numpy()[:tpl.n].T - gdef["dc"]).max(axis=1) / (1 + np.abs(gdef["dc"]).max(axis=1))
    li = eng.last_lsqr_iters.cpu().numpy().astype(int)
    assert np.median(e3) < 1e-4 and e3.max() < 5e-3, e3          # (both stopped at atol = btol = 1e-8: the answers are converged to about 1e-5, and a summation order apart)
    # the same recurrences, the same stopping tests: the same number of iterations up to what rounding does to LSQR's residual estimates late in a 450-iteration run (a few per cent)
    assert (np.abs(li - gdef["lsqr_iters"]) <= 0.05 * gdef["lsqr_iters"] + 3).mean() >= 0.9 and abs(li.mean() - gdef["lsqr_iters"].mean()) < 0.03 * gdef["lsqr_iters"].mean(), (li, gdef["lsqr_iters"])
