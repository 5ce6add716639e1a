"""The search-free adjoint kernel k_backward_ns (csrc/ce_backward_ns.h; round 6): null-space elimination of the equality rows by one wave + the reduced Hessian
formed and swept on the matrix cores, no pivot search.  It serves every ce_vjp call whose LSQR re-solve is armed (q_vals given) on plain-cone templates -- the
plugin's default path.  Checked against
  * the oracle's dense elimination of the full M^T (diffcp's dense mode) at tight eps, 1e-5 relative like the pivoting kernel's parity tests;
  * the pivoting kernel k_backward_rt on the same inputs (regular instances: the same unique solution, to 1e-6);
  * degenerate instances (more active rows than the null space leaves free, duplicated rows): flagged by the search-free kernel, re-solved by LSQR, equal to the
    oracle's LSQR mode."""
import numpy as np
import pytest
import torch

from cvxpylayers_amd import _lib
from cvxpylayers_amd import problems as P
from kit import TIGHT_LSQR
from test_gpu_parity import gpu_solve

pytestmark = pytest.mark.gpu


def _boundary(tpl, g, n):
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    B = g["dA"].shape[0]
    want = np.empty((tpl.nnz_aug, B))
    for k in range(tpl.nnz_aug):
        i, j = tpl.indices[k], cols[k]
        want[k] = -g["dA"][:, i, j] if j < n else g["db"][:, i]
    return want


def _run(n, cones, B, seed, eps=1e-9, max_iters=200000, expect_variant=None):
    from oracle import oracle
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=seed)
    ref = oracle.solve_batch(A, b, c, cones, eps=eps, max_iters=max_iters)
    ok = ref["status"] == 1
    assert ok.mean() > 0.9
    eng, A_bm, *_ = gpu_solve(tpl, A, b, c, eps=eps, max_iters=max_iters)
    v = _lib.lib().ce_adjoint_ns_variant(eng._h)
    assert v >= 0 and (expect_variant is None or v == expect_variant), v
    rng = np.random.default_rng(seed + 1)
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dxt, dyt = torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda()
    _, q_eval = tpl.values_from_dense(A, b, c); q_t = torch.from_numpy(q_eval).cuda()
    dA_ns, dq_ns, adj_ns = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance", lsqr=TIGHT_LSQR, q_eval=q_t)        # search-free kernel + LSQR re-solve of what it flags
    dA_rt, dq_rt, adj_rt = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_dense")                               # the pivoting kernel alone
    torch.cuda.synchronize()
    a_ns, a_rt = adj_ns.cpu().numpy(), adj_rt.cpu().numpy()
    gd = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="dense")
    want = _boundary(tpl, gd, n)
    got_ns, got_rt = dA_ns.cpu().numpy(), dA_rt.cpu().numpy()
    sc = 1 + np.abs(want).max(axis=0)
    reg = ok & (a_ns == 0) & (a_rt == 0)
    return dict(tpl=tpl, A=A, b=b, c=c, ref=ref, ok=ok, a_ns=a_ns, a_rt=a_rt, reg=reg, want=want, got_ns=got_ns, got_rt=got_rt, sc=sc, dq_ns=dq_ns.cpu().numpy(), dq_rt=dq_rt.cpu().numpy(),
                gd=gd, dx=dx, dy=dy, oracle=oracle, n=n, cones=cones)


def _check_regular(r, min_regular=0.8):
    reg = r["reg"]
    assert reg.mean() >= min_regular, (reg.mean(), np.bincount(r["a_ns"]), np.bincount(r["a_rt"]))
    e_ns = (np.abs(r["got_ns"] - r["want"]).max(axis=0) / r["sc"])[reg]
    e_rt = (np.abs(r["got_rt"] - r["want"]).max(axis=0) / r["sc"])[reg]
    assert e_ns.max() < 1e-5, (e_ns.max(), e_rt.max())
    assert np.median(e_ns) < 1e-8, (np.median(e_ns), np.median(e_rt))
    both = (np.abs(r["got_ns"] - r["got_rt"]).max(axis=0) / r["sc"])[reg]
    assert both.max() < 1e-6, both.max()
    n = r["n"]
    edq = np.abs(r["dq_ns"][:n].T - r["gd"]["dc"]).max(axis=1) / (1 + np.abs(r["gd"]["dc"]).max(axis=1))
    assert edq[reg].max() < 1e-5 and (r["dq_ns"][n] == 0).all()
    # what the search-free kernel flags (and LSQR re-solves) the pivoting kernel flags too, up to borderline pivots
    assert ((r["a_ns"] & 4) != 0).sum() <= ((r["a_rt"] & 4) != 0).sum() + max(2, int(0.02 * len(reg))), (np.bincount(r["a_ns"]), np.bincount(r["a_rt"]))


def _check_flagged_equal_oracle_lsqr(r):
    fl = (r["a_ns"] & 8) != 0
    if not fl.any():
        return
    assert ((r["a_ns"][fl] & 3) == 0).all()          # LSQR converged under the tight rule
    o = r["oracle"]
    idx = np.nonzero(fl)[0]
    gl = o.adjoint_batch(r["A"][idx], r["b"][idx], r["c"][idx], r["cones"], r["ref"]["x"][idx], r["ref"]["y"][idx], r["ref"]["s"][idx], r["dx"][idx], r["dy"][idx], mode="lsqr",
                         lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    want = _boundary(r["tpl"], gl, r["n"])
    el = np.abs(r["got_ns"][:, idx] - want).max(axis=0) / (1 + np.abs(want).max(axis=0))
    assert np.median(el) < 1e-6 and el.max() < 5e-3, el


def test_metric_configuration_search_free_adjoint():
    cfg = P.CONFIGS["M"]
    r = _run(cfg["n"], cfg["cones"], 192, seed=3, expect_variant=1)
    _check_regular(r, min_regular=0.95)
    _check_flagged_equal_oracle_lsqr(r)


def test_small_mixed_and_ragged_cones_search_free_adjoint():
    for n, cones, B, seed in ((12, {"z": 2, "l": 6, "q": [4, 5]}, 48, 1), (8, {"z": 4, "l": 6, "q": [4]}, 32, 2), (20, {"z": 3, "l": 10, "q": [3, 7, 2, 5, 1]}, 48, 4),
                              (25, {"z": 0, "l": 0, "q": [6, 6, 6, 6, 6, 6]}, 32, 5)):
        r = _run(n, cones, B, seed)
        _check_regular(r)
        _check_flagged_equal_oracle_lsqr(r)


def test_config3_search_free_adjoint_on_the_512_thread_variant():
    cfg = P.CONFIGS["C3"]
    r = _run(cfg["n"], cfg["cones"], 48, seed=7, expect_variant=2)
    _check_regular(r, min_regular=0.9)
    _check_flagged_equal_oracle_lsqr(r)


def test_lp_vertices_are_flagged_or_solved_and_always_equal_diffcps_element():
    """nonneg-only programs: the solution is a vertex, n active rows, H = 0 -- the null space is empty (nf = 0) when the vertex is non-degenerate; degenerate
    vertices (more active rows than variables) are rank deficient by counting and go to LSQR"""
    n, cones = 10, {"z": 0, "l": 30, "q": []}
    r = _run(n, cones, 64, seed=9, eps=1e-10)
    reg = r["reg"]
    assert reg.mean() > 0.5
    e_ns = (np.abs(r["got_ns"] - r["want"]).max(axis=0) / r["sc"])[reg]
    assert e_ns.max() < 1e-5, e_ns.max()
    _check_flagged_equal_oracle_lsqr(r)


def test_duplicated_equality_rows_are_flagged_by_the_row_elimination_and_resolved():
    from oracle import oracle
    n, cones, B = 12, {"z": 4, "l": 8, "q": [5]}, 24
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=11)
    deg = np.arange(B) % 2 == 0
    A[deg, 2, :] = A[deg, 0, :]; b[deg, 2] = b[deg, 0]
    ref = oracle.solve_batch(A, b, c, cones, eps=1e-10, max_iters=200000)
    keep = ref["status"] == 1                          # (a random draw may be unbounded: those instances are left out)
    assert keep.mean() > 0.8
    A, b, c, deg = A[keep], b[keep], c[keep], deg[keep]
    ref = {k: v[keep] for k, v in ref.items()}
    eng, A_bm, *_ = gpu_solve(tpl, A, b, c, eps=1e-10, max_iters=200000)
    assert _lib.lib().ce_adjoint_ns_variant(eng._h) >= 0
    rng = np.random.default_rng(12)
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    _, q_eval = tpl.values_from_dense(A, b, c)
    dA, dq, adj = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), path="per_instance", lsqr=TIGHT_LSQR, q_eval=torch.from_numpy(q_eval).cuda())
    torch.cuda.synchronize()
    a = adj.cpu().numpy()
    assert (a[deg] == 12).all() and (a[~deg] == 0).all(), a
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr", lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
    want = _boundary(tpl, g, n)
    assert np.abs(dA.cpu().numpy() - want).max() < 1e-6 * (1 + np.abs(want).max())


def test_random_templates_sparse_patterns_and_edge_shapes_against_the_pivoting_kernel():
    """A sweep over template shapes the fixed cases above do not reach: sparse structural patterns (empty columns of B, rows with one entry), templates without
    second-order cones (no weighted rows: the reduced Hessian is empty or zero), without nonnegative rows, a single variable, n at the edge of a tile variant.
    On every instance the search-free path (+ LSQR re-solve) must either agree with the pivoting kernel (both unflagged: one solution) or be flagged and equal the
    oracle's LSQR mode; the flags of the two eliminations may differ only on borderline pivots."""
    from oracle import oracle
    rng = np.random.default_rng(2024)
    shapes = [(1, {"z": 0, "l": 3, "q": []}, 1.0), (5, {"z": 1, "l": 4, "q": [3]}, 0.6), (9, {"z": 0, "l": 0, "q": [4, 3, 5]}, 0.7), (16, {"z": 3, "l": 12, "q": []}, 0.5),
              (28, {"z": 2, "l": 10, "q": [6, 9, 2]}, 0.4), (29, {"z": 0, "l": 16, "q": [8, 8]}, 1.0), (59, {"z": 4, "l": 30, "q": [12, 12, 7]}, 0.3), (60, {"z": 0, "l": 40, "q": [10] * 4}, 1.0)]
    for n, cones, dens in shapes:
        m = P.cone_rows(cones)
        pat = rng.random((m, n)) < dens
        pat[np.arange(m), rng.integers(0, n, m)] = True                 # no empty row
        pat[rng.integers(0, m, n), np.arange(n)] = True                 # no empty column
        tpl = P.dense_template(n, cones, pattern=pat)
        B = 24
        A, b, c = P.generate(n, cones, B, seed=int(rng.integers(1 << 30)))
        A = A * pat[None]
        ref = oracle.solve_batch(A, b, c, cones, eps=1e-9, max_iters=200000)
        ok = ref["status"] == 1
        if ok.sum() < 4:
            continue
        A, b, c = A[ok], b[ok], c[ok]; ref = {k: v[ok] for k, v in ref.items()}
        eng, A_bm, *_ = gpu_solve(tpl, A, b, c, eps=1e-9, max_iters=200000)
        if _lib.lib().ce_adjoint_ns_variant(eng._h) < 0:
            continue
        dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
        xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
        dxt, dyt = torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda()
        _, q_eval = tpl.values_from_dense(A, b, c); q_t = torch.from_numpy(q_eval).cuda()
        dA_ns, dq_ns, adj_ns = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance", lsqr=TIGHT_LSQR, q_eval=q_t)
        dA_rt, dq_rt, adj_rt = eng.vjp(A_bm, xr, yr, sr, dxt, dyt, path="per_instance_dense")
        torch.cuda.synchronize()
        a_ns, a_rt = adj_ns.cpu().numpy(), adj_rt.cpu().numpy()
        got_ns, got_rt = dA_ns.cpu().numpy(), dA_rt.cpu().numpy()
        assert np.isfinite(got_ns).all() and np.isfinite(dq_ns.cpu().numpy()).all(), (n, cones)
        sc = 1 + np.abs(got_rt).max(axis=0)
        reg = (a_ns == 0) & (a_rt == 0)
        if reg.any():
            d = (np.abs(got_ns - got_rt).max(axis=0) / sc)[reg]
            assert d.max() < 1e-6, (n, cones, d.max())
        assert ((a_ns & 3) == 0).all(), (n, cones, a_ns)                  # nothing left without a gradient, every LSQR converged under the tight rule
        fl = (a_ns & 8) != 0
        if fl.any():
            idx = np.nonzero(fl)[0]
            gl = oracle.adjoint_batch(A[idx], b[idx], c[idx], cones, ref["x"][idx], ref["y"][idx], ref["s"][idx], dx[idx], dy[idx], mode="lsqr",
                                      lsqr_atol=TIGHT_LSQR[0], lsqr_btol=TIGHT_LSQR[1], lsqr_iter_lim=TIGHT_LSQR[2])
            want = _boundary(tpl, gl, n)
            el = np.abs(got_ns[:, idx] - want).max(axis=0) / (1 + np.abs(want).max(axis=0))
            assert el.max() < 5e-3 and np.median(el) < 1e-5, (n, cones, el)
        # instances only ONE of the two eliminations flags sit on borderline pivots: rare
        assert ((a_ns & 4) != (a_rt & 4)).sum() <= max(2, B // 6), (n, cones, a_ns, a_rt)
