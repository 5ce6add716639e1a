"""Quadratic objectives 1/2 x^T P x at the plugin boundary (reference: P_eval for plugins in SUPPORTS_QUAD_OBJ, _quad_form_dpp.py:32;
tests/test_torch.py:790-1370).  The MI355 plugin reduces them to an epigraph SOC over the Cholesky factor of P with differentiable
torch ops (QuadEpigraph).  CPU: the reduction itself, solved by the oracle, against closed forms.  GPU: the whole plugin, values
and gradients with respect to P, q, A, b against autograd through the KKT system."""
import numpy as np
import pytest
import torch

from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, QuadEpigraph, dims_to_solver_dict


def _upper_structure(n):
    rows, ptr = [], [0]
    for j in range(n):
        rows.extend(range(j + 1)); ptr.append(len(rows))
    return np.asarray(rows, dtype=np.int32), np.asarray(ptr, dtype=np.int32), (n, n)


def _p_values(Pm, struct):
    idx, ptr, _ = struct
    cols = np.repeat(np.arange(Pm.shape[-1]), np.diff(ptr))
    return Pm[..., idx, cols]                      # (B, nnzP)


def _eq_qp(n, p, B, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((B, n, n)); Pm = G @ G.transpose(0, 2, 1) / n + 0.5 * np.eye(n)
    q = rng.standard_normal((B, n)); F = rng.standard_normal((B, p, n)); g = rng.standard_normal((B, p))
    return Pm, q, F, g


def _kkt_solution(Pm, q, F, g):
    B, n = q.shape; p = g.shape[1]
    K = np.zeros((B, n + p, n + p)); K[:, :n, :n] = Pm; K[:, :n, n:] = F.transpose(0, 2, 1); K[:, n:, :n] = F
    sol = np.linalg.solve(K, np.concatenate([-q, g], axis=1)[:, :, None])[:, :, 0]
    return sol[:, :n], sol[:, n:]


def test_epigraph_reduction_solved_by_the_oracle_matches_the_kkt_solution():
    from oracle import oracle
    n, p, B = 6, 2, 5
    Pm, q, F, g = _eq_qp(n, p, B, seed=0)
    cones = {"z": p, "l": 0, "q": [], "s": []}
    tpl = P.dense_template(n, cones)
    A_eval, q_eval = tpl.values_from_dense(F, g, q)              # F x + s = g, s in the zero cone
    pst = _upper_structure(n)
    qe = QuadEpigraph(pst, (tpl.indices, tpl.indptr), (tpl.m, n + 1), dims_to_solver_dict(cones))
    q_aug, A_aug = qe.assemble(torch.from_numpy(_p_values(Pm, pst).T.copy()), torch.from_numpy(q_eval), torch.from_numpy(A_eval))
    aug = P.ConeTemplate(n=n + 1, m=qe.m_aug, indices=qe.aug_indices, indptr=qe.aug_indptr, cones=qe.aug_cones)
    assert P.cone_rows(qe.aug_cones) == qe.m_aug and qe.aug_cones["q"] == [n + 2]
    Ad, bd, cd = aug.dense_from_values(A_aug.numpy(), q_aug.numpy())
    r = oracle.solve_batch(Ad, bd, cd, qe.aug_cones, eps=1e-10, max_iters=200000)
    assert (r["status"] == 1).all()
    xs, nus = _kkt_solution(Pm, q, F, g)
    np.testing.assert_allclose(r["x"][:, :n], xs, atol=1e-6)
    np.testing.assert_allclose(r["x"][:, n], 0.5 * np.einsum("bi,bij,bj->b", xs, Pm, xs), atol=1e-6)    # t = 1/2 x^T P x at the optimum
    xo, yo = qe.split(torch.from_numpy(r["x"]), torch.from_numpy(r["y"]))
    np.testing.assert_allclose(yo.numpy(), nus, atol=1e-5)       # multipliers of F x = g, rows back in template order


def test_epigraph_keeps_the_row_order_of_later_cone_blocks():
    # SOC block inserted after the template's own SOCs and before PSD / exponential rows (SCS order z, l, q, s, ep, p)
    cones = {"z": 1, "l": 2, "q": [3], "s": [2], "ep": 1}
    n = 4
    tpl = P.dense_template(n, cones)
    qe = QuadEpigraph(_upper_structure(n), (tpl.indices, tpl.indptr), (tpl.m, n + 1), dims_to_solver_dict(cones))
    assert qe.r0 == 6 and qe.aug_cones["q"] == [3, n + 2] and qe.aug_cones["s"] == [2] and qe.aug_cones["ep"] == 1
    assert list(qe.dual_rows) == [0, 1, 2, 3, 4, 5] + [6 + n + 2 + k for k in range(3 + 3)]
    assert qe.m_aug == tpl.m + n + 2 and len(qe.aug_indptr) == n + 3


def test_indefinite_P_is_refused():
    n = 3
    cones = {"z": 1, "l": 0, "q": [], "s": []}
    tpl = P.dense_template(n, cones)
    pst = _upper_structure(n)
    qe = QuadEpigraph(pst, (tpl.indices, tpl.indptr), (tpl.m, n + 1), dims_to_solver_dict(cones))
    Pm = np.diag([1.0, -1.0, 1.0])[None]
    from cvxpylayers_amd.interfaces.mi355_if import SolverError
    with pytest.raises(SolverError, match="not positive semidefinite"):
        qe.assemble(torch.from_numpy(_p_values(Pm, pst).T.copy()), torch.zeros(n + 1, 1, dtype=torch.float64), torch.zeros(tpl.nnz_aug, 1, dtype=torch.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["native", "epigraph"])
def test_quadratic_objective_values_and_gradients_on_gpu(form, monkeypatch):
    from cvxpylayers_amd.interfaces.mi355_if import _CvxpyLayer
    if form == "epigraph":
        monkeypatch.setenv("CE_QP_EPIGRAPH", "1")
    n, p, B = 6, 2, 7
    Pm, q, F, g = _eq_qp(n, p, B, seed=1)
    cones = {"z": p, "l": 0, "q": [], "s": []}
    tpl = P.dense_template(n, cones)
    pst = _upper_structure(n)
    ctx = MI355_ctx(pst, tpl.problem_data_index, cones, options={"eps": 1e-10, "max_iters": 200000})
    dev = torch.device("cuda", 0)
    Pt = torch.from_numpy(Pm).to(dev).requires_grad_(); qt = torch.from_numpy(q).to(dev).requires_grad_()
    Ft = torch.from_numpy(F).to(dev).requires_grad_(); gt = torch.from_numpy(g).to(dev).requires_grad_()
    idx, ptr, _ = pst
    pcols = torch.from_numpy(np.repeat(np.arange(n), np.diff(ptr))).to(dev); prow = torch.from_numpy(idx.astype(np.int64)).to(dev)
    P_eval = Pt[:, prow, pcols].t()                                         # (nnzP, B): upper triangle, CSC order
    cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
    ar = torch.from_numpy(tpl.indices.astype(np.int64)).to(dev); ac = torch.from_numpy(cols).to(dev)
    aug = torch.cat([Ft, gt[:, :, None]], dim=2)                            # [A_cvx | b] with A_cvx = -A = ... solver form F x + s = g  ->  A_cvx = -F
    aug = torch.cat([-Ft, gt[:, :, None]], dim=2)
    A_eval = aug[:, ar, ac].t()
    q_eval = torch.cat([qt.t(), torch.zeros(1, B, dtype=torch.float64, device=dev)], dim=0)
    primal, dual, info, _ = _CvxpyLayer.apply(P_eval, q_eval, A_eval, ctx, {}, True, None)
    assert primal.shape == (B, n) and dual.shape == (B, p)
    assert ctx.engine(dev).qp_native and (ctx._aug_ctx is None) == (form == "native")
    wx = torch.linspace(0.5, 1.5, n, dtype=torch.float64, device=dev)
    (primal * wx).sum().backward()
    grads = [t.grad.clone() for t in (Pt, qt, Ft, gt)]
    # reference: autograd through the KKT solve
    P2, q2, F2, g2 = (t.detach().clone().requires_grad_() for t in (Pt, qt, Ft, gt))
    Ps = 0.5 * (P2 + P2.transpose(1, 2))
    K = torch.cat([torch.cat([Ps, F2.transpose(1, 2)], dim=2), torch.cat([F2, torch.zeros(B, p, p, dtype=torch.float64, device=dev)], dim=2)], dim=1)
    sol = torch.linalg.solve(K, torch.cat([-q2, g2], dim=1)[:, :, None])[:, :, 0]
    assert torch.allclose(primal, sol[:, :n].detach(), atol=1e-6) and torch.allclose(dual, sol[:, n:].detach(), atol=1e-5)
    (sol[:, :n] * wx).sum().backward()
    # only the upper triangle of P reaches the solver: compare the gradient folded onto it
    up = torch.triu(torch.ones(n, n, dtype=torch.float64, device=dev))
    gP_ref = (P2.grad + P2.grad.transpose(1, 2)) * up - torch.diag_embed(torch.diagonal(P2.grad, dim1=1, dim2=2))
    assert torch.allclose(grads[0] * up, gP_ref, atol=2e-5), (grads[0] * up - gP_ref).abs().max()
    for got, want in zip(grads[1:], (q2.grad, F2.grad, g2.grad)):
        assert torch.allclose(got, want, atol=2e-5), (got - want).abs().max()


@pytest.mark.gpu
def test_box_qp_with_a_native_quadratic_objective():
    # BASELINE config 2 in its native form: min 1/2 x^T P x + q^T x, 0 <= x <= 1 with P = 2 I, q = -2 t  ->  clip(t, 0, 1)
    from cvxpylayers_amd.interfaces.mi355_if import _CvxpyLayer
    n, B = 8, 16
    cones = {"z": 0, "l": 2 * n, "q": [], "s": []}
    A = np.concatenate([-np.eye(n), np.eye(n)], axis=0)                      # x >= 0: s = x ; x <= 1: s = 1 - x
    b = np.concatenate([np.zeros(n), np.ones(n)])
    tpl = P.dense_template(n, cones, pattern=(A != 0), b_pattern=(b != 0))
    rng = np.random.default_rng(0)
    t = rng.standard_normal((B, n)) * 1.5
    A_eval, q_eval = tpl.values_from_dense(np.broadcast_to(A, (B,) + A.shape).copy(), np.broadcast_to(b, (B, 2 * n)).copy(), -2 * t)
    pst = (np.arange(n, dtype=np.int32), np.arange(n + 1, dtype=np.int32), (n, n))          # diagonal structure
    ctx = MI355_ctx(pst, tpl.problem_data_index, cones, options={"eps": 1e-9, "max_iters": 100000})
    dev = torch.device("cuda", 0)
    P_eval = torch.full((n, B), 2.0, dtype=torch.float64, device=dev)
    primal, dual, info, _ = _CvxpyLayer.apply(P_eval, torch.from_numpy(q_eval).to(dev), torch.from_numpy(A_eval).to(dev), ctx, {}, False, None)
    np.testing.assert_allclose(primal.cpu().numpy(), np.clip(t, 0, 1), atol=1e-5)


@pytest.mark.gpu
def test_frontend_layer_with_a_parametric_quad_form():
    """quad_form(x, P) with P a parameter (reference tests/test_torch.py:858-900, gradcheck of q and Q) through the frontend:
    min 1/2 x^T P x + q^T x  s.t.  1^T x = 1, x >= 0 -- strictly positive optimum here, so the KKT reference has the equality only."""
    from cvxpylayers_amd.torch import CvxpyLayer, VariableRecovery
    from cvxpylayers_amd.torch.templates import template_from_affine_builder
    n = 5

    def builder(Pp, qp):
        A = np.zeros((1 + n, n)); b = np.zeros(1 + n)
        A[0] = 1.0; b[0] = 1.0                         # 1^T x = 1
        A[1:] = -np.eye(n)                             # x >= 0
        return A, b, qp, 0.5 * (Pp + Pp.T)
    tpl = template_from_affine_builder(builder, [(n, n), (n,)], {"z": 1, "l": n, "q": [], "s": []}, [VariableRecovery(slice(0, n), None, (n,))])
    assert tpl.P_map is not None and tpl.P_structure[2] == (n, n)
    layer = CvxpyLayer(template=tpl, solver_args={"eps": 1e-10, "max_iters": 200000})
    torch.manual_seed(3)
    G = torch.randn(4, n, n, dtype=torch.float64, device="cuda")
    Pt = (G @ G.transpose(1, 2) / n + torch.eye(n, dtype=torch.float64, device="cuda")).requires_grad_()
    qt = (0.1 * torch.randn(4, n, dtype=torch.float64, device="cuda")).requires_grad_()
    (x,) = layer(Pt, qt)
    assert x.shape == (4, n) and bool((x > 1e-3).all())
    wts = torch.linspace(1.0, 2.0, n, dtype=torch.float64, device="cuda")
    (x * wts).sum().backward()
    P2 = Pt.detach().clone().requires_grad_(); q2 = qt.detach().clone().requires_grad_()
    Ps = 0.5 * (P2 + P2.transpose(1, 2))
    ones = torch.ones(4, 1, n, dtype=torch.float64, device="cuda")
    K = torch.cat([torch.cat([Ps, ones.transpose(1, 2)], dim=2), torch.cat([ones, torch.zeros(4, 1, 1, dtype=torch.float64, device="cuda")], dim=2)], dim=1)
    sol = torch.linalg.solve(K, torch.cat([-q2, torch.ones(4, 1, dtype=torch.float64, device="cuda")], dim=1)[:, :, None])[:, :, 0]
    assert torch.allclose(x, sol[:, :n].detach(), atol=1e-6)
    (sol[:, :n] * wts).sum().backward()
    assert torch.allclose(qt.grad, q2.grad, atol=2e-5)
    assert torch.allclose(Pt.grad + Pt.grad.transpose(1, 2), P2.grad + P2.grad.transpose(1, 2), atol=4e-5)    # symmetric part is what the problem sees


@pytest.mark.gpu
def test_native_qp_kernels_match_the_oracle_on_box_qps():
    """BASELINE config 2 in its native form (P = 2 F^T F, 100 box rows, n = 50) through ce_solve_qp / ce_vjp_qp against the oracle's
    QP embedding: solutions, iteration counts, dA / dq / dP."""
    from oracle import oracle
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    nx, B = 50, 24
    rng = np.random.default_rng(0)
    Fm = rng.standard_normal((nx, nx)) / np.sqrt(nx); g = rng.standard_normal((B, nx))
    lo = -0.5 - 0.5 * rng.random((B, nx)); hi = 0.5 + 0.5 * rng.random((B, nx))
    Pn = np.broadcast_to(2 * Fm.T @ Fm, (B, nx, nx)).copy() * (1 + 0.1 * rng.random((B, 1, 1)))       # per-instance P
    An = np.broadcast_to(np.concatenate([-np.eye(nx), np.eye(nx)], axis=0), (B, 2 * nx, nx)).copy()
    bn = np.concatenate([-lo, hi], axis=1); qn = -2 * g @ Fm
    cones = {"z": 0, "l": 2 * nx, "q": [], "s": []}
    tpl = P.dense_template(nx, cones, pattern=(An[0] != 0))
    pst = _upper_structure(nx)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0), p_structure=pst[:2])
    assert eng.qp_native
    A_eval, q_eval = tpl.values_from_dense(An, bn, qn)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).cuda())
    P_bm = torch.from_numpy(_p_values(Pn, pst)).cuda().contiguous()
    for eps in (1e-4, 1e-9):
        ref = oracle.solve_batch(An, bn, qn, cones, P=Pn, eps=eps, max_iters=100000)
        x, y, s, iters, status, resid = eng.solve(A_bm, torch.from_numpy(q_eval).cuda(), make_settings(dict(acceleration_lookback=0, eps=eps, max_iters=100000)), P_bm=P_bm)
        assert (status.cpu().numpy() == 1).all() and (ref["status"] == 1).all()
        tol = max(1e-6, 20 * eps)
        for got, want in ((x, ref["x"]), (y, ref["y"]), (s, ref["s"])):
            err = np.abs(got.cpu().numpy() - want).max(axis=1) / (1 + np.abs(want).max(axis=1))
            assert err.max() < tol, err.max()
        assert np.abs(iters.cpu().numpy() - ref["iters"]).max() <= 25, (iters.cpu().numpy(), ref["iters"])
    dx = rng.standard_normal(ref["x"].shape); dy = rng.standard_normal(ref["y"].shape)
    gr = oracle.adjoint_batch(An, bn, qn, cones, ref["x"], ref["y"], ref["s"], dx, dy, P=Pn, mode="dense")
    xr, yr, sr = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
    dA, dq, adj, dP = eng.vjp(A_bm, xr, yr, sr, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), P_bm=P_bm)
    assert (adj.cpu().numpy() == 0).all()
    cols = np.repeat(np.arange(nx + 1), np.diff(tpl.indptr))
    want = np.stack([-gr["dA"][:, i, j] if j < nx else gr["db"][:, i] for i, j in zip(tpl.indices, cols)])
    assert np.abs(dA.cpu().numpy() - want).max() < 1e-5 * (1 + np.abs(want).max())
    assert np.abs(dq.cpu().numpy()[:nx] - gr["dc"].T).max() < 1e-5 * (1 + np.abs(gr["dc"]).max())
    idx, ptr, _ = pst
    pc = np.repeat(np.arange(nx), np.diff(ptr))
    wantP = gr["dP"][:, idx, pc] + np.where(idx != pc, gr["dP"][:, pc, idx], 0.0)        # one stored entry stands for (i,j) and (j,i)
    assert np.abs(dP.cpu().numpy() - wantP).max() < 1e-5 * (1 + np.abs(wantP).max())


@pytest.mark.gpu
def test_indefinite_P_fails_loudly_in_the_native_kernels():
    from cvxpylayers_amd.interfaces.mi355_if import _CvxpyLayer, SolverError
    n = 4
    cones = {"z": 1, "l": 0, "q": [], "s": []}
    tpl = P.dense_template(n, cones)
    pst = _upper_structure(n)
    ctx = MI355_ctx(pst, tpl.problem_data_index, cones, options={"eps": 1e-8})
    dev = torch.device("cuda", 0)
    Pm = np.stack([np.eye(n), np.diag([1.0, -1.0, 1.0, 1.0])])          # second instance indefinite
    A_eval, q_eval = tpl.values_from_dense(np.ones((2, 1, n)), np.ones((2, 1)), np.zeros((2, n)))
    args = (torch.from_numpy(_p_values(Pm, pst).T.copy()).to(dev), torch.from_numpy(q_eval).to(dev), torch.from_numpy(A_eval).to(dev))
    with pytest.raises(SolverError, match="Failed"):
        _CvxpyLayer.apply(*args, ctx, {}, False, None)
    primal, dual, info, _ = _CvxpyLayer.apply(*args, ctx, {"raise_on_error": False}, False, None)
    st = info["status"].cpu().numpy()
    assert st[0] == 1 and st[1] == -4 and bool(torch.isnan(primal[1]).all())
    np.testing.assert_allclose(primal[0].cpu().numpy(), np.full(n, 1.0 / n), atol=1e-6)       # min 1/2 |x|^2, sum x = 1


def test_full_symmetric_structure_is_paired_and_asymmetric_structure_has_no_native_route():
    n = 3
    cones = {"z": 1, "l": 0, "q": [], "s": []}
    tpl = P.dense_template(n, cones)
    full = (np.tile(np.arange(n), n).astype(np.int32), (np.arange(n + 1) * n).astype(np.int32), (n, n))          # dense CSC, both triangles
    qe = QuadEpigraph(full, (tpl.indices, tpl.indptr), (tpl.m, n + 1), dims_to_solver_dict(cones))
    assert not qe.one_triangle and qe.sym_perm is not None
    rows, cols = qe.p_rows, qe.p_cols
    assert all((rows[qe.sym_perm[k]], cols[qe.sym_perm[k]]) == (cols[k], rows[k]) for k in range(n * n))
    # values given on both triangles are averaged: P = [[2, 1, 0], [3, 2, 0], [0, 0, 2]] acts as its symmetric part
    Pm = np.array([[2.0, 1.0, 0.0], [3.0, 2.0, 0.0], [0.0, 0.0, 2.0]])
    vals = Pm[rows, cols][:, None]
    q_aug, A_aug = qe.assemble(torch.from_numpy(vals), torch.zeros(n + 1, 1, dtype=torch.float64), torch.zeros(tpl.nnz_aug, 1, dtype=torch.float64))
    Ls = A_aug[:, 0].numpy()[[k for k, s_ in enumerate(qe.src) if tpl.nnz_aug <= s_ < tpl.nnz_aug + n * (n + 1) // 2]]
    Lm = np.zeros((n, n)); Lm[qe.tri_j, qe.tri_k] = 0.0
    order = np.argsort([s_ for s_ in qe.src if tpl.nnz_aug <= s_ < tpl.nnz_aug + n * (n + 1) // 2])
    Lm[qe.tri_j, qe.tri_k] = Ls[order] / np.sqrt(2.0)
    np.testing.assert_allclose(Lm @ Lm.T, 0.5 * (Pm + Pm.T), atol=1e-9)
    lop = (np.array([0, 1, 1], dtype=np.int32), np.array([0, 2, 3, 3], dtype=np.int32), (n, n))    # (0,0), (1,0), (1,1): lower triangle
    assert QuadEpigraph(lop, (tpl.indices, tpl.indptr), (tpl.m, n + 1), dims_to_solver_dict(cones)).one_triangle
    asym = (np.array([1, 0], dtype=np.int32), np.array([0, 1, 1, 2], dtype=np.int32), (n, n))      # (1,0) and (0,2): both triangles, no mirrors
    qa = QuadEpigraph(asym, (tpl.indices, tpl.indptr), (tpl.m, n + 1), dims_to_solver_dict(cones))
    assert not qa.one_triangle and qa.sym_perm is None          # -> never the in-kernel route; the epigraph form symmetrises densely
