"""The torch frontend (cvxpylayers_amd.torch.CvxpyLayer), restating the reference's API / shape / error contract tests
(tests/test_torch.py:251-352, :626-702; tests/test_parse_args.py:25-101) and closed-form value + gradient tests
(tests/test_torch.py:90-118 ridge LS, :355-384 broadcasting; tests/test_moreau.py:258-269 box QP; README.md:84-101)
on hand-canonicalised templates (CVXPY is not installable here)."""
import numpy as np
import pytest
import torch

import kit
from cvxpylayers_amd.torch import CvxpyLayer, VariableRecovery
from cvxpylayers_amd.torch.templates import template_from_affine_builder

torch.set_default_dtype(torch.double)


def ridge_template(mF, n):
    """min ||F x - g||^2 + ||x||^2 with parameters F (mF, n), g (mF,); variable x."""
    def builder(F, g):
        A, b, c, cones, _ = kit.ridge_ls(F, g)
        return A, b, c
    cones = {"z": 0, "l": 0, "q": [mF + 2, n + 2]}
    return template_from_affine_builder(builder, [(mF, n), (mF,)], cones, [VariableRecovery(slice(0, n), None, (n,))])


def boxqp_template(n):
    def builder(t):
        A, b, c, cones, _ = kit.box_qp(t)
        return A, b, c
    return template_from_affine_builder(builder, [(n,)], {"z": 0, "l": 2 * n, "q": [n + 2]}, [VariableRecovery(slice(0, n), None, (n,))])


# ------------------------------------------------------------------ CPU: template construction + API contract
def test_affine_probing_reproduces_the_builder():
    rng = np.random.default_rng(0)
    tpl = ridge_template(5, 3)
    F = rng.standard_normal((5, 3)); g = rng.standard_normal(5)
    p = np.concatenate([F.flatten(order="F"), g, [1.0]])
    A, b, c, cones, _ = kit.ridge_ls(F, g)
    indices, indptr, (m, np1) = tpl.A_structure
    cols = np.repeat(np.arange(np1), np.diff(indptr))
    aug = np.concatenate([-A, b[:, None]], axis=1)
    np.testing.assert_allclose(tpl.A_map @ p, aug[indices, cols], atol=1e-14)
    np.testing.assert_allclose((tpl.q_map @ p)[:-1], c, atol=1e-14)
    assert len(indices) == np.count_nonzero(aug)        # structural zeros are not part of the pattern


def test_param_count_shape_and_batch_errors():
    layer = CvxpyLayer(template=ridge_template(5, 3))
    with pytest.raises(ValueError, match="A tensor must be provided for each CVXPY parameter"):
        layer(torch.zeros(5, 3))
    with pytest.raises(ValueError, match="Invalid parameter shape for parameter 1"):
        layer(torch.zeros(5, 3), torch.zeros(4))
    with pytest.raises(ValueError, match="Invalid parameter dimensionality for parameter 0"):
        layer(torch.zeros(2, 2, 5, 3), torch.zeros(5))
    with pytest.raises(ValueError, match="Inconsistent batch sizes"):
        layer(torch.zeros(2, 5, 3), torch.zeros(3, 5))
    assert layer.validate_params([torch.zeros(5, 3), torch.zeros(5)]) == ()
    assert layer.validate_params([torch.zeros(7, 5, 3), torch.zeros(5)]) == (7,)


def test_unknown_solver_key():
    with pytest.raises(RuntimeError, match="Unknown solver"):
        CvxpyLayer(template=ridge_template(3, 2), solver="NOPE")


def test_cpu_tensors_are_refused_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    layer = CvxpyLayer(template=boxqp_template(3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(torch.tensor([2.0, 0.5, -1.0]))


# ------------------------------------------------------------------ GPU: values and gradients through the whole layer
@pytest.mark.gpu
def test_ridge_ls_gradient_matches_closed_form():
    # reference tests/test_torch.py:90-118 (m=100, n=20, seed 243): gradients of sum(x*) wrt F, g vs autograd of the closed form
    torch.manual_seed(243)
    mF, n = 100, 20
    layer = CvxpyLayer(template=ridge_template(mF, n), solver_args={"eps": 1e-10, "max_iters": 50000})
    F = torch.randn(mF, n, device="cuda", requires_grad=True)
    g = torch.randn(mF, device="cuda", requires_grad=True)
    (x,) = layer(F, g)
    assert x.shape == (n,)
    x.sum().backward()
    F2 = F.detach().clone().requires_grad_(); g2 = g.detach().clone().requires_grad_()
    xc = torch.linalg.solve(F2.t() @ F2 + torch.eye(n, device="cuda"), F2.t() @ g2)
    xc.sum().backward()
    assert torch.allclose(x, xc, atol=1e-6)
    assert torch.allclose(F.grad, F2.grad, atol=1e-6) and torch.allclose(g.grad, g2.grad, atol=1e-6)


@pytest.mark.gpu
def test_broadcast_unbatched_parameter_sums_gradient_over_batch():
    # reference tests/test_torch.py:355-384: batched g, unbatched (broadcast) F; d/dF is the SUM over the batch
    torch.manual_seed(0)
    mF, n, B = 12, 4, 5
    layer = CvxpyLayer(template=ridge_template(mF, n), solver_args={"eps": 1e-10, "max_iters": 50000})
    F = torch.randn(mF, n, device="cuda", requires_grad=True)
    g = torch.randn(B, mF, device="cuda", requires_grad=True)
    (x,) = layer(F, g)
    assert x.shape == (B, n)
    x.sum().backward()
    F2 = F.detach().clone().requires_grad_(); g2 = g.detach().clone().requires_grad_()
    xc = torch.linalg.solve(F2.t() @ F2 + torch.eye(n, device="cuda"), F2.t() @ g2.t()).t()
    xc.sum().backward()
    assert torch.allclose(x, xc, atol=1e-6)
    assert torch.allclose(F.grad, F2.grad, atol=1e-5) and torch.allclose(g.grad, g2.grad, atol=1e-6)


@pytest.mark.gpu
def test_shared_parameter_feeds_two_layers():
    # reference tests/test_torch.py:387-413 (test_shared_parameter): ONE parameter tensor A feeds two least-squares layers with different constants b1, b2; the
    # gradient of cat(x1, x2) with respect to A is the sum of both layers' contributions (there: gradcheck; here: autograd of the closed form x = (A^T A)^-1 A^T b)
    rng = np.random.default_rng(243)
    m, n = 10, 5
    b1, b2 = rng.standard_normal(m), rng.standard_normal(m)

    def ls_template(bc):
        def builder(Ap):
            Ac = np.zeros((m + 2, n + 1)); bcone = np.zeros(m + 2)
            Ac[0, n] = -1.0; bcone[0] = 1.0                 # s0 = 1 + t
            Ac[1, n] = 1.0; bcone[1] = 1.0                  # s1 = 1 - t
            Ac[2:, :n] = -2.0 * Ap; bcone[2:] = -2.0 * bc   # s_ = 2 (A x - b)
            c = np.zeros(n + 1); c[n] = 1.0
            return Ac, bcone, c
        return template_from_affine_builder(builder, [(m, n)], {"z": 0, "l": 0, "q": [m + 2]}, [VariableRecovery(slice(0, n), None, (n,))])
    args = {"eps": 1e-10, "acceleration_lookback": 0, "max_iters": 10000}          # the reference test's solver_args
    layer1 = CvxpyLayer(template=ls_template(b1), solver_args=args)
    layer2 = CvxpyLayer(template=ls_template(b2), solver_args=args)
    torch.manual_seed(243)
    A = torch.randn(m, n, dtype=torch.float64, device="cuda", requires_grad=True)
    wts = torch.linspace(0.5, 1.5, 2 * n, dtype=torch.float64, device="cuda")
    (x1,) = layer1(A); (x2,) = layer2(A)
    (torch.cat((x1, x2)) * wts).sum().backward()
    A2 = A.detach().clone().requires_grad_()
    bt = torch.tensor(np.stack([b1, b2], axis=1), dtype=torch.float64, device="cuda")
    xc = torch.linalg.solve(A2.t() @ A2, A2.t() @ bt)                               # (n, 2)
    (torch.cat((xc[:, 0], xc[:, 1])) * wts).sum().backward()
    assert torch.allclose(torch.cat((x1, x2)), torch.cat((xc[:, 0], xc[:, 1])).detach(), atol=1e-6)
    assert torch.allclose(A.grad, A2.grad, atol=1e-5), (A.grad - A2.grad).abs().max()


@pytest.mark.gpu
def test_box_qp_clip_and_batch_of_one_keeps_its_axis():
    # tests/test_moreau.py:258-269 (clip) and tests/test_torch.py:668-702 (batch of 1 != unbatched)
    layer = CvxpyLayer(template=boxqp_template(3), solver_args={"eps": 1e-9})
    t = torch.tensor([2.0, 0.5, -1.0], device="cuda", requires_grad=True)
    (x,) = layer(t)
    assert x.shape == (3,) and torch.allclose(x, torch.tensor([1.0, 0.5, 0.0], device="cuda"), atol=1e-5)
    x.sum().backward()
    assert t.grad.shape == (3,) and torch.allclose(t.grad, torch.tensor([0.0, 1.0, 0.0], device="cuda"), atol=1e-4)
    tb = t.detach().clone()[None].requires_grad_()
    (xb,) = layer(tb)
    assert xb.shape == (1, 3)
    xb.sum().backward()
    assert tb.grad.shape == (1, 3)


@pytest.mark.gpu
def test_no_grad_and_solver_args_reach_the_engine():
    # tests/test_torch.py:626-665 (no grad) and :705-752 (solver_args reach the solver: max_iters=1 vs many)
    layer = CvxpyLayer(template=boxqp_template(3))
    t = torch.tensor([2.0, 0.5, -1.0], device="cuda")
    with torch.no_grad():
        (x1,) = layer(t, solver_args={"max_iters": 1, "raise_on_error": False})
    (x2,) = layer(t, solver_args={"max_iters": 10000, "eps": 1e-9})
    assert not x2.requires_grad
    assert (x1 - x2).abs().max() > 1e-3
    assert int(layer.info["iters"][0]) > 1


@pytest.mark.gpu
def test_infeasible_raises_solver_error():
    from cvxpylayers_amd.interfaces.mi355_if import SolverError

    def builder(p):          # x >= 1 and x <= -p  (infeasible for p = 1), tests/test_torch.py:299-316
        return np.array([[-1.0], [1.0]]), np.array([-1.0, -float(p[0])]), np.array([0.0])
    tpl = template_from_affine_builder(builder, [(1,)], {"z": 0, "l": 2, "q": []}, [VariableRecovery(slice(0, 1), None, (1,))])
    layer = CvxpyLayer(template=tpl)
    with pytest.raises(SolverError):
        layer(torch.tensor([1.0], device="cuda"))


@pytest.mark.gpu
def test_sdp_min_eigenvector_value_dual_and_gradient():
    # min tr(C X) s.t. tr X = 1, X PSD  ->  X = v v^T (min eigenvector), PSD dual = C - lmin I
    # (reference tests/test_dual_variables.py:523-550 values; tests/test_torch.py:233-248 3x3 SDP gradient)
    k = 3
    d = k * (k + 1) // 2

    def builder(Cm):
        A, b, c, cones, _, _ = kit.sdp_min_eig(0.5 * (Cm + Cm.T))
        return A, b, c
    tpl = template_from_affine_builder(builder, [(k, k)], {"z": 1, "l": 0, "q": [], "s": [k]},
                                       [VariableRecovery(slice(0, d), None, (k, k), source="primal", unpack_fn="svec_dual"),
                                        VariableRecovery(None, slice(1, 1 + d), (k, k), source="dual", unpack_fn="svec_dual")])
    layer = CvxpyLayer(template=tpl, solver_args={"eps": 1e-10, "max_iters": 200000})
    torch.manual_seed(3)
    G = torch.randn(k, k)
    C0 = (G + G.t()) / 2 + torch.diag(torch.tensor([0.0, 1.0, 2.5]))
    C = C0.cuda().requires_grad_()
    X, Z = layer(C)
    w, V = torch.linalg.eigh(C0)
    v = V[:, 0]
    assert torch.allclose(X.cpu(), torch.outer(v, v), atol=1e-5)
    assert torch.allclose(Z.cpu(), C0 - w[0] * torch.eye(k), atol=1e-5)
    Wt = torch.randn(k, k)
    Wt = (Wt + Wt.t()) / 2
    (X * Wt.cuda()).sum().backward()
    C2 = C0.clone().requires_grad_()
    w2, V2 = torch.linalg.eigh((C2 + C2.t()) / 2)
    (torch.outer(V2[:, 0], V2[:, 0]) * Wt).sum().backward()
    assert torch.allclose(C.grad.cpu(), C2.grad, atol=1e-5), (C.grad.cpu(), C2.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [12, 40])
def test_logistic_regression_layer_through_the_exponential_cone(N):
    """The logistic-regression layer of reference tests/test_torch.py:158-230: data Z = lab * X is the parameter, the fitted
    weights come back; the gradient of sum(w*) wrt the data is checked against autograd through an unrolled Newton solve.
    N = 12: the register / LDS-resident kernels; N = 40 (80 exponential cones, per-instance A well beyond their sizes): the size-generic kernels."""
    rng = np.random.default_rng(1)
    d, lam = 3, 0.5
    X = rng.standard_normal((N, d)); lab = np.sign(X @ np.array([1.0, -2.0, 0.5]) + 0.3 * rng.standard_normal(N))

    def builder(Z):
        A, b, c, cones = kit.logistic_regression(Z, np.ones(N), lam)
        return A, b, c
    cones = kit.logistic_regression(X, lab, lam)[3]
    tpl = template_from_affine_builder(builder, [(N, d)], cones, [VariableRecovery(slice(0, d), None, (d,))])
    layer = CvxpyLayer(template=tpl, solver_args={"eps": 1e-10, "max_iters": 200000})
    Z = torch.tensor(lab[:, None] * X, device="cuda", requires_grad=True)
    (w,) = layer(Z)
    w.sum().backward()

    Z2 = Z.detach().clone().requires_grad_()
    f = lambda wv: torch.nn.functional.softplus(-(Z2 @ wv)).sum() + lam * wv.norm()
    wv = w.detach().clone()
    for _ in range(6):       # Newton from the layer's answer: converged after one step, differentiable through the last ones
        wv_ = wv if wv.requires_grad else wv.requires_grad_()
        g = torch.autograd.grad(f(wv_), wv_, create_graph=True)[0]
        H = torch.stack([torch.autograd.grad(g[i], wv_, create_graph=True)[0] for i in range(d)])
        wv = wv_ - torch.linalg.solve(H, g)
    assert torch.allclose(w, wv.detach(), atol=1e-6)
    wv.sum().backward()
    assert torch.allclose(Z.grad, Z2.grad, atol=1e-5), (Z.grad - Z2.grad).abs().max()


@pytest.mark.gpu
def test_geometric_program_layer_log_parameters_exp_variables():
    """gp=True layers (reference torch/cvxpylayer.py:411-420, 274-282; tests/test_torch.py:429-623): parameters enter through their
    logs, primal variables come back through exp.   min x + y  s.t.  x y >= a   ->   x = y = sqrt(a).
    Log space (u, v) = log(x, y), alpha = log a:  min t  s.t.  log(e^u + e^v) <= t,  u + v >= alpha;
    e^(u-t) + e^(v-t) <= 1  <=>  (u - t, 1, a1), (v - t, 1, a2) in K_exp, a1 + a2 <= 1.   Variables (u, v, t, a1, a2)."""
    def builder(alpha):
        A = np.zeros((8, 5)); b = np.zeros(8); c = np.zeros(5); c[2] = 1.0
        A[0, 3] = A[0, 4] = 1.0; b[0] = 1.0                 # 1 - a1 - a2 >= 0
        A[1, 0] = A[1, 1] = -1.0; b[1] = -alpha[0]          # u + v - alpha >= 0
        A[2, 0] = -1.0; A[2, 2] = 1.0; b[3] = 1.0; A[4, 3] = -1.0
        A[5, 1] = -1.0; A[5, 2] = 1.0; b[6] = 1.0; A[7, 4] = -1.0
        return A, b, c
    tpl = template_from_affine_builder(builder, [(1,)], {"z": 0, "l": 2, "q": [], "s": [], "ep": 2},
                                       [VariableRecovery(slice(0, 1), None, (1,)), VariableRecovery(slice(1, 2), None, (1,))])
    tpl.gp = True; tpl.gp_log_mask = (True,)
    layer = CvxpyLayer(template=tpl, solver_args={"eps": 1e-10, "max_iters": 100000})
    a = torch.tensor([[0.5], [2.0], [9.0]], device="cuda", dtype=torch.float64, requires_grad=True)
    x, y = layer(a)
    assert x.shape == (3, 1) and y.shape == (3, 1)
    assert torch.allclose(x, a.detach().sqrt(), atol=1e-6) and torch.allclose(y, a.detach().sqrt(), atol=1e-6)
    (x.sum() + 2 * y.sum()).backward()
    assert torch.allclose(a.grad, 1.5 / a.detach().sqrt(), atol=1e-5), a.grad


@pytest.mark.gpu
def test_lml_layer_entropy_terms():
    """Limited multi-label projection (reference tests/test_torch.py:219-230): min -x.y - sum entr(y) - sum entr(1-y), sum y = k,
    whose solution is y = sigmoid(x + nu) with nu fixed by sum y = k.  entr(y) >= t  <=>  (t, y, 1) in K_exp.
    Variables (y[d], t[d], r[d]); the gradient is checked against autograd through the scalar Newton solve for nu."""
    d, k = 4, 2

    def builder(x):
        n, m = 3 * d, 1 + 6 * d
        A = np.zeros((m, n)); b = np.zeros(m); c = np.zeros(n)
        c[:d] = -x; c[d:] = -1.0
        A[0, :d] = 1.0; b[0] = k
        for i in range(d):
            r = 1 + 6 * i
            A[r, d + i] = -1.0; A[r + 1, i] = -1.0; b[r + 2] = 1.0                    # (t_i, y_i, 1)
            A[r + 3, 2 * d + i] = -1.0; A[r + 4, i] = 1.0; b[r + 4] = 1.0; b[r + 5] = 1.0   # (r_i, 1 - y_i, 1)
        return A, b, c
    tpl = template_from_affine_builder(builder, [(d,)], {"z": 1, "l": 0, "q": [], "s": [], "ep": 2 * d}, [VariableRecovery(slice(0, d), None, (d,))])
    layer = CvxpyLayer(template=tpl, solver_args={"eps": 1e-10, "max_iters": 200000})
    x = torch.tensor([1.0, -1.0, -1.0, -1.0], device="cuda", dtype=torch.float64, requires_grad=True)
    (y,) = layer(x)
    wts = torch.tensor([1.0, 2.0, -1.0, 0.5], device="cuda", dtype=torch.float64)
    (y * wts).sum().backward()

    x2 = x.detach().clone().requires_grad_()
    nu = torch.zeros((), device="cuda", dtype=torch.float64)
    for _ in range(60):
        sg = torch.sigmoid(x2 + nu)
        nu = nu - (sg.sum() - k) / (sg * (1 - sg)).sum()
    y2 = torch.sigmoid(x2 + nu)
    assert torch.allclose(y, y2.detach(), atol=1e-6), (y, y2)
    (y2 * wts).sum().backward()
    assert torch.allclose(x.grad, x2.grad, atol=1e-5), (x.grad, x2.grad)


@pytest.mark.gpu
def test_matrix_variable_fortran_recovery_and_no_grad_inputs():
    """Matrix variable recovered column-major (reference tests/test_torch.py:755-780) and inputs that do not require grad give an
    output that does not either (:647-665).   min ||A X - B||_F^2 + ||X||_F^2  ->  X = (A^T A + I)^-1 A^T B."""
    m, n, k = 10, 4, 3

    def builder(A_, B_):
        F = np.kron(np.eye(k), A_)                          # vec_F(A X) = (I (x) A) vec_F(X)
        A, b, c, cones, _ = kit.ridge_ls(F, B_.reshape(-1, order="F"))
        return A, b, c
    cones = kit.ridge_ls(np.zeros((m * k, n * k)), np.zeros(m * k))[3]
    tpl = template_from_affine_builder(builder, [(m, n), (m, k)], cones, [VariableRecovery(slice(0, n * k), None, (n, k))])
    layer = CvxpyLayer(template=tpl, solver_args={"eps": 1e-10, "max_iters": 100000})
    torch.manual_seed(123)
    A_t = torch.randn(m, n, dtype=torch.float64, device="cuda"); B_t = torch.randn(m, k, dtype=torch.float64, device="cuda")
    (X,) = layer(A_t, B_t)
    assert X.shape == (n, k) and not X.requires_grad
    Xc = torch.linalg.solve(A_t.t() @ A_t + torch.eye(n, dtype=torch.float64, device="cuda"), A_t.t() @ B_t)
    assert torch.allclose(X, Xc, atol=1e-6)
    # batched, float32 inputs: float64 comes back (diffcp_if.py:374-375), batch axis first
    A_b = torch.randn(5, m, n, device="cuda"); B_b = torch.randn(5, m, k, device="cuda")
    (Xb,) = layer(A_b, B_b)
    assert Xb.shape == (5, n, k) and Xb.dtype == torch.float64
    Xbc = torch.linalg.solve(A_b.double().transpose(1, 2) @ A_b.double() + torch.eye(n, dtype=torch.float64, device="cuda"), A_b.double().transpose(1, 2) @ B_b.double())
    assert torch.allclose(Xb, Xbc, atol=1e-6)


@pytest.mark.gpu
def test_warm_start_reuses_the_previous_solution():
    """warm_start=True starts from the layer's previous solution (reference: the MOREAU cache, torch/cvxpylayer.py:464-487;
    tests/test_moreau.py:1363-1620): same answer, far fewer iterations after a small parameter change, cold start when the
    batch size changes."""
    n = 20
    layer = CvxpyLayer(template=boxqp_template(n), solver_args={"eps": 1e-6, "max_iters": 100000})
    torch.manual_seed(0)
    t = torch.randn(64, n, device="cuda", dtype=torch.float64) * 2
    (x0,) = layer(t)
    cold = layer.info["iters"].float().mean().item()
    t2 = t + 1e-3 * torch.randn_like(t)
    (xc,) = layer(t2)
    cold2 = layer.info["iters"].float().mean().item()
    (x0,) = layer(t)                                     # refresh the cache with the solution at t
    (xw,) = layer(t2, warm_start=True)
    warm = layer.info["iters"].float().mean().item()
    assert torch.allclose(xw, torch.clamp(t2, 0, 1), atol=1e-4) and torch.allclose(xw, xc, atol=1e-4)
    assert warm < 0.5 * cold2, (cold, cold2, warm)
    (xs,) = layer(t2[:7], warm_start=True)               # different batch size: cache ignored
    assert torch.allclose(xs, torch.clamp(t2[:7], 0, 1), atol=1e-4)
    t3 = t2.clone().requires_grad_()
    (xg,) = layer(t3, warm_start=True)                   # gradients still flow from a warm-started solve
    xg.sum().backward()
    inside = ((t2 > 1e-3) & (t2 < 1 - 1e-3)).double()
    outside = ((t2 < -1e-3) | (t2 > 1 + 1e-3)).double()
    assert ((t3.grad - 1).abs() * inside).max() < 1e-3 and (t3.grad.abs() * outside).max() < 1e-3


@pytest.mark.gpu
def test_float32_parameters_give_float64_results():
    """float32 parameters give float64 results (diffcp_if.py:374-375).  (An empty batch is not a concept of the reference's frontend:
    batch size 0 is its marker for "unbatched", utils/parse_args.py:94-143; the engine itself returns empty outputs for B = 0.)"""
    n = 6
    layer = CvxpyLayer(template=boxqp_template(n), solver_args={"eps": 1e-8})
    t = torch.linspace(-1, 2, 3 * n, device="cuda").reshape(3, n)           # float32
    (x,) = layer(t)
    assert x.dtype == torch.float64 and torch.allclose(x, torch.clamp(t.double(), 0, 1), atol=1e-5)
    from cvxpylayers_amd.interfaces.mi355_if import make_settings
    eng = layer.ctx.solver_ctx.engine(torch.device("cuda", 0))
    out = eng.solve(torch.empty((0, eng.nnz_aug), dtype=torch.float64, device="cuda"), torch.empty((eng.n + 1, 0), dtype=torch.float64, device="cuda"), make_settings({}))
    assert out[0].shape == (0, eng.n) and out[4].shape == (0,)


def test_affine_probing_with_a_quadratic_objective_builds_the_P_map():
    """builder returning (A, b, c, P): the upper triangle of P becomes the template's P structure and P_map @ (p, 1) its values"""
    n = 4

    def builder(Pp, qp):
        A = np.zeros((1, n)); A[0] = 1.0
        return A, np.ones(1), qp, 0.5 * (Pp + Pp.T) + np.eye(n)
    tpl = template_from_affine_builder(builder, [(n, n), (n,)], {"z": 1, "l": 0, "q": [], "s": []}, [VariableRecovery(slice(0, n), None, (n,))])
    idx, ptr, shape = tpl.P_structure
    assert shape == (n, n) and len(idx) == n * (n + 1) // 2 and tpl.P_map.shape == (len(idx), n * n + n + 1)
    rng = np.random.default_rng(0)
    Pp, qp = rng.standard_normal((n, n)), rng.standard_normal(n)
    pvec = np.concatenate([Pp.reshape(-1, order="F"), qp, [1.0]])
    vals = tpl.P_map @ pvec
    want = (0.5 * (Pp + Pp.T) + np.eye(n))[idx, np.repeat(np.arange(n), np.diff(ptr))]
    np.testing.assert_allclose(vals, want, atol=1e-14)


@pytest.mark.gpu
def test_failure_masking_keeps_the_good_instances_and_zeroes_the_bad_gradients():
    """solver_args={"raise_on_error": False} (per-instance failure masking): an infeasible instance inside a batch comes back as NaN
    with zero parameter gradient, the other instances and their gradients are untouched, info["status"] says which; the default still raises."""
    from cvxpylayers_amd.interfaces.mi355_if import SolverError

    def builder(p):          # min x  s.t.  x >= p0,  x <= p1   (infeasible when p0 > p1);  x* = p0, dx/dp0 = 1
        return np.array([[-1.0], [1.0]]), np.array([-float(p[0]), float(p[1])]), np.array([1.0])
    tpl = template_from_affine_builder(builder, [(2,)], {"z": 0, "l": 2, "q": []}, [VariableRecovery(slice(0, 1), None, (1,))])
    layer = CvxpyLayer(template=tpl, solver_args={"eps": 1e-9})
    p = torch.tensor([[0.5, 2.0], [1.0, -1.0], [-0.3, 0.7]], device="cuda", requires_grad=True)       # instance 1 is infeasible
    with pytest.raises(SolverError):
        layer(p)
    x, = layer(p, solver_args={"raise_on_error": False})
    st = layer.info["status"].cpu().numpy()
    assert st[0] == 1 and st[2] == 1 and st[1] < 0
    xv = x.detach().cpu().numpy()[:, 0]
    assert np.isnan(xv[1]) and abs(xv[0] - 0.5) < 1e-6 and abs(xv[2] + 0.3) < 1e-6
    torch.nan_to_num(x, nan=0.0).sum().backward()             # (a NaN-aware loss, as a training loop that masks failed samples would write it)
    g = p.grad.cpu().numpy()
    assert np.isfinite(g).all() and np.abs(g[1]).max() == 0.0
    np.testing.assert_allclose(g[0], [1.0, 0.0], atol=1e-5); np.testing.assert_allclose(g[2], [1.0, 0.0], atol=1e-5)
