"""Hand-canonicalised known-answer cone programs (no CVXPY), restating the closed-form cases the
reference's own tests assert (SURVEY.md section 8c list).  Each builder returns solver-form dense
(A (m,n), b (m,), c (n,), cones) plus whatever closed form the case has.

Rotated-cone trick used throughout:  ||r||^2 <= t   <=>   ||(1 - t, 2 r)|| <= 1 + t,
i.e. s = (1 + t, 1 - t, 2 r) in SOC(len(r) + 2).
"""
from __future__ import annotations

import numpy as np


def _soc_sumsq_rows(R, r0, tcol, nvar):
    """rows of  s = (1 + t, 1 - t, 2 (R x + r0)) = b - A x   for variable vector of length nvar (t at column tcol)."""
    k = R.shape[0]
    A = np.zeros((k + 2, nvar))
    b = np.zeros(k + 2)
    A[0, tcol] = -1.0; b[0] = 1.0          # s0 = 1 + t
    A[1, tcol] = 1.0; b[1] = 1.0           # s1 = 1 - t
    A[2:, :R.shape[1]] = -2.0 * R; b[2:] = 2.0 * r0
    return A, b


def ridge_ls(F, g):
    """min ||F x - g||^2 + ||x||^2  (reference tests/test_torch.py:90-118).  vars (x, t1, t2).
    closed form x = (F^T F + I)^{-1} F^T g."""
    mF, n = F.shape
    nv = n + 2
    A1, b1 = _soc_sumsq_rows(F, -g, n, nv)
    A2, b2 = _soc_sumsq_rows(np.eye(n), np.zeros(n), n + 1, nv)
    A = np.vstack([A1, A2]); b = np.concatenate([b1, b2])
    c = np.zeros(nv); c[n] = 1; c[n + 1] = 1
    cones = {"z": 0, "l": 0, "q": [mF + 2, n + 2]}
    x = np.linalg.solve(F.T @ F + np.eye(n), F.T @ g)
    return A, b, c, cones, x


def min_norm_eq(F, g):
    """min ||x||^2 s.t. F x = g  (tests/test_diffcp_optional_deps.py:30-58) -> x = F^T (F F^T)^{-1} g. vars (x, t)."""
    k, n = F.shape
    nv = n + 1
    Az = np.zeros((k, nv)); Az[:, :n] = F
    A2, b2 = _soc_sumsq_rows(np.eye(n), np.zeros(n), n, nv)
    A = np.vstack([Az, A2]); b = np.concatenate([g, b2])
    c = np.zeros(nv); c[n] = 1
    cones = {"z": k, "l": 0, "q": [n + 2]}
    x = F.T @ np.linalg.solve(F @ F.T, g)
    return A, b, c, cones, x


def box_qp(t):
    """min ||x - t||^2 s.t. 0 <= x <= 1 -> clip(t, 0, 1)  (tests/test_moreau.py:258-269). vars (x, u)."""
    n = len(t); nv = n + 1
    Al = np.zeros((2 * n, nv)); bl = np.zeros(2 * n)
    Al[:n, :n] = -np.eye(n)                   # x >= 0  : s = x
    Al[n:, :n] = np.eye(n); bl[n:] = 1.0      # x <= 1  : s = 1 - x
    A2, b2 = _soc_sumsq_rows(np.eye(n), -np.asarray(t, float), n, nv)
    A = np.vstack([Al, A2]); b = np.concatenate([bl, b2])
    c = np.zeros(nv); c[n] = 1
    return A, b, c, {"z": 0, "l": 2 * n, "q": [n + 2]}, np.clip(t, 0, 1)


def relu_proj(t):
    """min ||x - t||^2 s.t. x >= 0 -> max(t, 0), d x/d t = 1[t>0]  (tests/test_mlx.py:669-695)."""
    n = len(t); nv = n + 1
    Al = np.zeros((n, nv)); Al[:, :n] = -np.eye(n)
    A2, b2 = _soc_sumsq_rows(np.eye(n), -np.asarray(t, float), n, nv)
    A = np.vstack([Al, A2]); b = np.concatenate([np.zeros(n), b2])
    c = np.zeros(nv); c[n] = 1
    return A, b, c, {"z": 0, "l": n, "q": [n + 2]}, np.maximum(t, 0)


def simplex_lp(cvec, total=1.0):
    """min c^T x s.t. sum x = total, x >= 0 (tests/test_dual_variables.py:14-42): vertex at argmin c."""
    n = len(cvec)
    A = np.vstack([np.ones((1, n)), -np.eye(n)]); b = np.concatenate([[total], np.zeros(n)])
    x = np.zeros(n); x[int(np.argmin(cvec))] = total
    return A, b, np.asarray(cvec, float), {"z": 1, "l": n, "q": []}, x


def soc_lin(cvec, t):
    """min c^T x s.t. ||x|| <= t -> x = -t c/||c||  (tests/test_dual_variables.py:316-343).  s = (t, x)."""
    n = len(cvec)
    A = np.vstack([np.zeros((1, n)), -np.eye(n)]); b = np.concatenate([[t], np.zeros(n)])
    cvec = np.asarray(cvec, float)
    return A, b, cvec, {"z": 0, "l": 0, "q": [n + 1]}, -t * cvec / np.linalg.norm(cvec)


def svec_index(k):
    idx = {}
    p = 0
    for j in range(k):
        for i in range(j, k):
            idx[(i, j)] = p; idx[(j, i)] = p; p += 1
    return idx


def sdp_min_eig(Cm):
    """min tr(C X) s.t. tr X = 1, X PSD -> X = v v^T (min eigvec), dual of PSD constraint = C - lmin I
    (tests/test_dual_variables.py:523-550).  x = svec(X) scaled: variable is svec (sqrt2 off-diag)."""
    k = Cm.shape[0]; d = k * (k + 1) // 2
    idx = svec_index(k)
    c = np.zeros(d); tr = np.zeros(d)
    for j in range(k):
        for i in range(j, k):
            p = idx[(i, j)]
            c[p] = Cm[i, j] * (1.0 if i == j else np.sqrt(2.0))
            if i == j:
                tr[p] = 1.0
    A = np.vstack([tr[None, :], -np.eye(d)]); b = np.concatenate([[1.0], np.zeros(d)])
    w, V = np.linalg.eigh(Cm)
    X = np.outer(V[:, 0], V[:, 0])
    return A, b, c, {"z": 1, "l": 0, "q": [], "s": [k]}, X, Cm - w[0] * np.eye(k)


def infeasible():
    """x >= 1 and x <= -1  (tests/test_torch.py:299-316)."""
    A = np.array([[-1.0], [1.0]]); b = np.array([-1.0, -1.0]); c = np.array([0.0])
    return A, b, c, {"z": 0, "l": 2, "q": []}


def unbounded(p=1.0):
    """min x s.t. x <= p."""
    A = np.array([[1.0]]); b = np.array([p]); c = np.array([1.0])
    return A, b, c, {"z": 0, "l": 1, "q": []}


def entropy_max(k):
    """max sum_i -x_i log x_i  s.t. sum x = 1  ->  x = 1/k, value log k.   v = (x[k], t[k]); min -sum t;
    (t_i, x_i, 1) in K_exp  <=>  x_i exp(t_i / x_i) <= 1  <=>  t_i <= -x_i log x_i."""
    n, m = 2 * k, 1 + 3 * k
    A = np.zeros((m, n)); b = np.zeros(m); c = np.zeros(n); c[k:] = -1.0
    A[0, :k] = 1.0; b[0] = 1.0
    for i in range(k):
        r = 1 + 3 * i
        A[r, k + i] = -1.0; A[r + 1, i] = -1.0; b[r + 2] = 1.0
    return A, b, c, dict(z=1, l=0, q=[], s=[], ep=k), np.full(k, 1.0 / k)


def logistic_regression(X, lab, lam):
    """min sum_i log(1 + exp(-lab_i x_i^T w)) + lam * r,  ||w|| <= r   (cone form of tests/test_torch.py's logistic layer).
    v = (w[d], t[N], a[N], b[N], r);  log(1+e^u) <= t  <=>  e^(u-t) + e^(-t) <= 1  <=>  (u-t,1,a), (-t,1,b) in K_exp, a + b <= 1."""
    N, d = X.shape
    n = d + 3 * N + 1
    m = N + (d + 1) + 6 * N
    A = np.zeros((m, n)); b = np.zeros(m); c = np.zeros(n); c[d:d + N] = 1.0; c[-1] = lam
    for i in range(N):
        A[i, d + N + i] = 1.0; A[i, d + 2 * N + i] = 1.0; b[i] = 1.0           # 1 - a_i - b_i >= 0
    r0 = N
    A[r0, -1] = -1.0                                                          # SOC (r, w)
    for j in range(d):
        A[r0 + 1 + j, j] = -1.0
    e0 = r0 + d + 1
    for i in range(N):
        r = e0 + 6 * i
        A[r, :d] = lab[i] * X[i]; A[r, d + i] = 1.0                           # s_x = -lab x^T w - t
        b[r + 1] = 1.0
        A[r + 2, d + N + i] = -1.0                                            # s_z = a_i
        A[r + 3, d + i] = 1.0                                                 # s_x = -t
        b[r + 4] = 1.0
        A[r + 5, d + 2 * N + i] = -1.0                                        # s_z = b_i
    return A, b, c, dict(z=0, l=N, q=[d + 1], s=[], ep=2 * N)


def geo_mean_max(w, total, alpha=0.5):
    """max x1^a x2^(1-a)  s.t.  w1 x1 + w2 x2 = total  ->  x1 = a total / w1, x2 = (1-a) total / w2.   v = (x1, x2, t), (x1, x2, t) in K_a."""
    A = np.zeros((4, 3)); b = np.zeros(4); c = np.array([0.0, 0.0, -1.0])
    A[0, :2] = w; b[0] = total
    A[1, 0] = A[2, 1] = A[3, 2] = -1.0
    xs = np.array([alpha * total / w[0], (1 - alpha) * total / w[1]])
    return A, b, c, dict(z=1, l=0, q=[], s=[], ep=0, p=[alpha]), xs


# The shared-A adjoint stops like diffcp's LSQR by default (atol = btol = 1e-8, 2 N iterations: mi355_if.lsqr_rule).  Tests that compare its gradients with a
# DIRECT elimination to 1e-5 ask for a tight solve explicitly -- the same triple as solver_args lsqr_atol / lsqr_btol / lsqr_iter_lim.
TIGHT_LSQR = (1e-12, 1e-12, 20000)


TIGHTER_LSQR = (1e-14, 1e-14, 40000)
# The rules under which the ORACLE's own LSQR answer is re-computed to measure LSQR's accuracy on an instance: tighter atol / btol, and -- for the ill-conditioned
# instances, which stop on the condition-estimate test long before atol / btol matter -- the same tolerances with conlim relaxed (1e8 -> 3e8, 1e9)
LSQR_OWN_RULES = (dict(lsqr_atol=1e-14, lsqr_btol=1e-14, lsqr_iter_lim=40000), dict(lsqr_atol=1e-12, lsqr_btol=1e-12, lsqr_iter_lim=20000, lsqr_conlim=3e8),
                  dict(lsqr_atol=1e-12, lsqr_btol=1e-12, lsqr_iter_lim=20000, lsqr_conlim=1e9))


def lsqr_own_movement(run, base, dist):
    """max over LSQR_OWN_RULES of dist(base, run(**rule)): how far the oracle's own answer moves, per instance, when the stopping test that fired is pushed out"""
    import numpy as np
    own = None
    for rule in LSQR_OWN_RULES:
        d = dist(base, run(**rule))
        own = d if own is None else np.maximum(own, d)
    return own


def assert_lsqr_agreement_per_instance(el, own_move, strict=1e-5, factor=3.0, floor=0.7):
    """The per-instance rule for comparing two LSQR implementations on the same adjoint system (round 6; it replaces "95 % of the instances below 5e-3"):
    EVERY instance must be within `strict` of the oracle's LSQR answer, or no further from it than `factor` x the distance the ORACLE'S OWN answer moves when its
    stopping rule is pushed out (lsqr_own_movement) -- i.e. the difference must be explained, instance by instance, by LSQR's own accuracy on a near-singular
    system (the components along near-null directions converge on neither side; such instances stop on LSQR's condition-estimate test, conlim = 1e8, and the
    answer at that point depends on the summation order to about the amount it still moves over the following iterations).  At least `floor` of the instances
    must meet `strict` outright."""
    import numpy as np
    el, own_move = np.asarray(el), np.asarray(own_move)
    ok = (el <= strict) | (el <= factor * own_move)
    assert ok.all(), [(int(i), float(el[i]), float(own_move[i])) for i in np.nonzero(~ok)[0]]
    assert (el <= strict).mean() >= floor, (float((el <= strict).mean()), el)
