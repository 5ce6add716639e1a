"""The layers of the reference's example notebooks whose STORED OUTPUTS are numbers produced by the real diffcp / SCS stack
(/root/reference/examples/torch/*.ipynb), hand-canonicalised (no CVXPY here) as CanonTemplates.

Nothing in this file reads /root/reference: tests/golden/make_notebook_golden.py parses the notebooks' stored inputs / outputs into
tests/golden/ref_notebook_*.npz; tests/test_notebook_golden.py replays them through the oracle (CPU) and the HIP engine (GPU).
A canonicalisation by hand need not equal CVXPY's row / column order -- optimal variables and parameter gradients do not depend on it.
Cone row order: zero, nonneg, SOC, PSD, exponential (SCS order; oracle/cone_oracle.c:28)."""
from __future__ import annotations

import numpy as np

import kit
from cvxpylayers_amd.torch import VariableRecovery
from cvxpylayers_amd.torch.templates import template_from_affine_builder


def ot_template(n=3, m=3):
    """optimal_transport.ipynb cell 3:  min tr(P^T C) - eps (sum entr(P) + sum P)  s.t.  P 1 = a, P^T 1 = b, P >= 0;
    parameters [C (n,m), a (n,), b (m,), eps (1,)], variable P (n,m).
    v = (vec_F(P), t);  t_ij <= entr(P_ij)  <=>  (t_ij, P_ij, 1) in K_exp;  c = (vec_F(C) - eps, -eps)."""
    N = n * m
    pid = lambda i, j: i + n * j                       # Fortran position of P_ij

    def builder(C, a, b, eps):
        e = float(np.asarray(eps).reshape(-1)[0])
        nv = 2 * N
        rows = n + m + N + 3 * N
        A = np.zeros((rows, nv)); rhs = np.zeros(rows); c = np.zeros(nv)
        for i in range(n):
            for j in range(m):
                A[i, pid(i, j)] = 1.0; A[n + j, pid(i, j)] = 1.0
                c[pid(i, j)] = C[i, j] - e
        rhs[:n] = a; rhs[n:n + m] = b
        r0 = n + m
        for p in range(N):
            A[r0 + p, p] = -1.0                        # s = P_ij >= 0
        r0 += N
        for p in range(N):
            A[r0 + 3 * p, N + p] = -1.0                # x = t
            A[r0 + 3 * p + 1, p] = -1.0                # y = P
            rhs[r0 + 3 * p + 2] = 1.0                  # z = 1
            c[N + p] = -e
        return A, rhs, c
    cones = dict(z=n + m, l=N, q=[], s=[], ep=N)
    return template_from_affine_builder(builder, [(n, m), (n,), (m,), (1,)], cones, [VariableRecovery(slice(0, N), None, (n, m))])


def lqr_sdp_template(A, B, Q0, W):
    """lqr.ipynb cell 3:  max tr(P W)  s.t.  [[R0 + B'PB, B'PA], [A'PB, Q0 + A'PA - P]] >> 0,  P >> 0;  parameter R0 (m,m),
    variable P (n,n) symmetric (canonical variable = upper triangle, row-major, unscaled: "svec_primal", torch/cvxpylayer.py:183-198).
    Both PSD blocks are  s = svec(block)  (lower triangle, column-major, sqrt(2) off-diagonals)."""
    n, m = B.shape
    d = n * (n + 1) // 2
    iu = np.triu_indices(n)
    k1 = n + m
    d1 = k1 * (k1 + 1) // 2

    def svec(M):
        out = []
        for j in range(M.shape[0]):
            for i in range(j, M.shape[0]):
                out.append(M[i, j] * (1.0 if i == j else np.sqrt(2.0)))
        return np.array(out)

    def big(P, R0, const):
        return np.block([[const * R0 + B.T @ P @ B, B.T @ P @ A], [A.T @ P @ B, const * Q0 + A.T @ P @ A - P]])

    def builder(R0):
        Rs = 0.5 * (R0 + R0.T)
        Am = np.zeros((d1 + d, d)); rhs = np.zeros(d1 + d); c = np.zeros(d)
        rhs[:d1] = svec(big(np.zeros((n, n)), Rs, 1.0))
        for p, (i, j) in enumerate(zip(*iu)):
            E = np.zeros((n, n)); E[i, j] = E[j, i] = 1.0
            Am[:d1, p] = -svec(big(E, np.zeros((m, m)), 0.0))
            Am[d1:, p] = -svec(E)
            c[p] = -np.sum(E * W)                     # maximise tr(P W)
        return Am, rhs, c
    cones = dict(z=0, l=0, q=[], s=[k1, n])
    return template_from_affine_builder(builder, [(m, m)], cones, [VariableRecovery(slice(0, d), None, (n, n), unpack_fn="svec_primal")])


def fit_lr_template(mtrain, nfeat=1):
    """tutorial.ipynb cell 16-17:  min (1/m) ||X a + b - Y||^2 + lam ||a||_1 + alpha ||a||^2;  parameters [X (m,n), Y (m,), lam (), alpha ()],
    variables [a (n,), b ()].   v = (a, b, t1, u, t2):  ||X a + b 1 - Y||^2 <= t1, |a| <= u, ||a||^2 <= t2; c = (0, 0, 1/m, lam 1, alpha)."""
    n = nfeat
    nv = 2 * n + 3

    def builder(X, Y, lam, alpha):
        R = np.concatenate([X, np.ones((mtrain, 1))], axis=1)
        A1, b1 = kit._soc_sumsq_rows(R, -Y, n + 1, nv)
        A2, b2 = kit._soc_sumsq_rows(np.eye(n), np.zeros(n), 2 * n + 2, nv)
        Al = np.zeros((2 * n, nv))
        for i in range(n):
            Al[i, i] = 1.0; Al[i, n + 2 + i] = -1.0            # u - a >= 0
            Al[n + i, i] = -1.0; Al[n + i, n + 2 + i] = -1.0   # u + a >= 0
        Am = np.vstack([Al, A1, A2]); rhs = np.concatenate([np.zeros(2 * n), b1, b2])
        c = np.zeros(nv); c[n + 1] = 1.0 / mtrain; c[n + 2:2 * n + 2] = float(lam); c[2 * n + 2] = float(alpha)
        return Am, rhs, c
    cones = dict(z=0, l=2 * n, q=[mtrain + 2, n + 2])
    return template_from_affine_builder(builder, [(mtrain, n), (mtrain,), (), ()], cones,
                                        [VariableRecovery(slice(0, n), None, (n,)), VariableRecovery(slice(n, n + 1), None, ())])


def denoise_template(n=100):
    """signal_denoising.ipynb cell 14:  min ||x - y||^2 + lam ||diff(y)||^2;  parameters [x (n,), lam ()], variable y (n,).
    v = (y, t1, t2):  ||y - x||^2 <= t1,  ||D y||^2 <= t2 (D: first differences),  c = (0, 1, lam).  A does not depend on the parameters: the batch
    of 500 signals the notebook evaluates runs on the shared-A kernels."""
    nv = n + 2
    D = np.diff(np.eye(n), axis=0)

    def builder(x, lam):
        A1, b1 = kit._soc_sumsq_rows(np.eye(n), -np.asarray(x), n, nv)
        A2, b2 = kit._soc_sumsq_rows(D, np.zeros(n - 1), n + 1, nv)
        c = np.zeros(nv); c[n] = 1.0; c[n + 1] = float(np.asarray(lam).reshape(-1)[0])
        return np.vstack([A1, A2]), np.concatenate([b1, b2]), c
    cones = dict(z=0, l=0, q=[n + 2, n + 1])
    return template_from_affine_builder(builder, [(n,), ()], cones, [VariableRecovery(slice(0, n), None, (n,))])


def resource_allocation_template(m=10):
    """resource_allocation.ipynb cell 2:  max sum t  s.t.  sum y = B, y >= 0, -exp(-alpha_i u_i) >= alpha_i t_i, u = y * inverse_p;
    parameters [B (), inverse_p (m,), alpha (m,)], variable y (m,).   v = (y, u, t);  (-alpha_i u_i, 1, -alpha_i t_i) in K_exp."""
    def builder(Bt, invp, alpha):
        nv = 3 * m
        rows = 1 + m + m + 3 * m
        A = np.zeros((rows, nv)); rhs = np.zeros(rows); c = np.zeros(nv); c[2 * m:] = -1.0
        A[0, :m] = 1.0; rhs[0] = float(Bt)
        for i in range(m):
            A[1 + i, m + i] = 1.0; A[1 + i, i] = -invp[i]      # u_i - invp_i y_i = 0
            A[1 + m + i, i] = -1.0                             # y_i >= 0
            r = 1 + 2 * m + 3 * i
            A[r, m + i] = alpha[i]                             # x = -alpha u
            rhs[r + 1] = 1.0                                   # y = 1
            A[r + 2, 2 * m + i] = alpha[i]                     # z = -alpha t
        return A, rhs, c
    cones = dict(z=1 + m, l=m, q=[], s=[], ep=m)
    return template_from_affine_builder(builder, [(), (m,), (m,)], cones, [VariableRecovery(slice(0, m), None, (m,))])


def adp_policy_template(n=2, m=3):
    """convex_approximate_dynamic_programming.ipynb cell 2:  min 1/2 ||P_sqrt u||^2 + x^T y + q^T u  s.t.  ||u|| <= 1, y = P_21 u;
    parameters [x (n,1), P_sqrt (m,m), P_21 (n,m), q (m,1)], variable u (m,1) (y is the DPP device that keeps x^T P_21 u affine in each parameter).
    v = (u, y, t):  y - P_21 u = 0 (zero cone),  (1, u) in SOC(m + 1),  ||P_sqrt u||^2 <= t,  c = (q, x, 1/2)."""
    nv = m + n + 1

    def builder(x, Ps, P21, q):
        Az = np.zeros((n, nv)); Az[:, :m] = -np.asarray(P21); Az[:, m:m + n] = np.eye(n)
        Aq = np.zeros((m + 1, nv)); bq = np.zeros(m + 1); bq[0] = 1.0
        Aq[1:, :m] = -np.eye(m)                            # s = (1, u)
        As, bs = kit._soc_sumsq_rows(np.asarray(Ps), np.zeros(m), m + n, nv)
        c = np.zeros(nv); c[:m] = np.asarray(q).reshape(m); c[m:m + n] = np.asarray(x).reshape(n); c[m + n] = 0.5
        return np.vstack([Az, Aq, As]), np.concatenate([np.zeros(n), bq, bs]), c
    cones = dict(z=n, l=0, q=[m + 1, m + 2])
    return template_from_affine_builder(builder, [(n, 1), (m, m), (n, m), (m, 1)], cones, [VariableRecovery(slice(0, m), None, (m, 1))])


def adp_problem():
    """The data of convex_approximate_dynamic_programming.ipynb cell 2 (np.random.seed(1)) and the LQR initialisation of cell 3."""
    from scipy.linalg import solve_discrete_are, sqrtm
    np.random.seed(1)
    n, m = 2, 3
    A = np.eye(n) + 1e-2 * np.random.randn(n, n)
    B = 1e-2 / 3 * np.random.randn(n, m)
    P_lqr = solve_discrete_are(A, B, np.eye(n), np.eye(m))
    return dict(n=n, m=m, A=A, B=B, P_sqrt0=np.real(sqrtm(np.eye(m) + B.T @ P_lqr @ B)), P_21_0=A.T @ P_lqr @ B)


def adp_train(policy, iters, device="cpu"):
    """cell 3's train(): SGD with momentum on (P_sqrt, P_21, q) through N = 8 closed-loop roll-outs of T = 25 policy solves each (200 chained
    forward solves and their adjoints per step), same RNG call sequence.  Returns the losses it would print."""
    import torch
    d = adp_problem(); n, m = d["n"], d["m"]
    P_sqrt = torch.tensor(d["P_sqrt0"], device=device).requires_grad_(True)
    P_21 = torch.tensor(d["P_21_0"], device=device).requires_grad_(True)
    q = torch.zeros((m, 1), dtype=torch.double, device=device, requires_grad=True)
    A_t, B_t = torch.tensor(d["A"], device=device), torch.tensor(d["B"], device=device)

    def evaluate(T):
        x = torch.zeros(n, 1, dtype=torch.double, device=device); cost = 0.0
        for _ in range(T):
            u, = policy(x, P_sqrt, P_21, q)
            cost = cost + (x.t() @ x + u.t() @ u).squeeze() / T                        # Q = I, R = I
            x = A_t @ x + B_t @ u + (.2 * torch.randn(n, 1).double()).to(device)      # drawn in float32 on the CPU like the notebook
        return cost
    opt = torch.optim.SGD([P_sqrt, P_21, q], lr=.02, momentum=.9)
    out = []
    for _ in range(iters):
        torch.manual_seed(1)                                                           # "use same seeds each iteration"
        opt.zero_grad()
        loss = sum(evaluate(25) for _ in range(8)) / 8
        loss.backward(); opt.step()
        out.append(loss.item())
    return out


def constrained_lqr_problem():
    """constrained_lqr.ipynb cell 2 (np.random.seed(1)): n = 8 states, m = 2 inputs, |u|_inf <= 0.1, noise std 0.5."""
    from scipy.linalg import solve_discrete_are
    np.random.seed(1)
    n, m = 8, 2
    A = np.random.randn(n, n)
    A /= np.max(np.abs(np.linalg.eig(A)[0]))
    B = np.random.randn(n, m)
    return dict(n=n, m=m, A=A, B=B, noise=np.sqrt(.25), u_max=.1, P_are=solve_discrete_are(A, B, np.eye(n), np.eye(m)))


def constrained_lqr_template(A, B, u_max):
    """constrained_lqr.ipynb cell 7:  min u^T R0 u + ||P_sqrt xnext||^2 + q^T xnext  s.t.  xnext = A x + B u, |u|_inf <= u_max  (R0 = I);
    parameters [x (n,1), P_sqrt (n,n), q (n,)], variable u (m,1).  v = (u, xnext, t1, t2): xnext - B u = A x (the parameter x enters b),
    +-u <= u_max, ||u||^2 <= t1, ||P_sqrt xnext||^2 <= t2 (the parameter P_sqrt enters the cone rows), c = (0, q, 1, 1)."""
    n, m = B.shape
    nv = m + n + 2

    def builder(x, Ps, q):
        Az = np.zeros((n, nv)); Az[:, :m] = -B; Az[:, m:m + n] = np.eye(n)
        Al = np.zeros((2 * m, nv))
        for i in range(m):
            Al[i, i] = 1.0; Al[m + i, i] = -1.0
        A1, b1 = kit._soc_sumsq_rows(np.eye(m), np.zeros(m), m + n, nv)
        Pf = np.zeros((n, m + n)); Pf[:, m:] = np.asarray(Ps)
        A2, b2 = kit._soc_sumsq_rows(Pf, np.zeros(n), m + n + 1, nv)
        c = np.zeros(nv); c[m:m + n] = np.asarray(q).reshape(n); c[m + n] = 1.0; c[m + n + 1] = 1.0
        return np.vstack([Az, Al, A1, A2]), np.concatenate([A @ np.asarray(x).reshape(n), np.full(2 * m, u_max), b1, b2]), c
    return template_from_affine_builder(builder, [(n, 1), (n, n), (n,)], dict(z=n, l=2 * m, q=[m + 2, n + 2]), [VariableRecovery(slice(0, m), None, (m, 1))])


def constrained_lqr_loss(policy, d, device="cpu"):
    """cell 12's closed-loop cost: batch of `batch_size` trajectories, `time_horizon` policy solves in sequence, same RNG call sequence (the notebook runs in double)."""
    import torch
    n = d["n"]
    At, Bt = torch.tensor(d["A"], device=device), torch.tensor(d["B"], device=device)
    noise = float(d["noise"])

    def loss(time_horizon, batch_size, P_sqrt, q, seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        x = (noise * torch.randn(batch_size, n, 1, dtype=torch.float64)).to(device)
        Pb = P_sqrt.repeat(batch_size, 1, 1); qb = q.repeat(batch_size, 1)
        total = 0.0
        for _ in range(time_horizon):
            u, = policy(x, Pb, qb)
            total = total + ((x * x).sum() + (u * u).sum()) / (time_horizon * batch_size)          # Q0 = I, R0 = I
            x = At @ x + Bt @ u + (noise * torch.randn(batch_size, n, 1, dtype=torch.float64)).to(device)
        return total
    return loss


def monotone_template(m=10):
    """monotonic_output_regression.ipynb cell 3:  min ||y - yhat||_2  s.t.  diff(y) >= 0;  parameter yhat (m,), variable y (m,).
    v = (y, t):  y[i+1] - y[i] >= 0,  (t, y - yhat) in SOC(m + 1),  c = (0, 1).  Its solution is the isotonic regression of yhat."""
    nv = m + 1

    def builder(yhat):
        Al = np.zeros((m - 1, nv))
        for i in range(m - 1):
            Al[i, i] = 1.0; Al[i, i + 1] = -1.0
        Aq = np.zeros((m + 1, nv)); bq = np.zeros(m + 1)
        Aq[0, m] = -1.0; Aq[1:, :m] = -np.eye(m); bq[1:] = -np.asarray(yhat)
        c = np.zeros(nv); c[m] = 1.0
        return np.vstack([Al, Aq]), np.concatenate([np.zeros(m - 1), bq]), c
    return template_from_affine_builder(builder, [(m,)], dict(z=0, l=m - 1, q=[m + 1]), [VariableRecovery(slice(0, m), None, (m,))])


# ---------------------------------------------------------------------------------------------- independent high-precision answers
def sinkhorn(C, a, b, eps, iters=20000):
    """Entropic OT by Sinkhorn's fixed point in torch (differentiable): P = diag(u) exp(-C/eps) diag(v).  Stationarity of the
    notebook's objective: C_ij + eps log P_ij = f_i + g_j."""
    import torch
    K = torch.exp(-C / eps)
    u = torch.ones_like(a); v = torch.ones_like(b)
    for _ in range(iters):
        u = a / (K @ v)
        v = b / (K.t() @ u)
    return u[:, None] * K * v[None, :]


def resource_allocation_exact(Bt, invp, alpha):
    """KKT of the (smooth, separable) resource problem:  invp_i exp(-alpha_i invp_i y_i) = nu  or  y_i = 0; bisection on nu."""
    Bt = np.atleast_1d(np.asarray(Bt, float)); out = np.zeros((len(Bt), len(alpha)))
    for k in range(len(Bt)):
        ip = np.asarray(invp[k], float)
        y_of = lambda nu: np.maximum(np.log(ip / nu) / (alpha * ip), 0.0)
        lo, hi = 1e-300, ip.max()
        for _ in range(300):
            mid = np.sqrt(lo * hi) if lo > 0 else 0.5 * (lo + hi)
            if y_of(mid).sum() > Bt[k]:
                lo = mid
            else:
                hi = mid
        out[k] = y_of(np.sqrt(lo * hi))
    return out


# ---------------------------------------------------------------------------------------------- supply_chain.ipynb
SC_A_IN = np.array([[1, 0, 0, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 1, 0, 0], [0, 0, 0, 1, 1, 0, 0, 0]], float)
SC_A_OUT = np.array([[0, 0, 1, 1, 0, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 1, 0, 1]], float)
SC = dict(n=4, k=2, c=2, m=8, h_max=3.0, u_max=2.0, alpha=0.01, beta=0.01, tau=0.05, r=1.3, retail_links=[6, 7])


def supply_chain_template():
    """supply_chain.ipynb cell 6:  min [p; tau; -r]^T u + ||P_sqrt h_next||^2 + q^T h_next  s.t.  h_next = h + (A_in - A_out) u,
    h_next <= h_max, 0 <= u <= u_max, A_out u <= h, u[retail] <= d;  parameters [x (8,1) = (h, p, d), P_sqrt (4,4), q (4,1)], variable u (8,1).
    v = (u, h_next, t),  ||P_sqrt h_next||^2 <= t  as the rotated cone of tests/kit.py."""
    n, k, c, m = SC["n"], SC["k"], SC["c"], SC["m"]
    nv = m + n + 1
    D = SC_A_IN - SC_A_OUT

    def builder(x, Ps, q):
        h, p, d = x[:n, 0], x[n:n + k, 0], x[n + k:, 0]
        Az = np.zeros((n, nv)); Az[:, :m] = -D; Az[:, m:m + n] = np.eye(n)
        rows = [Az]; rhs = [h]
        Ah = np.zeros((n, nv)); Ah[:, m:m + n] = np.eye(n); rows.append(Ah); rhs.append(np.full(n, SC["h_max"]))
        Al = np.zeros((m, nv)); Al[:, :m] = -np.eye(m); rows.append(Al); rhs.append(np.zeros(m))
        Au = np.zeros((m, nv)); Au[:, :m] = np.eye(m); rows.append(Au); rhs.append(np.full(m, SC["u_max"]))
        Ao = np.zeros((n, nv)); Ao[:, :m] = SC_A_OUT; rows.append(Ao); rhs.append(h)
        Ar = np.zeros((c, nv))
        for i, l in enumerate(SC["retail_links"]):
            Ar[i, l] = 1.0
        rows.append(Ar); rhs.append(d)
        R = np.zeros((n, nv)); R[:, m:m + n] = Ps
        A2, b2 = kit._soc_sumsq_rows(R[:, :m + n], np.zeros(n), nv - 1, nv)
        rows.append(A2); rhs.append(b2)
        cvec = np.zeros(nv)
        cvec[:k] = p; cvec[k:m - c] = SC["tau"]; cvec[m - c:m] = -SC["r"]
        cvec[m:m + n] = q[:, 0]; cvec[nv - 1] = 1.0
        return np.vstack(rows), np.concatenate(rhs), cvec
    cones = dict(z=n, l=n + m + m + n + c, q=[n + 2])
    return template_from_affine_builder(builder, [(n + k + c, 1), (n, n), (n, 1)], cones, [VariableRecovery(slice(0, m), None, (m, 1))])


def supply_chain_sim(policy, device="cpu"):
    """The closed-loop simulation of supply_chain.ipynb cells 2, 5, 7 (same torch RNG call sequence), around `policy(x, P_sqrt, q) -> u`.
    Returns loss(params, time_horizon, batch_size, seed) -> mean stage cost (differentiable w.r.t. params)."""
    import torch
    n, k, c, m = SC["n"], SC["k"], SC["c"], SC["m"]
    mu = torch.cat([torch.tensor([0, 0.1]).double(), torch.tensor([0.0, 0.4]).double()], 0)
    sig = torch.cat([torch.tensor([0.2, 0.2]).double(), torch.tensor([0.2, 0.2]).double()], 0)
    w_dist = torch.distributions.log_normal.LogNormal(mu, sig)                         # sampled on the CPU like the notebook
    A_d = torch.zeros(n + k + c, n + k + c, dtype=torch.double, device=device); A_d[:n, :n] = torch.eye(n, dtype=torch.double)
    B_d = torch.zeros(n + k + c, m, dtype=torch.double, device=device); B_d[:n] = torch.tensor(SC_A_IN - SC_A_OUT, device=device)
    tau = torch.full((m - k - c, 1), SC["tau"], dtype=torch.double, device=device); r = torch.full((k, 1), SC["r"], dtype=torch.double, device=device)

    def stage_cost(x, u):
        bs = x.shape[0]
        h, p = x[:, :n], x[:, n:n + k]
        s_vec = torch.cat([p, tau.repeat(bs, 1, 1), -r.repeat(bs, 1, 1)], 1)
        S = torch.bmm(s_vec.transpose(1, 2), u)
        H = SC["alpha"] * h + SC["beta"] * (h ** 2)
        return torch.sum(S, 1) + torch.sum(H, 1)

    def simulate(x, u):
        bs = x.shape[0]
        w = w_dist.sample((bs,)).double().view((bs, k + c, 1)).to(device)
        w_batch = torch.cat([torch.zeros(bs, n, 1, dtype=torch.double, device=device), w], 1)
        return torch.bmm(A_d.repeat(bs, 1, 1), x) + torch.bmm(B_d.repeat(bs, 1, 1), u) + w_batch

    def loss(params, time_horizon, batch_size=1, seed=None):
        P_sqrt, q = params
        if seed is not None:
            torch.manual_seed(seed)
        x_b_0 = SC["h_max"] * torch.rand(batch_size, n, 1, dtype=torch.float32).double()      # the notebook draws in float32 (torch default)
        w_0 = w_dist.sample((batch_size,)).double().view((batch_size, k + c, 1))
        x_t = torch.cat([x_b_0, w_0], 1).double().to(device)
        Pb = P_sqrt.repeat(batch_size, 1, 1); qb = q.repeat(batch_size, 1, 1)
        cost = 0.0
        for _ in range(time_horizon):
            u_t = policy(x_t, Pb, qb)
            x_t = simulate(x_t, u_t)
            cost = cost + stage_cost(x_t, u_t).mean() / time_horizon
        return cost
    return loss
