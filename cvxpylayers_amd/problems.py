"""Cone-level synthetic problem kit (no CVXPY needed).

Generates batches of cone programs  min c^T x  s.t.  A x + s = b, s in K  that are strictly
primal-dual feasible by construction:  b = A x0 + s0,  c = -A^T y0  with
s0 in int K, y0 in int K*.  Also builds the canonical *template* the plugin boundary consumes:
the CSC structure of CVXPY's augmented matrix [A_cvx | b_cvx] (m x (n+1)), where the solver
sees A = -A_cvx, b = b_cvx (reference: cvxpylayers/interfaces/diffcp_if.py:59-67,114-118).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def cone_rows(cones: dict) -> int:
    return (int(cones.get("z", 0)) + int(cones.get("l", 0)) + sum(int(d) for d in cones.get("q", []))
            + sum(int(k) * (int(k) + 1) // 2 for k in cones.get("s", [])) + 3 * int(cones.get("ep", 0)) + 3 * len(cones.get("p", [])))


@dataclass
class ConeTemplate:
    """What CVXPY's ParamConeProg hands a solver plugin once per layer (host side).

    indices/indptr: CSC structure of the augmented m x (n+1) matrix [A_cvx | b_cvx]
    (param_prob.reduced_A.problem_data_index, interfaces/__init__.py:26-33)."""
    n: int
    m: int
    indices: np.ndarray
    indptr: np.ndarray
    cones: dict = field(default_factory=dict)

    @property
    def nnz_aug(self) -> int:
        return int(self.indptr[-1])

    @property
    def nnzA(self) -> int:
        return int(self.indptr[-2])

    @property
    def b_idx(self) -> np.ndarray:
        return self.indices[self.indptr[-2]:self.indptr[-1]]

    @property
    def problem_data_index(self):
        return (self.indices, self.indptr, (self.m, self.n + 1))

    def dense_from_values(self, A_eval: np.ndarray, q_eval: np.ndarray):
        """(nnz_aug,B),(n+1,B) boundary values -> solver-form dense (A (B,m,n), b (B,m), c (B,n)).
        Follows _build_diffcp_matrices (diffcp_if.py:46-70): A = -A_aug[:, :-1], b = A_aug[:, -1], c = q[:-1]."""
        A_eval = np.asarray(A_eval, dtype=np.float64)
        q_eval = np.asarray(q_eval, dtype=np.float64)
        if A_eval.ndim == 1:
            A_eval = A_eval[:, None]
            q_eval = q_eval[:, None]
        B = A_eval.shape[1]
        aug = np.zeros((B, self.m, self.n + 1))
        cols = np.repeat(np.arange(self.n + 1), np.diff(self.indptr))
        aug[:, self.indices, cols] = A_eval.T
        return -aug[:, :, :self.n].copy(), aug[:, :, self.n].copy(), q_eval[:self.n].T.copy()

    def values_from_dense(self, A: np.ndarray, b: np.ndarray, c: np.ndarray):
        """inverse of dense_from_values: solver-form (A,b,c) -> boundary (A_eval (nnz_aug,B), q_eval (n+1,B))."""
        B = A.shape[0]
        aug = np.concatenate([-A, b[:, :, None]], axis=2)
        cols = np.repeat(np.arange(self.n + 1), np.diff(self.indptr))
        A_eval = aug[:, self.indices, cols].T.copy()
        q_eval = np.ascontiguousarray(np.concatenate([c.T, np.zeros((1, B))], axis=0))      # (n+1, B) row-major like the reference's q_map product (np.concatenate of a transposed view comes out column-major: autograd then copies every dq into that layout)
        return A_eval, q_eval


def dense_template(n: int, cones: dict, pattern: np.ndarray | None = None, b_pattern: np.ndarray | None = None) -> ConeTemplate:
    """Template with a given (m x n) boolean sparsity pattern for A (default: dense) and b (default: all rows)."""
    m = cone_rows(cones)
    if pattern is None:
        pattern = np.ones((m, n), dtype=bool)
    if b_pattern is None:
        b_pattern = np.ones(m, dtype=bool)
    full = np.concatenate([pattern, b_pattern[:, None]], axis=1)
    indices, indptr = [], [0]
    for j in range(n + 1):
        rows = np.nonzero(full[:, j])[0]
        indices.extend(rows.tolist())
        indptr.append(len(indices))
    return ConeTemplate(n=n, m=m, indices=np.asarray(indices, dtype=np.int32), indptr=np.asarray(indptr, dtype=np.int32), cones=dict(cones))


def _interior_point(rng, cones: dict, B: int):
    """s0 in int K, y0 in int K* blockwise."""
    z, l = int(cones.get("z", 0)), int(cones.get("l", 0))
    s_parts, y_parts = [], []
    if z:
        s_parts.append(np.zeros((B, z)))
        y_parts.append(rng.standard_normal((B, z)))
    if l:
        s_parts.append(np.abs(rng.standard_normal((B, l))) + 0.1)
        y_parts.append(np.abs(rng.standard_normal((B, l))) + 0.1)
    for d in cones.get("q", []):
        for parts in (s_parts, y_parts):
            u = rng.standard_normal((B, d - 1))
            t = np.linalg.norm(u, axis=1, keepdims=True) + 0.1 + np.abs(rng.standard_normal((B, 1)))
            parts.append(np.concatenate([t, u], axis=1))
    for k in cones.get("s", []):
        for parts in (s_parts, y_parts):
            G = rng.standard_normal((B, k, k))
            S = G @ np.swapaxes(G, 1, 2) / k + 0.1 * np.eye(k)
            parts.append(sym_to_svec(S))
    for _ in range(int(cones.get("ep", 0))):
        # s0 in int K_exp = {y e^(x/y) < z}; y0 in int K_exp^* = {u < 0, -u e^(v/u) < e w}
        yy = np.abs(rng.standard_normal((B, 1))) + 0.1
        xx = rng.standard_normal((B, 1)) * yy
        s_parts.append(np.concatenate([xx, yy, yy * np.exp(xx / yy) + 0.1 + np.abs(rng.standard_normal((B, 1)))], axis=1))
        u = -(np.abs(rng.standard_normal((B, 1))) + 0.1)
        v = rng.standard_normal((B, 1)) * np.abs(u)
        y_parts.append(np.concatenate([u, v, -u * np.exp(v / u - 1.0) + 0.1 + np.abs(rng.standard_normal((B, 1)))], axis=1))
    for al in cones.get("p", []):
        # K_a = {x^a y^(1-a) >= |z|}, K_a^* = {(u/a)^a (v/(1-a))^(1-a) >= |w|}; a negative entry is the dual cone (SCS convention)
        a = abs(float(al))
        def prim():
            xx = np.abs(rng.standard_normal((B, 1))) + 0.1; yy = np.abs(rng.standard_normal((B, 1))) + 0.1
            return np.concatenate([xx, yy, rng.uniform(-0.5, 0.5, (B, 1)) * xx ** a * yy ** (1 - a)], axis=1)
        def dual():
            t = prim(); t[:, 0] *= a; t[:, 1] *= (1 - a)
            return t
        s_parts.append(prim() if al > 0 else dual())
        y_parts.append(dual() if al > 0 else prim())
    return np.concatenate(s_parts, axis=1), np.concatenate(y_parts, axis=1)


def sym_to_svec(S: np.ndarray) -> np.ndarray:
    """(...,k,k) symmetric -> lower-triangular column-major svec with sqrt(2) off-diagonals."""
    k = S.shape[-1]
    out = []
    for j in range(k):
        for i in range(j, k):
            out.append(S[..., i, j] * (1.0 if i == j else np.sqrt(2.0)))
    return np.stack(out, axis=-1)


def svec_to_sym(v: np.ndarray, k: int) -> np.ndarray:
    S = np.zeros(v.shape[:-1] + (k, k))
    idx = 0
    for j in range(k):
        for i in range(j, k):
            val = v[..., idx] * (1.0 if i == j else 1.0 / np.sqrt(2.0))
            S[..., i, j] = val
            S[..., j, i] = val
            idx += 1
    return S


def generate(n: int, cones: dict, B: int, seed: int = 0, batched=("A", "b", "c"), pattern: np.ndarray | None = None):
    """Synthetic batch G(n, cones, B, seed, batched) -> (A (B,m,n), b (B,m), c (B,n)) float64, solver form."""
    rng = np.random.default_rng(seed)
    m = cone_rows(cones)
    A0 = rng.standard_normal((m, n)) / np.sqrt(n)
    if "A" in batched:
        A = A0[None] + 0.1 * rng.standard_normal((B, m, n)) / np.sqrt(n)
    else:
        A = np.broadcast_to(A0, (B, m, n)).copy()
    if pattern is not None:
        A = A * pattern[None]
    x0 = rng.standard_normal((B, n))
    s0, y0 = _interior_point(rng, cones, B)
    if "b" not in batched:
        x0[:] = x0[0]
        s0[:] = s0[0]
    if "c" not in batched:
        y0[:] = y0[0]
    b = np.einsum("bij,bj->bi", A, x0) + s0
    c = -np.einsum("bij,bi->bj", A, y0)
    return A, b, c


# The configurations BASELINE.json names, at cone level
CONFIGS = {
    "M": dict(n=50, cones={"z": 0, "l": 20, "q": [10] * 8}, B=4096),            # metric: n=50, m=100 SOC
    "C2": dict(n=50, cones={"z": 0, "l": 100, "q": []}, B=4096),                 # nonneg cone only (LP)
    "C2Q": dict(n=51, cones={"z": 0, "l": 100, "q": [52]}, B=4096),              # box QP n=50 in the SOC-epigraph form DIFFCP sees (box_qp_batch)
    "E": dict(n=40, cones={"z": 0, "l": 12, "q": [4], "s": [], "ep": 24}, B=4096),   # 24 exponential cones (logistic-regression layer shape)
    "E0": dict(n=40, cones={"z": 0, "l": 84, "q": [4], "s": []}, B=4096),             # same size, the exponential triples replaced by nonneg rows
    "C3": dict(n=100, cones={"z": 0, "l": 10, "q": [11] * 10}, B=4096),          # SOCP n=100, 10 SOC cones
    "C4": dict(n=210, cones={"z": 20, "l": 0, "q": [], "s": [20]}, B=1024),      # SDP one 20x20 PSD cone
}


def box_qp_batch(nx: int, B: int, seed: int = 0):
    """BASELINE config 2 as DIFFCP would see it: min ||F x - g||^2 s.t. lo <= x <= hi (F shared, g / lo / hi batched), written
    with the epigraph variable u and the rotated cone  ||(1 - u, 2 (F x - g))|| <= 1 + u.  Variables (x, u): n = nx + 1; rows:
    x - lo >= 0 (nx), hi - x >= 0 (nx), SOC(nx + 2): m = 3 nx + 2.  Returns solver-form (A (B,m,n), b (B,m), c (B,n), cones)."""
    rng = np.random.default_rng(seed)
    F = rng.standard_normal((nx, nx)) / np.sqrt(nx)
    g = rng.standard_normal((B, nx))
    lo = -0.5 - 0.5 * rng.random((B, nx)); hi = 0.5 + 0.5 * rng.random((B, nx))
    n = nx + 1; m = 3 * nx + 2
    A = np.zeros((m, n)); b = np.zeros((B, m))
    A[:nx, :nx] = -np.eye(nx); b[:, :nx] = -lo                    # s = x - lo
    A[nx:2 * nx, :nx] = np.eye(nx); b[:, nx:2 * nx] = hi          # s = hi - x
    r = 2 * nx
    A[r, nx] = -1.0; b[:, r] = 1.0                                # s0 = 1 + u
    A[r + 1, nx] = 1.0; b[:, r + 1] = 1.0                         # s1 = 1 - u
    A[r + 2:, :nx] = -2.0 * F; b[:, r + 2:] = -2.0 * g            # s_ = 2 (F x - g)
    c = np.zeros((B, n)); c[:, nx] = 1.0
    cones = {"z": 0, "l": 2 * nx, "q": [nx + 2]}
    return np.broadcast_to(A, (B, m, n)).copy(), b, c, cones


def native_box_qp_batch(nx: int, B: int, seed: int = 0):
    """BASELINE config 2 in its NATIVE form: min 1/2 x^T (2 F^T F) x - 2 g^T F x  s.t.  lo <= x <= hi -- the same F, g, lo, hi as
    box_qp_batch (same seed, same draws), P = 2 F^T F shared, 2 nx nonnegative rows.  Returns (A (2nx, nx) shared, b (B, 2nx), q (B, nx), P (nx, nx),
    p_structure = CSC structure of the upper triangle (indices, indptr, (nx, nx)), p_values (nnzP,))."""
    rng = np.random.default_rng(seed)
    F = rng.standard_normal((nx, nx)) / np.sqrt(nx); g = rng.standard_normal((B, nx))
    lo = -0.5 - 0.5 * rng.random((B, nx)); hi = 0.5 + 0.5 * rng.random((B, nx))
    A = np.concatenate([-np.eye(nx), np.eye(nx)], axis=0)
    rows, ptr = [], [0]
    for j in range(nx):
        rows.extend(range(j + 1)); ptr.append(len(rows))
    pst = (np.asarray(rows, dtype=np.int32), np.asarray(ptr, dtype=np.int32), (nx, nx))
    Pm = 2 * F.T @ F
    pv = Pm[pst[0], np.repeat(np.arange(nx), np.diff(pst[1]))]
    return A, np.concatenate([-lo, hi], axis=1), -2 * g @ F, Pm, pst, pv


def sdp_c4_batch(B: int, seed: int = 0, k: int = 20, neq: int = 20):
    """BASELINE config 4: SDP with one k x k PSD cone, x = svec(X) (n = k(k+1)/2 = 210), `neq` equality rows
    <A_j, X> = b_j and the PSD block s = svec(X) (A_psd = -I): m = neq + n = 230.  A is SHARED by the batch; b and c (= svec(C)) are
    per instance, strictly feasible by construction.  Returns (A (m,n), b (B,m), c (B,n), cones, template)."""
    rng = np.random.default_rng(seed)
    d = k * (k + 1) // 2
    n = d; cones = {"z": neq, "l": 0, "q": [], "s": [k]}; m = neq + d
    A = np.zeros((m, n)); A[:neq] = rng.standard_normal((neq, n)) / np.sqrt(n); A[neq:] = -np.eye(d)

    def pd():
        G = rng.standard_normal((B, k, k))
        return sym_to_svec(G @ np.swapaxes(G, 1, 2) / k + 0.1 * np.eye(k))
    x0 = pd()
    s0 = np.concatenate([np.zeros((B, neq)), x0], axis=1)
    y0 = np.concatenate([rng.standard_normal((B, neq)), pd()], axis=1)
    b = x0 @ A.T + s0
    c = -(y0 @ A)
    tpl = dense_template(n, cones, pattern=(A != 0), b_pattern=np.ones(m, bool))
    return A, b, c, cones, tpl


def portfolio_c5_batch(B: int, seed: int = 0, nw: int = 500, kf: int = 50, gamma: float = 1.0):
    """BASELINE config 5: min -mu^T w + gamma t  s.t.  1^T w = 1, w >= 0, ||F^T w|| <= t  (n = nw + 1 = 501,
    m = 1 + nw + kf + 1 = 552); A and b are SHARED, only mu (in c) is per instance.  Returns (A (m,n), b (m,), c (B,n), cones, template)."""
    rng = np.random.default_rng(seed)
    F = rng.standard_normal((nw, kf)) / np.sqrt(kf) * 0.3
    n = nw + 1; cones = {"z": 1, "l": nw, "q": [kf + 1]}; m = cone_rows(cones)
    A = np.zeros((m, n)); b = np.zeros(m)
    A[0, :nw] = 1.0; b[0] = 1.0                      # 1^T w = 1
    A[1:1 + nw, :nw] = -np.eye(nw)                   # w >= 0 : s = w
    A[1 + nw, nw] = -1.0                             # SOC: s0 = t
    A[2 + nw:, :nw] = -F.T                           # s_{1..k} = F^T w
    mu = 0.05 + 0.1 * rng.random((B, nw))
    c = np.concatenate([-mu, gamma * np.ones((B, 1))], axis=1)
    tpl = dense_template(n, cones, pattern=(A != 0), b_pattern=(b != 0))
    return A, b, c, cones, tpl
