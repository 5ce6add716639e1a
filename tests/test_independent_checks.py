"""Solver-independent evidence for the oracle (and, on the GPU, for the engine): neither SCS nor diffcp can be imported here, so besides the
closed-form answers of test_oracle_known_answers.py the optimum itself is pinned two more ways that do not share code with the oracle:
  * random LPs of the BASELINE config 2 shape against scipy's HiGHS (an unrelated simplex / interior-point code);
  * a conic optimality CERTIFICATE for every BASELINE shape: primal feasibility A x + s = b, s in K; dual feasibility A^T y + c = 0, y in K*;
    complementarity s.y = 0 -- cone memberships evaluated here with numpy eigenvalues / norms, not with the oracle's projections.
    A point with a small certificate residual is optimal whatever produced it."""
import numpy as np
import pytest
from scipy.optimize import linprog

from cvxpylayers_amd import problems as P
from oracle import oracle


def cone_violation(v, cones, dual=False):
    """max distance-like violation of v in K (or K*) per instance; self-dual cones except the zero cone (K* = free)."""
    v = np.atleast_2d(v); B = v.shape[0]
    out = np.zeros(B); off = 0
    z = int(cones.get("z", 0))
    if not dual: out = np.maximum(out, np.abs(v[:, :z]).max(axis=1) if z else 0.0)
    off += z
    l = int(cones.get("l", 0))
    if l: out = np.maximum(out, np.maximum(-v[:, off:off + l], 0).max(axis=1))
    off += l
    for d in cones.get("q", []):
        blk = v[:, off:off + d]
        out = np.maximum(out, np.maximum(np.linalg.norm(blk[:, 1:], axis=1) - blk[:, 0], 0)); off += d
    for k in cones.get("s", []):
        d = k * (k + 1) // 2
        S = P.svec_to_sym(v[:, off:off + d], k) if hasattr(P, "svec_to_sym") else None
        if S is None:
            S = np.zeros((B, k, k)); idx = 0
            for j in range(k):
                for i in range(j, k):
                    val = v[:, off + idx] * (1.0 if i == j else 1 / np.sqrt(2)); S[:, i, j] = val; S[:, j, i] = val; idx += 1
        out = np.maximum(out, np.maximum(-np.linalg.eigvalsh(S)[:, 0], 0)); off += d
    assert off == v.shape[1] - 3 * (int(cones.get("ep", 0)) + len(cones.get("p", []))), "exp / power cones are checked in test_oracle_known_answers.py"
    return out


def certificate(A, b, c, cones, x, y, s):
    """relative residuals (primal, dual, cone memberships, complementarity) per instance"""
    pri = np.abs(np.einsum("bij,bj->bi", A, x) + s - b).max(axis=1) / (1 + np.abs(b).max(axis=1))
    dua = np.abs(np.einsum("bij,bi->bj", A, y) + c).max(axis=1) / (1 + np.abs(c).max(axis=1))
    gap = np.abs((s * y).sum(axis=1)) / (1 + np.abs((c * x).sum(axis=1)))
    return np.maximum.reduce([pri, dua, gap, cone_violation(s, cones) / (1 + np.abs(s).max(axis=1)), cone_violation(y, cones, dual=True) / (1 + np.abs(y).max(axis=1))])


def _instances(name, B, seed=0):
    if name == "C4":
        A, b, c, cones, _ = P.sdp_c4_batch(B, seed=seed, k=8, neq=6)          # (the 20 x 20 shape at size is the GPU test's job)
        return np.broadcast_to(A, (B,) + A.shape).copy(), b, c, cones
    if name == "C5":
        A, b, c, cones, _ = P.portfolio_c5_batch(B, seed=seed, nw=60, kf=8)
        return np.broadcast_to(A, (B,) + A.shape).copy(), np.broadcast_to(b, (B,) + b.shape).copy(), c, cones
    cfg = P.CONFIGS[name]
    A, b, c = P.generate(cfg["n"], cfg["cones"], B, seed=seed)
    return A, b, c, cfg["cones"]


@pytest.mark.parametrize("name", ["M", "C3", "C4", "C5"])
def test_oracle_solutions_carry_an_optimality_certificate(name):
    A, b, c, cones = _instances(name, 6)
    r = oracle.solve_batch(A, b, c, cones, eps=1e-10, max_iters=400000)
    assert (r["status"] == 1).all()
    res = certificate(A, b, c, cones, r["x"], r["y"], r["s"])
    assert res.max() < 5e-8, res


def test_random_lps_match_highs():
    """BASELINE config 2 shape (n = 50, 100 inequality rows): min c^T x s.t. A x <= b."""
    cfg = P.CONFIGS["C2"]
    A, b, c = P.generate(cfg["n"], cfg["cones"], 6, seed=0)
    r = oracle.solve_batch(A, b, c, cfg["cones"], eps=1e-10, max_iters=400000)
    assert (r["status"] == 1).all()
    for k in range(A.shape[0]):
        ref = linprog(c[k], A_ub=A[k], b_ub=b[k], bounds=[(None, None)] * cfg["n"], method="highs")
        assert ref.status == 0
        assert abs(ref.fun - c[k] @ r["x"][k]) < 1e-7 * (1 + abs(ref.fun)), (ref.fun, c[k] @ r["x"][k])
        # duals of the inequality rows: HiGHS reports marginals <= 0 for A_ub x <= b_ub; the cone form has y >= 0 with A^T y + c = 0
        np.testing.assert_allclose(-ref.ineqlin.marginals, r["y"][k], atol=1e-6 * (1 + np.abs(r["y"][k]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["M", "C3"])
def test_engine_solutions_carry_an_optimality_certificate(name):
    import torch
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
    A, b, c, cones = _instances(name, 64)
    n = A.shape[2]
    tpl = P.dense_template(n, cones)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    dev = torch.device("cuda", 0)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
    A_bm = eng.to_batch_major(torch.from_numpy(A_eval).to(dev)); q_t = torch.from_numpy(q_eval).to(dev)
    x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(acceleration_lookback=0, eps=1e-10, max_iters=400000)))
    assert (status.cpu().numpy() == 1).all()
    res = certificate(A, b, c, cones, x.cpu().numpy(), y.cpu().numpy(), s.cpu().numpy())
    assert res.max() < 5e-8, res
