"""Seeded layer cases shared by the reference-glue fixture generator (tests/golden/make_refglue.py, runs the REFERENCE's own
Python code here) and by the tests that replay them (tests/test_ref_glue.py on CPU, tests/test_gpu_refglue.py on the GPU box).

Nothing in this file touches /root/reference: a case is a hand-canonicalised template (affine probing of a solver-form
builder, cvxpylayers_amd.torch.templates) + seeded parameter values + fixed loss weights.  Each case restates a problem family
of the reference's own tests (file:line in the docstrings)."""
from __future__ import annotations

import numpy as np

import kit
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.torch import VariableRecovery
from cvxpylayers_amd.torch.templates import template_from_affine_builder

SOLVER_ARGS = {"eps": 1e-10, "max_iters": 200000}


def _ridge(mF, n, col_order=None):
    def builder(F, g):
        A, b, c, cones, _ = kit.ridge_ls(F, g)
        return A, b, c
    cones = {"z": 0, "l": 0, "q": [mF + 2, n + 2]}
    return template_from_affine_builder(builder, [(mF, n), (mF,)], cones, [VariableRecovery(slice(0, n), None, (n,))], col_order=col_order)


def case_ridge_mixed():
    """tests/test_torch.py:355-384: ridge LS, F unbatched (broadcast), g batched; canonical column order != user order."""
    rng = np.random.default_rng(243)
    mF, n, B = 12, 4, 5
    return dict(template=_ridge(mF, n, col_order=[1, 0]), params=[rng.standard_normal((mF, n)), rng.standard_normal((B, mF))],
                weights=[rng.standard_normal((B, n))])


def case_ridge_batched_matrix_param():
    """tests/test_torch.py:90-118: ridge LS with the matrix parameter batched too (Fortran flattening of a batched matrix)."""
    rng = np.random.default_rng(7)
    mF, n, B = 9, 3, 4
    return dict(template=_ridge(mF, n), params=[rng.standard_normal((B, mF, n)), rng.standard_normal((B, mF))], weights=[rng.standard_normal((B, n))])


def case_ridge_unbatched():
    """tests/test_torch.py:41-59 (README example shape): no batch axis anywhere -> outputs without batch axis, gradients squeezed."""
    rng = np.random.default_rng(11)
    mF, n = 8, 3
    return dict(template=_ridge(mF, n), params=[rng.standard_normal((mF, n)), rng.standard_normal(mF)], weights=[rng.standard_normal(n)])


def case_matrix_variable():
    """tests/test_torch.py:755-780: matrix variable recovered column-major.  min ||A X - B||_F^2 + ||X||_F^2."""
    rng = np.random.default_rng(123)
    m, n, k, B = 7, 3, 2, 3

    def builder(A_, B_):
        F = np.kron(np.eye(k), A_)
        A, b, c, cones, _ = kit.ridge_ls(F, B_.reshape(-1, order="F"))
        return A, b, c
    cones = kit.ridge_ls(np.zeros((m * k, n * k)), np.zeros(m * k))[3]
    tpl = template_from_affine_builder(builder, [(m, n), (m, k)], cones, [VariableRecovery(slice(0, n * k), None, (n, k))])
    return dict(template=tpl, params=[rng.standard_normal((B, m, n)), rng.standard_normal((B, m, k))], weights=[rng.standard_normal((B, n, k))])


def case_sdp_symmetric_primal_and_psd_dual():
    """tests/test_dual_variables.py:523-550, tests/test_torch.py:233-248: min tr(C X) s.t. tr X = 1, X PSD with the canonical
    variable = upper triangle of X, row-major, unscaled ("svec_primal" recovery, torch/cvxpylayer.py:183-198) and the PSD
    constraint's dual returned too ("svec_dual": lower triangle, column-major, sqrt(2) off-diagonals, :201-222)."""
    rng = np.random.default_rng(5)
    k, B = 4, 3
    d = k * (k + 1) // 2
    iu = np.triu_indices(k)                                     # canonical variable order: (0,0),(0,1),...,(0,k-1),(1,1),...
    pos = {(int(i), int(j)): p for p, (i, j) in enumerate(zip(*iu))}
    sv = kit.svec_index(k)                                      # svec position of (i, j): lower triangle, column-major

    def builder(Cm):
        Cs = 0.5 * (Cm + Cm.T)
        c = np.array([Cs[i, j] * (1.0 if i == j else 2.0) for i, j in zip(*iu)])
        A = np.zeros((1 + d, d)); b = np.zeros(1 + d); b[0] = 1.0
        for (i, j), p in pos.items():
            if i == j:
                A[0, p] = 1.0                                   # tr X = 1
            A[1 + sv[(i, j)], p] = -1.0 if i == j else -np.sqrt(2.0)     # s = svec(X)
        return A, b, c
    tpl = template_from_affine_builder(builder, [(k, k)], {"z": 1, "l": 0, "q": [], "s": [k]},
                                       [VariableRecovery(slice(0, d), None, (k, k), source="primal", unpack_fn="svec_primal"),
                                        VariableRecovery(None, slice(1, 1 + d), (k, k), source="dual", unpack_fn="svec_dual")])
    G = rng.standard_normal((B, k, k))
    C = 0.5 * (G + np.swapaxes(G, 1, 2)) + np.diag(np.arange(k) * 0.7)[None]
    W = rng.standard_normal((B, k, k)); W = 0.5 * (W + np.swapaxes(W, 1, 2))
    return dict(template=tpl, params=[C], weights=[W, 0.3 * W[::-1].copy()])


def case_metric_shape():
    """BASELINE metric configuration (n=50, m=100: 20 nonneg rows + 8 SOC(10), dense A) as a layer whose parameters are the
    solver-form A (m, n), b (m,), c (n,) themselves, all batched; B = 6 instances."""
    cfg = P.CONFIGS["M"]; n, cones = cfg["n"], cfg["cones"]; m = P.cone_rows(cones)
    tpl = template_from_affine_builder(lambda A, b, c: (A, b, c), [(m, n), (m,), (n,)], cones,
                                       [VariableRecovery(slice(0, n), None, (n,)), VariableRecovery(None, slice(0, m), (m,), source="dual")])
    A, b, c = P.generate(n, cones, 6, seed=31)
    rng = np.random.default_rng(32)
    return dict(template=tpl, params=[A, b, c], weights=[rng.standard_normal((6, n)), 0.1 * rng.standard_normal((6, m))])


CASES = {
    "ridge_mixed": case_ridge_mixed,
    "ridge_batched_matrix_param": case_ridge_batched_matrix_param,
    "ridge_unbatched": case_ridge_unbatched,
    "matrix_variable": case_matrix_variable,
    "sdp_sym_primal_psd_dual": case_sdp_symmetric_primal_and_psd_dual,
    "metric_shape": case_metric_shape,
}
