"""BASELINE config 1: the README example (/root/reference/README.md:84-101, tests/test_torch.py:41-59) as an ASSERTED test.

    min 1/2 ||A x - b||_1  s.t. x >= 0,   A (3, 2), b (3,) = torch.randn,   solution.sum().backward()

The reference's own test only smoke-runs it.  Canonical form: v = (x[2], t[3]), c = (0, 0, 1/2, 1/2, 1/2), eight
nonnegative rows (x >= 0, t - (A x - b) >= 0, t + (A x - b) >= 0): n = 5, m = 8, cone l = 8.  Checked unbatched (the README call) and as an
explicit batch of one (tests/test_torch.py:668-702 keeps that axis), against scipy's HiGHS for the value and, for the gradient, against
autograd through the linear system of the active constraints at the optimal vertex (an LP solution is locally the solution of those
n equalities; a generic random instance is non-degenerate)."""
import numpy as np
import pytest
import torch
from scipy.optimize import linprog

from cvxpylayers_amd.torch import VariableRecovery
from cvxpylayers_amd.torch.templates import template_from_affine_builder
from layer_backends import BACKENDS


def readme_template(n=2, m=3):
    def builder(A, b):
        nv = n + m
        G = np.zeros((n + 2 * m, nv)); h = np.zeros(n + 2 * m)
        G[:n, :n] = -np.eye(n)                                   # s = x
        G[n:n + m, :n] = A; G[n:n + m, n:] = -np.eye(m); h[n:n + m] = b          # s = t - (A x - b)
        G[n + m:, :n] = -A; G[n + m:, n:] = -np.eye(m); h[n + m:] = -b           # s = t + (A x - b)
        c = np.concatenate([np.zeros(n), 0.5 * np.ones(m)])
        return G, h, c
    return template_from_affine_builder(builder, [(m, n), (m,)], dict(z=0, l=n + 2 * m, q=[]), [VariableRecovery(slice(0, n), None, (n,))])


def _vertex_solution(A, b):
    """HiGHS solution + a differentiable re-solve of the active system (value identical, gradient by autograd)."""
    m, n = A.shape
    An, bn = A.detach().numpy(), b.detach().numpy()
    c = np.concatenate([np.zeros(n), 0.5 * np.ones(m)])
    G = np.block([[An, -np.eye(m)], [-An, -np.eye(m)]]); h = np.concatenate([bn, -bn])
    r = linprog(c, A_ub=G, b_ub=h, bounds=[(0, None)] * n + [(None, None)] * m, method="highs")
    assert r.status == 0
    x = r.x[:n]; res = An @ x - bn
    rows = [("x", i) for i in range(n) if abs(x[i]) < 1e-9] + [("r", j) for j in range(m) if abs(res[j]) < 1e-9]
    assert len(rows) == n, "degenerate vertex: pick another seed"
    M = torch.stack([torch.eye(n, dtype=torch.float64)[i] if k == "x" else A[i] for k, i in rows])
    rhs = torch.stack([torch.zeros((), dtype=torch.float64) if k == "x" else b[i] for k, i in rows])
    return x, torch.linalg.solve(M, rhs)


@pytest.mark.parametrize("make", BACKENDS)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_readme_l1_example_value_and_gradient(make, seed):
    torch.manual_seed(seed)
    A = torch.randn(3, 2, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, dtype=torch.float64, requires_grad=True)
    layer = make(readme_template(), eps=1e-10)
    solution, = layer(A, b)                                      # the README call: unbatched in, unbatched out
    assert tuple(solution.shape) == (2,)
    solution.sum().backward()
    gA, gb = A.grad.clone(), b.grad.clone()
    A.grad = None; b.grad = None
    x_ref, x_diff = _vertex_solution(A, b)
    if x_diff.requires_grad:
        x_diff.sum().backward()
    else:                                                        # vertex x = 0: locally constant
        A.grad = torch.zeros_like(A); b.grad = torch.zeros_like(b)
    np.testing.assert_allclose(solution.detach().cpu().numpy(), x_ref, atol=1e-7)
    np.testing.assert_allclose(gA.numpy(), A.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(gb.numpy(), b.grad.numpy(), atol=1e-5)
    # explicit batch of one keeps its axis and gives the same numbers
    A1 = A.detach()[None].clone().requires_grad_(True); b1 = b.detach()[None].clone().requires_grad_(True)
    sol1, = layer(A1, b1)
    assert tuple(sol1.shape) == (1, 2)
    sol1.sum().backward()
    np.testing.assert_allclose(sol1.detach().cpu().numpy()[0], x_ref, atol=1e-7)
    np.testing.assert_allclose(A1.grad.numpy()[0], A.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(b1.grad.numpy()[0], b.grad.numpy(), atol=1e-5)
