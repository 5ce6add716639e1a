"""N>1 path on CPU: world_size-2 gloo processes exercise the batch sharding helpers of cvxpylayers_amd/parallel.py
(shard bounds, differentiable all-gather of primal/dual rows with its reduce-scatter backward, all-reduce of
broadcast-parameter gradients, sharded_apply plumbing with a stand-in layer function)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    from cvxpylayers_amd.parallel import shard_bounds
    for total in (1, 7, 8, 4096, 16385):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeLayer(torch.autograd.Function):
    """Stand-in with the plugin's calling convention: primal = (A_eval * w).sum(0) rows, dual = q_eval rows."""

    @staticmethod
    def forward(ctx, P_eval, q_eval, A_eval, cl_ctx, solver_args, needs_grad, warm_start):
        ctx.save_for_backward(A_eval, q_eval)
        primal = (A_eval * 2.0).t().contiguous()          # (B, K)
        dual = (q_eval * 3.0).t().contiguous()            # (B, n+1)
        return primal, dual, {"iters": None}, None

    @staticmethod
    def backward(ctx, dprimal, ddual, _i, _d):
        return None, 3.0 * ddual.t(), 2.0 * dprimal.t(), None, None, None, None


def _worker(rank, world, port, ragged):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cvxpylayers_amd.parallel import allreduce_broadcast_grad, gather_rows, shard_bounds, sharded_apply
    torch.manual_seed(0)                         # replicated inputs
    total = 7 if ragged else 8
    A = torch.randn(5, total, dtype=torch.double, requires_grad=True)
    q = torch.randn(3, total, dtype=torch.double, requires_grad=True)
    wts = torch.arange(1, total + 1, dtype=torch.double)[:, None]
    lo, hi = shard_bounds(total, rank, world)
    want = torch.zeros_like(A)
    want[:, lo:hi] = 2.0 * wts[lo:hi, 0][None, :]              # d/dA of sum(primal * wts) restricted to this rank's shard
    # (1) loss="replicated" (default): every rank evaluates the same loss on the gathered tensor; each rank's shard gradient is
    #     the true gradient, no factor, no collective in the backward
    primal, dual, info = sharded_apply(_FakeLayer, q, A, None, {}, True, total=total)
    assert primal.shape == (total, 5) and dual.shape == (total, 3)
    assert torch.allclose(primal, 2.0 * A.detach().t()) and torch.allclose(dual, 3.0 * q.detach().t())
    (primal * wts).sum().backward()
    assert torch.allclose(A.grad, want), (A.grad, want)
    # (2) loss="partial": the loss lives on rank 0 only (the other ranks contribute a zero term); same gradients
    A.grad = None
    primal, dual, info = sharded_apply(_FakeLayer, q, A, None, {}, True, total=total, loss="partial")
    ((primal * wts).sum() * (1.0 if rank == 0 else 0.0)).backward()
    assert torch.allclose(A.grad, want), (A.grad, want)
    # (3) loss="partial" with every rank scoring its own half of the rows: L = sum_r L_r
    A.grad = None
    primal, dual, info = sharded_apply(_FakeLayer, q, A, None, {}, True, total=total, loss="partial")
    mine = torch.zeros(total, 1, dtype=torch.double); mine[rank::world] = 1.0
    (primal * wts * mine).sum().backward()
    assert torch.allclose(A.grad, want), (A.grad, want)
    # broadcast-parameter gradient: sum over the batch = sum over ranks of shard sums
    g = allreduce_broadcast_grad(A.grad.sum(dim=1))
    assert torch.allclose(g, 2.0 * wts.sum() * torch.ones(5, dtype=torch.double))
    # plain gather of local rows
    loc = torch.full((hi - lo, 2), float(rank), dtype=torch.double)
    sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    full = gather_rows(loc, sizes)
    assert full.shape[0] == total and float(full[0, 0]) == 0.0 and float(full[-1, 0]) == world - 1
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_world_size_2_gloo(ragged):
    port = 29500 + (os.getpid() % 2000) + (1 if ragged else 0)
    mp.spawn(_worker, args=(2, port, ragged), nprocs=2, join=True)
