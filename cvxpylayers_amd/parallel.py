"""Multi-GPU batch sharding (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Instances are independent: every rank solves a contiguous slice of the batch with no
data-path collective.  The only exchange step the path has is reassembling solutions into one autograd
graph: an all-gather of primal/dual rows.  Each rank ends up with the gradient of ITS shard, and dA/dq for
batched parameters stay sharded.  Gradients of broadcast (unbatched) parameters are sums over the batch
(expand backward, torch/cvxpylayer.py:111-117) -> `allreduce_broadcast_grad`.

GRADIENT CONTRACT of the gather (who computes the loss) -- `loss=` of gather_rows / gather_solution / sharded_apply:
  "replicated" (default)  every rank evaluates the SAME scalar loss L on the gathered tensor (the SPMD pattern: identical code
                          on every rank, as bench.py does).  The gradient of L with respect to this rank's rows is simply this
                          rank's slice of the incoming gradient: the backward needs NO collective and no factor.
  "partial"               rank r evaluates its own term L_r on the gathered tensor and the objective is L = sum_r L_r
                          (e.g. the loss lives on rank 0 only and the others contribute 0, or every rank scores a different
                          slice of a downstream batch).  The backward is a reduce-scatter(sum) of the incoming gradients.
                          Using "partial" with a replicated loss would scale every gradient by the world size.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of `total` instances for `rank`."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _AllGatherRows(torch.autograd.Function):
    """out = concat over ranks of x (rows);  backward = this rank's slice of the incoming gradient (loss "replicated") or the
    reduce-scatter(sum) of the incoming gradients (loss "partial"); see the module docstring."""

    @staticmethod
    def forward(ctx, x, sizes, group, loss="replicated"):
        if loss not in ("replicated", "partial"):
            raise ValueError(f"loss must be 'replicated' or 'partial', got {loss!r}")
        world = dist.get_world_size(group)
        ctx.group, ctx.sizes, ctx.rank, ctx.loss = group, sizes, dist.get_rank(group), loss
        x = x.contiguous()
        if len(set(sizes)) == 1:
            out = torch.empty((sizes[0] * world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(out, x, group=group)
            return out
        # ragged shards: pad every shard to the largest one (collectives need equal sizes), gather, drop the pads
        smax = max(sizes)
        pad = torch.zeros((smax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[:x.shape[0]] = x
        out = torch.empty((smax * world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, pad, group=group)
        return torch.cat([out[r * smax:r * smax + sizes[r]] for r in range(world)], dim=0)

    @staticmethod
    def backward(ctx, g):
        sizes, rank = ctx.sizes, ctx.rank
        lo = sum(sizes[:rank])
        if ctx.loss == "replicated":       # every rank holds the same gradient: no exchange
            return g[lo:lo + sizes[rank]], None, None, None
        g = g.contiguous()
        backend = dist.get_backend(ctx.group)
        if backend == "nccl" and len(set(sizes)) == 1:
            out = torch.empty((sizes[0],) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=ctx.group)
            return out, None, None, None
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)   # gloo / ragged: all-reduce then slice
        return g[lo:lo + sizes[rank]].clone(), None, None, None


def gather_rows(x: torch.Tensor, sizes=None, group=None, loss: str = "replicated") -> torch.Tensor:
    """All-gather the leading (batch) axis across ranks, differentiable (`loss`: module docstring)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return x
    world = dist.get_world_size(group)
    if sizes is None:
        sizes = [x.shape[0]] * world
    return _AllGatherRows.apply(x, list(sizes), group, loss)


def gather_solution(primal: torch.Tensor, dual: torch.Tensor, sizes=None, group=None, loss: str = "replicated"):
    """primal (b, n) and dual (b, m) of every rank in ONE collective: the rows are fused into a (b, n + m) buffer, all-gathered
    once (the messages are latency-bound: 600 KB per rank at the metric configuration) and split again.
    Differentiable like gather_rows."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return primal, dual
    n = primal.shape[1]
    both = gather_rows(torch.cat([primal, dual], dim=1), sizes, group, loss)
    return both[:, :n], both[:, n:]


def allreduce_broadcast_grad(g: torch.Tensor, group=None) -> torch.Tensor:
    """Sum a broadcast-parameter gradient over ranks (each rank holds the sum over its shard)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g


def sharded_apply(layer_cls, q_eval, A_eval, ctx, solver_args, needs_grad=True, total=None, gather=True, group=None, P_eval=None,
                  loss: str = "replicated"):
    """Solve this rank's slice of a replicated (…, B_total) batch and (optionally) all-gather primal/dual.

    q_eval (n+1, B_total), A_eval (nnz_aug, B_total) (and P_eval (nnz_p, B_total) for a quadratic objective) replicated on every
    rank (or pass already-local slices with total=None).  Returns primal (B_total, n), dual (B_total, m) when gather else the
    local rows.  `loss`: who evaluates the loss on the gathered tensors (module docstring)."""
    if dist.is_initialized() and total is not None:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        lo, hi = shard_bounds(total, rank, world)
        q_eval = q_eval[:, lo:hi]
        A_eval = A_eval[:, lo:hi]
        P_eval = P_eval[:, lo:hi] if P_eval is not None else None
        sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    else:
        sizes = None
    primal, dual, info, data = layer_cls.apply(P_eval, q_eval, A_eval, ctx, solver_args, needs_grad, None)
    if gather:
        primal, dual = gather_solution(primal, dual, sizes, group, loss)
    return primal, dual, info
