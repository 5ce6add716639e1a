"""N>1 path on CPU: world_size-2 gloo processes exercise the batch sharding of cvxpylayers_amd/parallel.py --
  * shard bounds, the differentiable all-gather of primal/dual rows with both gradient contracts, ragged shards (stand-in layer function);
  * the REAL plugin calling convention (tests/oracle_layer.OraclePlugin: same 7-argument apply / 7-tuple backward as MI355's and DIFFCP's plugin,
    solved by the CPU oracle) sharded over two ranks and checked against the fixtures recorded from the reference's own glue
    (tests/golden/refglue_*.npz): gathered primal / dual, this rank's columns of dA_eval / dq_eval, and -- through the frontend's flattening
    of an UNBATCHED parameter (torch/cvxpylayer.py:111-117: expand -> its gradient is a sum over the batch) -- allreduce_broadcast_grad;
  * bench.py --dry-run-ranks 2: the script's own self-spawn / rendezvous / max-over-ranks timing / JSON plumbing on CPU."""
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


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _real_plugin_worker(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import ref_cases
    from oracle_layer import OraclePlugin
    from cvxpylayers_amd.parallel import allreduce_broadcast_grad, shard_bounds, sharded_apply
    # ---- (1) plugin boundary, all parameters batched: the metric shape, B = 6 -> 3 instances per rank
    f = np.load(os.path.join(GOLD, "refglue_metric_shape.npz"))
    tpl = ref_cases.CASES["metric_shape"]()["template"]
    B = f["A_eval"].shape[1]
    A = torch.tensor(f["A_eval"], requires_grad=True); q = torch.tensor(f["q_eval"], requires_grad=True)
    primal, dual, info = sharded_apply(OraclePlugin, q, A, tpl, dict(ref_cases.SOLVER_ARGS), True, total=B)
    assert tuple(primal.shape) == f["primal"].shape and tuple(dual.shape) == f["dual"].shape
    assert np.abs(primal.detach().numpy() - f["primal"]).max() < 1e-8 and np.abs(dual.detach().numpy() - f["dual"]).max() < 1e-8
    ((primal * torch.tensor(f["dprimal"])).sum() + (dual * torch.tensor(f["ddual"])).sum()).backward()      # the same loss on every rank ("replicated")
    lo, hi = shard_bounds(B, rank, world)
    want_A = np.zeros_like(f["dA_eval"]); want_A[:, lo:hi] = f["dA_eval"][:, lo:hi]
    want_q = np.zeros_like(f["dq_eval"]); want_q[:, lo:hi] = f["dq_eval"][:, lo:hi]
    sc = 1.0 + np.abs(f["dA_eval"]).max()
    assert np.abs(A.grad.numpy() - want_A).max() < 1e-5 * sc and np.abs(q.grad.numpy() - want_q).max() < 1e-5 * sc
    # ---- (2) a broadcast (unbatched) parameter through the frontend's flattening: ridge LS, F unbatched, g batched, B = 5 (ragged 3 + 2)
    f = np.load(os.path.join(GOLD, "refglue_ridge_mixed.npz"))
    tpl = ref_cases.CASES["ridge_mixed"]()["template"]
    F = torch.tensor(f["param0"], requires_grad=True); g = torch.tensor(f["param1"], requires_grad=True)
    B = g.shape[0]
    flat = [F.t().reshape(-1)[None, :].expand(B, -1), g]                 # Fortran flattening; expand = the broadcast whose backward sums over the batch
    p = torch.zeros(B, tpl.n_params_total + 1, dtype=torch.float64); p[:, -1] = 1.0
    cols = []
    for v, off in zip(flat, tpl.col_offsets):
        cols.append((off, v))
    p = torch.cat([v for _, v in sorted(cols, key=lambda t: t[0])] + [torch.ones(B, 1, dtype=torch.float64)], dim=1)
    A_eval = torch.tensor(tpl.A_map.toarray()) @ p.t(); q_eval = torch.tensor(tpl.q_map.toarray()) @ p.t()
    np.testing.assert_allclose(A_eval.detach().numpy(), f["A_eval"], atol=1e-12)
    primal, dual, info = sharded_apply(OraclePlugin, q_eval, A_eval, tpl, dict(ref_cases.SOLVER_ARGS), True, total=B)
    x = primal[:, tpl.var_recover[0].primal]
    assert np.abs(x.detach().numpy() - f["out0"]).max() < 1e-8
    (x * torch.tensor(f["weight0"])).sum().backward()
    lo, hi = shard_bounds(B, rank, world)
    want_g = np.zeros_like(f["grad1"]); want_g[lo:hi] = f["grad1"][lo:hi]
    assert np.abs(g.grad.numpy() - want_g).max() < 1e-6                  # batched parameter: this rank's rows only
    assert np.abs(F.grad.numpy() - f["grad0"]).max() > 1e-3              # broadcast parameter: a PARTIAL sum before the all-reduce ...
    Fg = allreduce_broadcast_grad(F.grad.clone())
    assert np.abs(Fg.numpy() - f["grad0"]).max() < 1e-6                  # ... and the reference's gradient (sum over the whole batch) after it
    dist.destroy_process_group()


def test_world_size_2_gloo_with_the_real_plugin_convention_against_reference_fixtures():
    port = 29500 + (os.getpid() % 2000) + 7
    mp.spawn(_real_plugin_worker, args=(2, port), nprocs=2, join=True)


def _frontend_worker(rank, world, port):
    """The FRONTEND's own code on two ranks: cvxpylayers_amd.torch.CvxpyLayer.validate_params / _flatten_params (a broadcast parameter is expanded there:
    its gradient is a sum over the batch) and _recover_results_torch, around parallel.sharded_apply with the CPU oracle behind the plugin convention.  Only
    the two device-only pieces are replaced: the parameter maps are applied with the layer's own (re-indexed) scipy matrices instead of the HIP SpMM, the
    plugin is OraclePlugin instead of MI355's.  Checked against the fixture the REFERENCE's glue produced for this layer (outputs and both gradients)."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import ref_cases
    from oracle_layer import OraclePlugin
    from cvxpylayers_amd.parallel import allreduce_broadcast_grad, shard_bounds, sharded_apply
    from cvxpylayers_amd.torch import CvxpyLayer
    f = np.load(os.path.join(GOLD, "refglue_ridge_mixed.npz"))
    tpl = ref_cases.CASES["ridge_mixed"]()["template"]
    layer = CvxpyLayer(template=tpl)                                          # (construction needs no device: engines are created at the first GPU call)
    F = torch.tensor(f["param0"], requires_grad=True); g = torch.tensor(f["param1"], requires_grad=True)      # F unbatched (broadcast), g batched
    batch = layer.validate_params([F, g])
    B = batch[0]
    p_bm = layer._flatten_params((F, g), batch)                                # the frontend's flattening (expand of the unbatched parameter)
    A_bm = p_bm @ torch.tensor(layer._A.mat.toarray()).t(); q_bm = p_bm @ torch.tensor(layer._q.mat.toarray()).t()
    np.testing.assert_allclose(A_bm.t().detach().numpy(), f["A_eval"], atol=1e-12)     # what the reference's maps hand the plugin
    primal, dual, info = sharded_apply(OraclePlugin, q_bm.t(), A_bm.t(), tpl, dict(ref_cases.SOLVER_ARGS), True, total=B)
    x, = layer._recover_results_torch(primal, dual, batch)                     # the frontend's recovery
    assert np.abs(x.detach().numpy() - f["out0"]).max() < 1e-8
    (x * torch.tensor(f["weight0"])).sum().backward()                          # the same loss on every rank (bench.py's contract)
    lo, hi = shard_bounds(B, rank, world)
    want_g = np.zeros_like(f["grad1"]); want_g[lo:hi] = f["grad1"][lo:hi]
    assert np.abs(g.grad.numpy() - want_g).max() < 1e-6                        # batched parameter: this rank's rows, the others zero
    Fg = allreduce_broadcast_grad(F.grad.clone())
    assert np.abs(Fg.numpy() - f["grad0"]).max() < 1e-6                        # broadcast parameter: the single-process (reference) gradient after the all-reduce
    assert np.abs(F.grad.numpy() - f["grad0"]).max() > 1e-3                    # ... which this rank alone does not hold
    dist.destroy_process_group()


def test_world_size_2_gloo_through_the_frontend_with_a_broadcast_parameter():
    port = 29500 + (os.getpid() % 2000) + 11
    mp.spawn(_frontend_worker, args=(2, port), nprocs=2, join=True)


def test_bench_dry_run_ranks_exercises_the_self_spawn_plumbing():
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-ranks", "2", "--steps", "6", "--rotate", "4"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")})
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-400:], out.stderr[-800:])
    rec = json.loads(lines[0])
    assert rec["dry_run"] and rec["ranks"] == 2 and rec["gather_ok"] and rec["backend"] == "gloo" and rec["master_addr"] == "127.0.0.1"
    assert rec["rotating_batches"] == 4          # every rank visits the same slot of its K rotating batches at the same step (gather_ok checks the gathered sums per slot)


@pytest.mark.parametrize("scaling,config,batch,want_sizes", [("weak", "C3", 4096, [64] * 8), ("strong", "C5", 100, [13, 13, 13, 13, 12, 12, 12, 12])])      # (8 ranks x one torch import each: two cases keep the CPU suite short)
def test_bench_dry_run_eight_ranks_in_both_scaling_modes(scaling, config, batch, want_sizes):
    """`bench.py --scaling strong|weak --config C3|C5` under ranks.  BASELINE.json's "SOCP n=100, batch=4096, 1->8 GPU batch shard" is
    `--config C3 --scaling strong`, "portfolio n=500, batch=16384 sharded across 8 GPUs" is `--config C5 --batch 16384 --scaling strong`; eight CPU ranks over gloo
    run the script's own launch / shard / gather / timing plumbing (ragged shards included: 100 instances over 8 ranks).  No RCCL run with more than one rank exists."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-ranks", "8", "--steps", "3", "--rotate", "2", "--scaling", scaling, "--config", config, "--batch", str(batch)],
                         capture_output=True, text=True, timeout=600, env={**{k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}, "OMP_NUM_THREADS": "1"})
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-400:], out.stderr[-800:])
    rec = json.loads(lines[0])
    assert rec["dry_run"] and rec["ranks"] == 8 and rec["gather_ok"] and rec["scaling"] == scaling and rec["config"] == config
    assert rec["shard_sizes"] == want_sizes
