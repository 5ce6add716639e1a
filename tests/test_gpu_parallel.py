"""The N>1 plumbing with the REAL plugin on the GPU, under a one-rank RCCL group (the GPU box has one device; the world-size-2
logic is covered on CPU by tests/test_parallel_gloo.py): sharded_apply -> _CvxpyLayer.apply -> gather, and the collectives of
_AllGatherRows themselves (all_gather_into_tensor forward, reduce_scatter_tensor backward) on backend "nccl" (= RCCL)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from cvxpylayers_amd import problems as P

pytestmark = pytest.mark.gpu


@pytest.fixture
def rccl_one_rank():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 2000)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")          # one node, rendezvous on 127.0.0.1: no interface probing (RCCL's bootstrap took 31 s on one box of round 5)
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_sharded_apply_with_the_real_plugin_under_rccl(rccl_one_rank):
    from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
    from cvxpylayers_amd.parallel import _AllGatherRows, sharded_apply
    cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 64
    tpl = P.dense_template(n, cones)
    A, b, c = P.generate(n, cones, B, seed=21)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-9})
    A_t = torch.from_numpy(A_eval).cuda().requires_grad_(); q_t = torch.from_numpy(q_eval).cuda().requires_grad_()
    primal, dual, info = sharded_apply(_CvxpyLayer, q_t, A_t, ctx, {}, True, total=B)
    assert primal.shape == (B, n) and dual.shape == (B, tpl.m) and bool((info["status"] == 1).all())
    primal.sum().backward()
    gA, gq = A_t.grad.clone(), q_t.grad.clone()
    # the same through the plugin alone
    A2 = torch.from_numpy(A_eval).cuda().requires_grad_(); q2 = torch.from_numpy(q_eval).cuda().requires_grad_()
    p2, d2, *_ = _CvxpyLayer.apply(None, q2, A2, ctx, {}, True, None)
    p2.sum().backward()
    assert torch.equal(primal, p2) and torch.allclose(gA, A2.grad, rtol=1e-12, atol=1e-14) and torch.allclose(gq, q2.grad, rtol=1e-12, atol=1e-14)
    # the collectives themselves on RCCL (world 1: identity, but the nccl code paths run): both gradient contracts
    for loss in ("replicated", "partial"):
        x = torch.randn(B, n + tpl.m, dtype=torch.float64, device="cuda", requires_grad=True)
        out = _AllGatherRows.apply(x, [B], None, loss)
        assert torch.equal(out, x)
        w = torch.randn_like(out)
        (out * w).sum().backward()
        assert torch.allclose(x.grad, w)
