"""k_backward_rt at 1, 2, 3 workgroups per CU and at the full batch: is the kernel bound by the latency of ONE instance (time ~ flat up to 768 instances) or by a
shared resource (time grows with the workgroups per CU)?   usage: python scripts/bwd_occupancy_probe.py [config]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
name = sys.argv[1] if len(sys.argv) > 1 else "M"
cfg = P.CONFIGS[name]; n, cones = cfg["n"], cfg["cones"]
tpl = P.dense_template(n, cones)
dev = torch.device("cuda", 0)
for B in (256, 512, 768, 1536, 4096):
    A, b, c = P.generate(n, cones, B, seed=0)
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
    A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
    x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-4, max_iters=10000)))
    dx = torch.ones_like(x); dy = torch.zeros_like(y)
    for _ in range(3): eng.vjp(A_bm, x, y, s, dx, dy)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.vjp(A_bm, x, y, s, dx, dy)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} B={B:5d}: vjp {e0.elapsed_time(e1) / 20:8.4f} ms  ({B / 256:.1f} workgroups per CU)", flush=True)
