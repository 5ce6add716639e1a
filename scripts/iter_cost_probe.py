"""Per-iteration cost of k_fwd2 on the metric configuration: fixed iteration counts (eps so small that nothing converges), slope of the kernel time.
   CE_ENGINE_SO selects the library (A/B of experimental builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
name = sys.argv[1] if len(sys.argv) > 1 else "M"
cfg = P.CONFIGS[name]; n, cones = cfg["n"], cfg["cones"]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
res = {}
for mi in (1, 26, 76, 151, 76, 1):
    st = make_settings(dict(eps=1e-300, max_iters=mi, acceleration_lookback=0))
    for _ in range(3): eng.solve(A_bm, q_t, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = eng.solve(A_bm, q_t, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res.setdefault(mi, []).append(ms)
    print(f"max_iters {mi:4d}: {ms:8.4f} ms   iters mean {out[3].float().mean().item():.1f}", flush=True)
t1, t76, t151 = min(res[1]), min(res[76]), min(res[151])
print(f"{os.environ.get('CE_ENGINE_SO', 'default')}: setup {t1:.4f} ms, per iteration {(t151 - t76) / 75 * 1e3:.3f} us (x B={B}), 75 iterations {t76 - t1:.4f} ms")
