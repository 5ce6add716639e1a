"""How much of k_fwd2's time at the metric configuration is the spread of the iteration counts (tail / imbalance)?
   real run (eps 1e-4) vs a run where EVERY instance does the mean number of iterations (eps tiny, max_iters = mean)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
cfg = P.CONFIGS["M"]; n, cones = cfg["n"], cfg["cones"]; B = 4096
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
def timeit(st, reps=20):
    for _ in range(3): out = eng.solve(A_bm, q_t, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = eng.solve(A_bm, q_t, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out[3].float()
for look in (10, 0):
    ms, it = timeit(make_settings(dict(eps=1e-4, max_iters=10000, acceleration_lookback=look)))
    vals, cnt = np.unique(it.cpu().numpy(), return_counts=True)
    mean = float(it.mean())
    ms_u, it_u = timeit(make_settings(dict(eps=1e-300, max_iters=int(round(mean)), acceleration_lookback=look)))
    print(f"lookback {look}: real {ms:.4f} ms (mean {mean:.1f} iterations, histogram {dict(zip(vals.astype(int).tolist(), cnt.tolist()))}) | every instance {int(round(mean))} iterations: {ms_u:.4f} ms | spread costs {ms - ms_u:.4f} ms", flush=True)
