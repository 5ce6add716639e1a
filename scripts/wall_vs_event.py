"""Wall-clock (host, synchronised) against HIP-event time of the forward / backward launches: exposes launch-side stalls (scratch
allocation, lazy module load) that event timing inside the engine does not see."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
cfgname = sys.argv[1] if len(sys.argv) > 1 else "E"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = P.CONFIGS[cfgname]; n, cones = cfg["n"], cfg["cones"]
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
st = make_settings(dict(eps=1e-4, max_iters=20000))
eng.set_profiling(True)
out = eng.solve(A_bm, q_t, st); torch.cuda.synchronize(); eng.reset_profile()
t0 = time.perf_counter()
for _ in range(5): out = eng.solve(A_bm, q_t, st)
torch.cuda.synchronize(); wall_f = (time.perf_counter() - t0) / 5
ev_f = eng.profile(0)[0]
x, y, s = out[0], out[1], out[2]; dx = torch.ones_like(x); dy = torch.zeros_like(y)
eng.vjp(A_bm, x, y, s, dx, dy); torch.cuda.synchronize(); eng.reset_profile()
t0 = time.perf_counter()
for _ in range(5): eng.vjp(A_bm, x, y, s, dx, dy)
torch.cuda.synchronize(); wall_b = (time.perf_counter() - t0) / 5
ev_b = eng.profile(1)[0]
print(f"{cfgname}: forward wall {wall_f*1e3:.3f} ms / event {ev_f:.3f} ms ; backward wall {wall_b*1e3:.3f} ms / event {ev_b:.3f} ms")
