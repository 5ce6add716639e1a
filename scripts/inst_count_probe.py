"""One fixed-iteration solve configuration for counter passes: python scripts/inst_count_probe.py <config> <max_iters> [vjp]
   (eps so small that nothing converges, acceleration off: `max_iters` plain iterations incl. a check every 25).  Run under
   rocprofv3 --pmc ...; scripts/inst_count_summary.py turns the passes into instructions per wave and per wave-iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
name = sys.argv[1]; mi = int(sys.argv[2]); B = 4096
cfg = P.CONFIGS[name]; n, cones = cfg["n"], cfg["cones"]
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
st = make_settings(dict(eps=1e-300, max_iters=mi, acceleration_lookback=0))
for _ in range(4):
    out = eng.solve(A_bm, q_t, st)
torch.cuda.synchronize()
print("iters mean", out[3].float().mean().item())
