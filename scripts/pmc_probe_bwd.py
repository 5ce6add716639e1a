"""Runs forward once (eps 1e-4) then the backward kernel twice, for rocprofv3 PMC collection."""
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
x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-4, max_iters=10000)))
dx = torch.ones_like(x); dy = torch.zeros_like(y)
for _ in range(2):
    eng.vjp(A_bm, x, y, s, dx, dy)
torch.cuda.synchronize()
v = (y - s)
print("mean active nonneg rows", float((v[:, :20] > 0).float().sum(1).mean()))
