"""Runs the forward kernel at two fixed iteration counts so that PMC deltas give per-iteration costs."""
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
for mi in (101, 201):
    st = make_settings(dict(eps=0.0, eps_infeas=0.0, max_iters=mi, adaptive_scale=0))
    out = eng.solve(A_bm, q_t, st)
torch.cuda.synchronize()
