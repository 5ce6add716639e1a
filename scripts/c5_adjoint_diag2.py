import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine
from cvxpylayers_amd.interfaces import const_a
from oracle import oracle
B = 64
A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=0)
Ab = np.broadcast_to(A, (B,) + A.shape).copy(); bb = np.broadcast_to(b, (B,) + b.shape).copy()
n, m = tpl.n, tpl.m
ref = oracle.solve_batch(Ab, bb, c, cones, eps=1e-6, max_iters=100000)
gd = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], np.ones((B, n)), np.zeros((B, m)), mode="dense")
A_eval, _ = tpl.values_from_dense(Ab, bb, c)
eng = ConeEngine(tpl.indices, tpl.indptr, n, m, cones, torch.device("cuda", 0))
A_bm = torch.from_numpy(A_eval).cuda().t().contiguous()
xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
dA, dq, adj = const_a.vjp_const_a(eng, A_bm, xo, yo, so, torch.ones_like(xo), torch.zeros_like(yo))
dc = dq.cpu().numpy()[:n].T
sc = 1 + np.abs(gd["dc"]).max(axis=1)
e = np.abs(dc - gd["dc"]).max(axis=1) / sc
bad = np.argsort(e)[-4:]
print("worst", bad, e[bad], "flagged", adj.cpu().numpy()[bad])
v = ref["y"] - ref["s"]
for i in bad:
    print(i, "min|v| nonneg", np.abs(v[i, 1:501]).min(), "soc t-|z|", v[i, 501] - np.linalg.norm(v[i, 502:]), "#active", int((v[i,1:501] > 0).sum()), "|dc| max", np.abs(gd["dc"][i]).max(), np.abs(dc[i]).max())
