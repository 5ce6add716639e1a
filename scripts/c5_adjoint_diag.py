import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine
from cvxpylayers_amd.interfaces import const_a
from oracle import oracle
B = 32
A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=0)
Ab = np.broadcast_to(A, (B,) + A.shape).copy(); bb = np.broadcast_to(b, (B,) + b.shape).copy()
n, m = tpl.n, tpl.m
ref = oracle.solve_batch(Ab, bb, c, cones, eps=1e-6, max_iters=100000)
gd = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], np.ones((B, n)), np.zeros((B, m)), mode="dense")
gl = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], np.ones((B, n)), np.zeros((B, m)), mode="lsqr")
A_eval, _ = tpl.values_from_dense(Ab, bb, c)
eng = ConeEngine(tpl.indices, tpl.indptr, n, m, cones, torch.device("cuda", 0))
A_bm = torch.from_numpy(A_eval).cuda().t().contiguous()
xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
_, q_eval = tpl.values_from_dense(Ab, bb, c)
q_t = torch.from_numpy(q_eval).cuda()
N = n + m + 1
for fac in (2, 16):          # LSQR iteration limit = fac * N (diffcp: 2 N) at diffcp's atol = btol = 1e-8 on the full system
    dA, dq, adj = const_a.vjp_const_a(eng, A_bm, xo, yo, so, torch.ones_like(xo), torch.zeros_like(yo), atol=1e-8, btol=1e-8, iter_lim=fac * N, q_eval=q_t)
    dc = dq.cpu().numpy()[:n].T
    sc = 1 + np.abs(gd["dc"]).max(axis=1)
    e_d = np.abs(dc - gd["dc"]).max(axis=1) / sc; e_l = np.abs(dc - gl["dc"]).max(axis=1) / sc; e_ld = np.abs(gl["dc"] - gd["dc"]).max(axis=1) / sc
    print("iter_lim / N", fac, "flagged", int(adj.sum()))
    print(" gpu vs dense :", np.sort(e_d)[[0, B // 2, -3, -2, -1]])
    print(" gpu vs o-lsqr:", np.sort(e_l)[[0, B // 2, -3, -2, -1]])
    print(" o-lsqr vs dense:", np.sort(e_ld)[[0, B // 2, -3, -2, -1]])
v = ref["y"] - ref["s"]
print("min |v| over nonneg rows per instance (degeneracy indicator):", np.sort(np.abs(v[:, 1:501]).min(axis=1))[:5])
