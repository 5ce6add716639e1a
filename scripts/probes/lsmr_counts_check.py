"""LSQR / LSMR iteration counts, engine against oracle, on the small shared-A portfolio template and on the metric shape (debug aid for tests/test_gpu_lsqr_mode.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine
from oracle import oracle
os.environ["CE_CONST_A"] = "1"
B = 6
A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=3, nw=60, kf=9)
Ab = np.broadcast_to(A, (B,) + A.shape).copy(); bb = np.broadcast_to(b, (B,) + b.shape).copy()
ref = oracle.solve_batch(Ab, bb, c, cones, eps=1e-8, max_iters=200000)
A_eval, q_eval = tpl.values_from_dense(Ab, bb, c)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
dx = np.random.default_rng(2).standard_normal((B, tpl.n)); dy = np.zeros_like(ref["y"])
N = tpl.n + tpl.m + 1
for meth in ("lsqr", "lsmr"):
    for tol in (1e-6, 1e-8, 1e-10):
        dA, dq, adj = eng.vjp(A_bm, xo, yo, so, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), path="const_a", lsqr=(tol, tol, 2 * N, "full", meth), q_eval=q_t)
        torch.cuda.synchronize()
        g = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode=meth, lsqr_atol=tol, lsqr_btol=tol, lsqr_iter_lim=2 * N)
        print(meth, tol, "engine", eng.last_lsqr_iters.cpu().numpy(), "oracle", g["lsqr_iters"], "max|dc diff|", np.abs(dq.cpu().numpy()[:tpl.n] - g["dc"].T).max())
from cvxpylayers_amd.interfaces.const_a import vjp_const_a
from cvxpylayers_amd import _lib
print("-- conlim off (engine only), tol 1e-10 / 1e-8")
for meth in ("lsqr", "lsmr"):
    for tol in (1e-8, 1e-10):
        _lib.lib().ce_set_lsqr_variant(eng._h, 1 if meth == "lsmr" else 0)
        dA, dq, adj = vjp_const_a(eng, A_bm, xo, yo, so, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), atol=tol, btol=tol, iter_lim=2 * N, q_eval=q_t, conlim=0.0)
        torch.cuda.synchronize()
        _lib.lib().ce_set_lsqr_variant(eng._h, 0)
        g = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode=meth, lsqr_atol=tol, lsqr_btol=tol, lsqr_iter_lim=2 * N, lsqr_conlim=0.0)
        print(meth, tol, "engine", eng.last_lsqr_iters.cpu().numpy(), "oracle", g["lsqr_iters"])
print("-- fixed iteration counts (atol = btol = conlim = 0): iterates engine vs oracle")
for meth in ("lsqr", "lsmr"):
    for k in (3, 10, 40, 90):
        _lib.lib().ce_set_lsqr_variant(eng._h, 1 if meth == "lsmr" else 0)
        dA, dq, adj = vjp_const_a(eng, A_bm, xo, yo, so, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), atol=0.0, btol=0.0, iter_lim=k, q_eval=q_t, conlim=0.0)
        torch.cuda.synchronize()
        _lib.lib().ce_set_lsqr_variant(eng._h, 0)
        g = oracle.adjoint_batch(Ab, bb, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode=meth, lsqr_atol=0.0, lsqr_btol=0.0, lsqr_iter_lim=k, lsqr_conlim=0.0)
        print(meth, k, "engine its", eng.last_lsqr_iters.cpu().numpy(), "oracle its", g["lsqr_iters"], "max|dc diff|", np.abs(dq.cpu().numpy()[:tpl.n] - g["dc"].T).max(), "scale", np.abs(g["dc"]).max())
