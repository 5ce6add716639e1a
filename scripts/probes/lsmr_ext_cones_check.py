"""mode="lsmr" on a per-instance template with every cone type (PSD, exponential, power): engine against the oracle's LSMR (debug aid / one-off check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
from oracle import oracle
n, cones, B = 10, {"z": 2, "l": 3, "q": [4], "s": [3], "ep": 2, "p": [0.4, -0.7]}, 12
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
ref = oracle.solve_batch(A, b, c, cones, eps=1e-9, max_iters=200000)
ok = ref["status"] == 1
A_eval, q_eval = tpl.values_from_dense(A, b, c)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
xo, yo, so = (torch.from_numpy(ref[k]).cuda() for k in ("x", "y", "s"))
rng = np.random.default_rng(1); dx = rng.standard_normal((B, n)); dy = rng.standard_normal(ref["y"].shape)
for tol in (1e-12, 1e-8):
  for meth in ("lsqr", "lsmr"):
    dA, dq, adj = eng.vjp(A_bm, xo, yo, so, torch.from_numpy(dx).cuda(), torch.from_numpy(dy).cuda(), path="per_instance_lsqr", lsqr=(tol, tol, 20000, "full", meth), q_eval=q_t)
    torch.cuda.synchronize()
    g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode=meth, lsqr_atol=tol, lsqr_btol=tol, lsqr_iter_lim=20000)
    err = np.abs(dq.cpu().numpy()[:n].T - g["dc"]).max(axis=1) / (1 + np.abs(g["dc"]).max(axis=1))
    print(tol, meth, "max rel dc err", err[ok].max(), "engine its", eng.last_lsqr_iters.cpu().numpy()[ok], "oracle its", g["lsqr_iters"][ok])
for conlim in (1e8, 0.0):
    for tol in (1e-12, 1e-8):
        g = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsmr", lsqr_atol=tol, lsqr_btol=tol, lsqr_iter_lim=20000, lsqr_conlim=conlim)
        g2 = oracle.adjoint_batch(A, b, c, cones, ref["x"], ref["y"], ref["s"], dx, dy, mode="lsqr", lsqr_atol=tol, lsqr_btol=tol, lsqr_iter_lim=20000, lsqr_conlim=conlim)
        print("oracle conlim", conlim, "tol", tol, "lsmr its", g["lsqr_iters"], "lsqr its", g2["lsqr_iters"])
