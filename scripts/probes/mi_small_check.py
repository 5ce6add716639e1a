"""k_sa_lsqr_mi against k_sa_lsqr on a small portfolio template (debug aid for tests/test_gpu_atsize.py's NI test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
os.environ["CE_CONST_A"] = "1"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=3, nw=60, kf=9)
Ab = np.broadcast_to(A, (B,) + A.shape).copy(); bb = np.broadcast_to(b, (B,) + b.shape).copy()
A_eval, q_eval = tpl.values_from_dense(Ab, bb, c)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0))
A_bm = torch.from_numpy(A_eval).cuda().t().contiguous(); q_t = torch.from_numpy(q_eval).cuda()
x, y, s, it, status, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-7, max_iters=100000)))
print("path", eng.last_path, "status", status.cpu().numpy(), "iters", it.cpu().numpy())
dx = torch.from_numpy(np.random.default_rng(1).standard_normal((B, tpl.n))).cuda(); dy = torch.zeros_like(y)
out = {}
for ni in ("1", "2", "3"):
    os.environ["CE_SA_LSQR_NI"] = ni
    for rule in (None, (1e-13, 1e-13, 20000)):
        dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy, path="const_a", q_eval=q_t, lsqr=rule)
        out[(ni, rule is not None)] = (dA.cpu().numpy().copy(), dq.cpu().numpy().copy(), eng.last_lsqr_iters.cpu().numpy().copy(), adj.cpu().numpy().copy())
for tight in (False, True):
    r = out[("1", tight)]
    print("tight" if tight else "default", "iters ni1", r[2], "adj", r[3])
    for ni in ("2", "3"):
        o = out[(ni, tight)]
        print("  ni", ni, "iters", o[2], "adj", o[3], "max|dA diff| per instance", np.abs(o[0] - r[0]).max(axis=0), "max|dq diff|", np.abs(o[1] - r[1]).max(axis=0), "scale", np.abs(r[0]).max(), np.abs(r[1]).max())
