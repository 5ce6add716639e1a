"""Kernel times (HIP events) of the native quadratic-objective path on BASELINE config 2 (native box QP, n=50, m=100)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
nx, B = 50, 4096
rng = np.random.default_rng(0)
F = rng.standard_normal((nx, nx)) / np.sqrt(nx); g = rng.standard_normal((B, nx))
lo = -0.5 - 0.5 * rng.random((B, nx)); hi = 0.5 + 0.5 * rng.random((B, nx))
An = np.concatenate([-np.eye(nx), np.eye(nx)], axis=0)
rows, ptr = [], [0]
for j in range(nx):
    rows.extend(range(j + 1)); ptr.append(len(rows))
pidx, pptr = np.asarray(rows, dtype=np.int32), np.asarray(ptr, dtype=np.int32)
Pm = 2 * F.T @ F
pv = Pm[pidx, np.repeat(np.arange(nx), np.diff(pptr))]
cones = {"z": 0, "l": 100, "q": [], "s": []}
tpl = P.dense_template(nx, cones, pattern=(An != 0))
A_eval, q_eval = tpl.values_from_dense(np.broadcast_to(An, (B,) + An.shape).copy(), np.concatenate([-lo, hi], axis=1), -2 * g @ F)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev, p_structure=(pidx, pptr))
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
P_bm = torch.from_numpy(np.broadcast_to(pv, (B, len(pv))).copy()).to(dev)
print(eng.launch_info(), "qp_native", eng.qp_native)
eng.set_profiling(True)
for tag, kw in (("max_iters=1", dict(max_iters=1)), ("max_iters=26", dict(max_iters=26)), ("max_iters=101 eps=0", dict(max_iters=101, eps=0.0, eps_infeas=0.0)),
                ("max_iters=201 eps=0", dict(max_iters=201, eps=0.0, eps_infeas=0.0)), ("eps=1e-4", dict(eps=1e-4, max_iters=20000))):
    st = make_settings(kw)
    eng.solve(A_bm, q_t, st, P_bm=P_bm); torch.cuda.synchronize(); eng.reset_profile()
    for _ in range(5): out = eng.solve(A_bm, q_t, st, P_bm=P_bm)
    torch.cuda.synchronize()
    print(f"{tag:24s} fwd {eng.profile(0)[0]:8.3f} ms  iters {out[3].float().mean().item():7.1f}")
x, y, s = out[0], out[1], out[2]
dx = torch.ones_like(x); dy = torch.zeros_like(y)
eng.vjp(A_bm, x, y, s, dx, dy, P_bm=P_bm); torch.cuda.synchronize(); eng.reset_profile()
for _ in range(5): r = eng.vjp(A_bm, x, y, s, dx, dy, P_bm=P_bm)
torch.cuda.synchronize()
print("bwd %.3f ms, flagged %d" % (eng.profile(1)[0], int((r[2] != 0).sum())))
