"""BASELINE config 5 shape (portfolio: min -mu^T w + gamma t  s.t. 1^T w = 1, w >= 0, ||F^T w|| <= t; n=501, m=552, A constant,
only mu batched) through the size-generic kernels: does it run, converge and agree with the oracle?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
from oracle import oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nw, kf = 500, 50
rng = np.random.default_rng(0)
F = rng.standard_normal((nw, kf)) / np.sqrt(kf) * 0.3
n = nw + 1; cones = {"z": 1, "l": nw, "q": [kf + 1]}; m = P.cone_rows(cones)
A = np.zeros((m, n)); b = np.zeros(m)
A[0, :nw] = 1.0; b[0] = 1.0                      # 1^T w = 1
A[1:1 + nw, :nw] = -np.eye(nw)                   # w >= 0 : s = w
A[1 + nw, nw] = -1.0                             # SOC: s0 = t
A[2 + nw:, :nw] = -F.T                           # s_{1..k} = F^T w
pattern = A != 0
tpl = P.dense_template(n, cones, pattern=pattern, b_pattern=(b != 0))
mu = 0.05 + 0.1 * rng.random((B, nw))
c = np.concatenate([-mu, 1.0 * np.ones((B, 1))], axis=1)
Ab = np.broadcast_to(A, (B, m, n)).copy(); bb = np.broadcast_to(b, (B, m)).copy()
A_eval, q_eval = tpl.values_from_dense(Ab, bb, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
print("launch info", eng.launch_info(), "nnz_aug", tpl.nnz_aug)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
st = make_settings(dict(eps=1e-6, max_iters=20000))
eng.set_profiling(True)
if B >= 1024: eng.solve(A_bm[:64].contiguous(), q_t[:, :64].contiguous(), make_settings(dict(eps=1e-6, max_iters=200))); torch.cuda.synchronize()   # warm-up
t0 = time.perf_counter(); x, y, s, it, status, res = eng.solve(A_bm, q_t, st); torch.cuda.synchronize(); t1 = time.perf_counter()
print("path", eng.last_path, "B", B)
print("fwd wall %.1f ms  kernel %.1f ms  iters mean %.0f  status ok %.2f" % ((t1 - t0) * 1e3, eng.profile(0)[0], it.float().mean().item(), (status == 1).float().mean().item()))
nb = min(B, 8)
t0 = time.perf_counter(); ref = oracle.solve_batch(Ab[:nb], bb[:nb], c[:nb], cones, eps=1e-6, max_iters=20000); t1 = time.perf_counter()
print("oracle %d instances %.2f s, iters %s" % (nb, t1 - t0, ref["iters"][:4]), "max |x - x_ref|", np.abs(x.cpu().numpy()[:nb] - ref["x"]).max())
dx = torch.ones_like(x); dy = torch.zeros_like(y)
try:
    t0 = time.perf_counter(); dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("bwd wall %.1f ms kernel %.1f ms adj flags %s" % ((t1 - t0) * 1e3, eng.profile(1)[0], adj.cpu().numpy()[:8]))
    torch.cuda.synchronize(); t0 = time.perf_counter(); dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy); torch.cuda.synchronize(); print("bwd (2nd call) wall %.1f ms, flagged %d" % ((time.perf_counter() - t0) * 1e3, int((adj != 0).sum())))
    g = oracle.adjoint_batch(Ab[:nb], bb[:nb], c[:nb], cones, x.cpu().numpy()[:nb], y.cpu().numpy()[:nb], s.cpu().numpy()[:nb], np.ones((nb, n)), np.zeros((nb, m)), mode="lsqr")
    print("max |dc - dc_ref| / scale", np.abs(dq.cpu().numpy()[:n, :nb].T - g["dc"]).max() / (1 + np.abs(g["dc"]).max()))
except Exception as e:
    print("bwd failed:", e)
