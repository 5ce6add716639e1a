"""BASELINE config 4 shape: SDP with one 20x20 PSD cone, x = svec(X) (n = 210), 20 equality rows <A_k, X> = b_k, PSD block -I
(m = 230); C (= c) and b batched, A shared -> constant-A path (PSD projection by batched eigendecompositions)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
from oracle import oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
k, neq = 20, 20
d = k * (k + 1) // 2
n = d; cones = {"z": neq, "l": 0, "q": [], "s": [k]}; m = neq + d
rng = np.random.default_rng(0)
A = np.zeros((m, n)); A[:neq] = rng.standard_normal((neq, n)) / np.sqrt(n); A[neq:] = -np.eye(d)
x0 = P.sym_to_svec(np.stack([(lambda G: G @ G.T / k + 0.1 * np.eye(k))(rng.standard_normal((k, k))) for _ in range(B)]))
s0 = np.concatenate([np.zeros((B, neq)), x0], axis=1)           # s = b - A x : PSD block s = x0 (A_psd = -I, b_psd = 0) 
yz = rng.standard_normal((B, neq))
ypsd = P.sym_to_svec(np.stack([(lambda G: G @ G.T / k + 0.1 * np.eye(k))(rng.standard_normal((k, k))) for _ in range(B)]))
y0 = np.concatenate([yz, ypsd], axis=1)
b = x0 @ A.T + s0
c = -(y0 @ A)
tpl = P.dense_template(n, cones, pattern=(A != 0), b_pattern=np.ones(m, bool))
Ab = np.broadcast_to(A, (B, m, n)).copy()
A_eval, q_eval = tpl.values_from_dense(Ab, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
aa = int(sys.argv[3]) if len(sys.argv) > 3 else 1
st = make_settings(dict(eps=eps, max_iters=20000, acceleration_lookback=aa))      # (the oracle below runs the same: one-pair acceleration or plain)
eng.solve(A_bm, q_t, st); torch.cuda.synchronize()      # warm-up: rocBLAS / rocSOLVER initialisation, kernel loading
t0 = time.perf_counter(); x, y, s, it, status, res = eng.solve(A_bm, q_t, st); torch.cuda.synchronize(); t1 = time.perf_counter()
print("path", eng.last_path, getattr(eng, "last_const_a_kernel", None), "B", B, "eps", eps, "fwd %.1f ms  iters mean %.0f max %d  solved %.3f" % ((t1 - t0) * 1e3, it.float().mean().item(), int(it.max()), (status == 1).float().mean().item()))
dx = torch.ones_like(x); dy = torch.zeros_like(y)
eng.vjp(A_bm, x, y, s, dx, dy); torch.cuda.synchronize()
t0 = time.perf_counter(); dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy); torch.cuda.synchronize(); t1 = time.perf_counter()
print("bwd %.1f ms, LSQR not converged for %d" % ((t1 - t0) * 1e3, int((adj != 0).sum())), "LSQR iterations mean", float(getattr(eng, "last_lsqr_iters", torch.zeros(1)).float().mean()))
nb = min(B, 8)
t0 = time.perf_counter(); ref = oracle.solve_batch(Ab[:nb], b[:nb], c[:nb], cones, eps=eps, max_iters=20000, acceleration_lookback=aa); t1 = time.perf_counter()
print("oracle %d instances %.2f s (%d threads), iters %s" % (nb, t1 - t0, oracle.num_threads(), ref["iters"][:4]), "max |x - x_ref|", np.abs(x.cpu().numpy()[:nb] - ref["x"]).max())

del eng
import gc; gc.collect()
