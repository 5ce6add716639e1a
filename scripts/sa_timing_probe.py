"""Debug-build probe (make timing): shader cycles per phase of k_sa_fwd's iteration (thread 0 of every instance; mean per iteration).  usage: sa_timing_probe.py C5|C4 [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (2048 if name == "C5" else 1024)
dev = torch.device("cuda", 0)
if name == "C5":
    A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=0)
    A1, _ = tpl.values_from_dense(A[None], b[None], c[:1])
    A_eval = np.repeat(A1, B, axis=1); q_eval = np.concatenate([c.T, np.zeros((1, B))], axis=0)
else:
    A, b, c, cones, tpl = P.sdp_c4_batch(B, seed=0)
    A_eval, q_eval = tpl.values_from_dense(np.broadcast_to(A, (B,) + A.shape).copy(), b, c)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
look = int(os.environ.get("LOOK", "0"))
x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-4, max_iters=20000, acceleration_lookback=look)))
torch.cuda.synchronize()
print("path", eng.last_path, "iters mean", it.float().mean().item())
t = s[:, :12].cpu().numpy(); iters = it.cpu().numpy().astype(float)[:, None]
names = ["refresh (total)", "phi.w + t' gather", "pass 1 (A_d t') + K0 w_d", "A_d u sum", "z = K^-1 v", "pass 2 (A_d^T)", "elementwise ut / cone input", "cones (SOC / PSD / exp)", "project + update (fast)", "check iterations (all)"]
idx = [0, 1, 2, 3, 3, 4, 5, 6, 7, 8]
# stamps: 0 refresh, 1 gather, 2 pass1, 3 sum+Kinv (two phases under one stamp pair), 4 pass2, 5 elementwise, 6 cones, 7 update, 8 check tail
per_iter = t / iters
for k, nm in ((0, "refresh (per iteration share)"), (1, "phi.w + wyd + t' gather"), (2, "pass 1 (A_d t') + K0 w_d"), (3, "A_d u sum + z = K^-1 v"), (4, "pass 2 (A_d^T (w_d + z))"),
              (5, "elementwise ut / cone input"), (6, "cones (SOC / PSD / exp)"), (7, "project + update (fast path)"), (8, "check iterations (residual products etc., per iteration share)")):
    print(f"  {nm:62s} {per_iter[:, k].mean():10.1f} cycles / iteration")
print(f"  {'sum':62s} {per_iter[:, :9].sum(axis=1).mean():10.1f}")

# ---- adjoint (k_sa_lsqr): cycles per phase of one LSQR iteration
dx = torch.ones_like(x); dy = torch.zeros_like(y)
dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy, path="const_a")
torch.cuda.synchronize()
li = getattr(eng, "last_lsqr_iters", None)
li = li.cpu().numpy().astype(float)[:, None] if li is not None else np.ones((B, 1))
t = dA.t()[:, :8].cpu().numpy() / li
print("k_sa_lsqr: LSQR iterations mean", float(li.mean()))
for k, nm in ((0, "N v: both products (fused pass over A_d^T + row loop)"), (1, "cone derivative D Pi (A vx - vy) + u-hat"), (2, "|u-hat| (block reduce)"),
              (3, "u scaling + q = D Pi(u_y)"), (4, "N^T u: both products"), (5, "|v-hat| (block reduce)"), (6, "v, w, r updates"), (7, "scalar recurrences / loop top")):
    print(f"  {nm:62s} {t[:, k].mean():10.1f} cycles / LSQR iteration")
print(f"  {'sum':62s} {t.sum(axis=1).mean():10.1f}")

# ---- the multi-instance adjoint kernel (k_sa_lsqr_mi, CE_SA_LSQR_NI=2|3): stamps of the workgroup's first thread, per iteration of the workgroup (= the slowest of its instances)
ni = int(os.environ.get("CE_SA_LSQR_NI", "0") or 0)
if ni >= 2:
    t = dA.t()[::ni, :13].cpu().numpy()
    lim = li[: (B // ni) * ni].reshape(-1, ni).max(axis=1)[:, None]
    t = t[: lim.shape[0]] / lim
    print(f"k_sa_lsqr_mi NI={ni}: workgroup iterations mean", float(lim.mean()))
    for k, nm in enumerate(("A: dense entries of v_y", "A: streaming pass", "A: wait at the pass's barrier", "A: row loop", "D Pi (t_y) + u-hat_y", "|u-hat| (reduce)", "u scaling, q = D Pi(u_y), publish",
                            "E: dense entries of q", "E: streaming pass", "E: wait at the pass's barrier", "E: row loop", "|v-hat| (reduce)", "v, w, r updates, stop tests, live mask")):
        print(f"  {nm:62s} {t[:, k].mean():10.1f} cycles / iteration")
    print(f"  {'sum':62s} {t.sum(axis=1).mean():10.1f}")
