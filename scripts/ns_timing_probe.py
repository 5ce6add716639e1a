"""Debug-build probe (-DCE_TIMING; CE_ENGINE_SO=.../libcone_engine_timing.so): mean per-phase shader cycles of the search-free adjoint kernel k_backward_ns.
usage: ns_timing_probe.py [config] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
cfg = P.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "M"]; n, cones = cfg["n"], cfg["cones"]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-4, max_iters=10000)))
# (the timing build of the FORWARD kernel writes its own stamps over s[:, :32]: restore those slack entries from s = b - A x, or the adjoint would be probed on a distorted active set)
s_fix = torch.from_numpy(b - np.einsum("bij,bj->bi", A, x.cpu().numpy())).to(dev)
s[:, :32] = torch.where(s_fix[:, :32] > 0, s_fix[:, :32], torch.zeros_like(s_fix[:, :32])) if False else s_fix[:, :32]
dx = torch.ones_like(x); dy = torch.zeros_like(y)
for _ in range(2):
    dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy, path="per_instance", q_eval=q_t)
torch.cuda.synchronize()
t = dA.t()[:, :20].cpu().numpy()
names = ["load", "classify + d (16 lanes per cone)", "a_z, numbering, f, lists", "row elimination of B (registers, one barrier per pivot)", "null-space transform of the rows", "reduced Hessian on the matrix cores", "sweep", "x, q, g, mu", "r_y", "outputs"]
for k, nm in enumerate(names):
    print(f"{nm:40s} mean {t[:, k].mean():12.1f}  max {t[:, k].max():12.1f}")
print(f"{'sum of phases':40s} {t[:, :10].sum(1).mean():12.1f}")
print("mean NK = n + neq %.1f, mean nf %.1f (sweep blocks %.1f)" % (t[:, 10].mean(), t[:, 11].mean(), np.ceil(t[:, 11] / 4).mean()))
print("reduced Hessian, thread 0: set-up (columns, masks, Z^T f requests) %.0f, the loop %.0f (%.1f list rows = %.1f steps of four), epilogue %.0f" % (t[:, 12].mean(), t[:, 13].mean(), t[:, 15].mean(), np.ceil((t[:, 15] + 1) / 4).mean(), t[:, 14].mean()))
v = (y - s); l_ = cones.get("l", 0); z_ = cones.get("z", 0)
neq_t = torch.full((B,), float(z_), dtype=torch.float64, device=dev) + (v[:, z_:z_ + l_] > 0).sum(dim=1)
off = z_ + l_
for d in cones.get("q", []):
    t0, nz = v[:, off], v[:, off + 1:off + d].norm(dim=1)
    inside = nz <= t0; bnd = (~inside) & ~(nz <= -t0)
    neq_t = neq_t + inside * float(d) + bnd * 1.0; off += d
print("equality rows counted from y - s on the host side: mean %.2f (the kernel's n + neq - n = %.2f)" % (float(neq_t.mean()), t[:, 10].mean() - n))
neq_k = np.maximum(t[:, 10] - n, 1)
print("single-wave row elimination, cycles per pivot (instances that took it): search %.0f, reciprocal + scaled row %.0f, row updates %.0f, loop top %.0f" % tuple((t[:, 16 + k][t[:, 16] > 0] / neq_k[t[:, 16] > 0]).mean() for k in range(4)))
