"""Debug-build probe (-DCE_TIMING): mean per-phase shader cycles of the backward kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
cfg = P.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "M"]; n, cones = cfg["n"], cfg["cones"]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096      # (B = 256: one workgroup per CU, the latency of one instance)
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-4, max_iters=10000)))
torch.cuda.synchronize()
ts = s[:, :32].cpu().numpy()
print("forward phases (cycles, mean over instances); iters mean", it.float().mean().item())
for nm, a, b2 in (("init+load b,c", 0, 1), ("equilibration", 1, 2), ("  equilibration: FP32 tile gathers (first touch of A)", 1, 14), ("  equilibration: 26 passes", 14, 15), ("  equilibration: D, E out, b / c scaling, sigma", 15, 2), ("refactor: materialize+S", 7, 8), ("refactor: GJ", 8, 9), ("refactor: g,phi", 9, 10),
                  ("  g,phi: zero, Dy b, gather A^T tile, A^T(Dy b)", 9, 11), ("  g,phi: G (c -+ a)", 11, 12), ("  g,phi: gather A tile, A gx, A gk", 12, 13), ("  g,phi: h.g reduce, phi tile, zero", 13, 10),
                  ("first refactor total (2->3 includes)", 2, 3), ("iterations (incl. later refactors)", 3, 4), ("writeback", 4, 5), ("total", 0, 5)):
    print(f"  {nm:38s} {(ts[:, b2] - ts[:, a]).mean():12.1f}")
itn = it.float().cpu().numpy()
acc = ts[:, 16:24]
if acc.sum() > 0:
    print("iteration phases of k_fwd2 (cycles per iteration, wave 0's clock; fast-path iterations only are complete):")
    for k, nm in enumerate(("acceleration step / safeguard / renormalisation (amortised)", "P1a  A^T w_y up to its barrier", "the three barriers", "P1b  G t up to its barrier", "fused: A p_x, tau, cone input", "fused: projection, update",
                            "loop bookkeeping, thread coordinates", "slow path of the check iterations (amortised)")):
        print(f"  {nm:62s} {(acc[:, k] / itn).mean():10.1f}")
    print(f"  {'sum':62s} {(acc[:, :8].sum(1) / itn).mean():10.1f}")
ea = ts[:, 24:32]
if ea.sum() > 0:
    print("equilibration pass (cycles per pass, 26 passes):   " + "  ".join(f"{nm} {ea[:, k].mean() / 26:7.1f}" for k, nm in enumerate(("norms+column factor", "block sums+row factor", "barrier", "factor reads+scaling"))))
    print("Gauss-Jordan block (cycles per block, 13 blocks):  " + "  ".join(f"{nm} {ea[:, 4 + k].mean() / 13:7.1f}" for k, nm in enumerate(("pivot block+inverse", "multipliers", "rank-4 update", "publish+barrier"))))
x, y, s, it, st, res = eng.solve(A_bm, q_t, make_settings(dict(eps=1e-4, max_iters=10000)))
dx = torch.ones_like(x); dy = torch.zeros_like(y)
dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy)
torch.cuda.synchronize()
dA, dq, adj = eng.vjp(A_bm, x, y, s, dx, dy)          # (the second call runs the two-tile plan: the first has no history)
torch.cuda.synchronize()
t = dA.t()[:, :19].cpu().numpy()
names = ["load", "classify+number", "dv,ay,as,fvec", "assemble", "ptol+GJ", "solve+q+ry", "output", "NK"]
for k, nm in enumerate(names):
    print(f"{nm:18s} mean {t[:, k].mean():12.1f}  max {t[:, k].max():12.1f}")
print("assemble: H part", t[:, 8].mean())
print("sum of phases", t[:, :7].sum(1).mean())
for w0, nm in ((9, "wave 0"), (14, "wave 3")):
    print(f"elimination, cycles per pivot ({nm}): " + "  ".join(f"{lab} {(t[:, w0 + k] / t[:, 7]).mean():7.1f}" for k, lab in enumerate(("search+publish", "barrier", "reads+reciprocal", "pivot row bpermute", "update"))))
