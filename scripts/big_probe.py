"""Per-instance-A templates beyond the register / LDS-resident kernels (size-generic kernels, A and G in global memory):
forward / backward time and parity with the oracle.   python scripts/big_probe.py n m_l nsoc soc_dim B"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
n, ml, nsoc, sd, B = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (200, 100, 10, 21, 512)))
cones = {"z": 0, "l": ml, "q": [sd] * nsoc}
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
print(dict(n=n, m=tpl.m, B=B), eng.launch_info())
eng.set_profiling(True)
if "--breakdown" in sys.argv:
    for tag, kw in (("max_iters=1 normalize=0", dict(max_iters=1, normalize=0)), ("max_iters=1", dict(max_iters=1)), ("max_iters=26 eps=0", dict(max_iters=26, eps=0.0, eps_infeas=0.0)),
                    ("max_iters=101 eps=0", dict(max_iters=101, eps=0.0, eps_infeas=0.0))):
        stb = make_settings(kw)
        eng.solve(A_bm, q_t, stb); torch.cuda.synchronize(); eng.reset_profile()
        for _ in range(2): eng.solve(A_bm, q_t, stb)
        torch.cuda.synchronize()
        print("  %-26s fwd %.2f ms" % (tag, eng.profile(0)[0]))
st = make_settings(dict(eps=1e-4, max_iters=10000))
out = eng.solve(A_bm, q_t, st); torch.cuda.synchronize(); eng.reset_profile()
for _ in range(2): out = eng.solve(A_bm, q_t, st)
torch.cuda.synchronize()
print("fwd %.2f ms  iters %.1f  solved %.3f" % (eng.profile(0)[0], out[3].float().mean().item(), (out[4] == 1).float().mean().item()))
x, y, s = out[0], out[1], out[2]
dx = torch.ones_like(x); dy = torch.zeros_like(y)
eng.vjp(A_bm, x, y, s, dx, dy); torch.cuda.synchronize(); eng.reset_profile()
for _ in range(2): g = eng.vjp(A_bm, x, y, s, dx, dy)
torch.cuda.synchronize()
print("bwd %.2f ms" % eng.profile(1)[0])
if "--check" in sys.argv:
    from oracle import oracle
    k = min(B, 8)
    ref = oracle.solve_batch(A[:k], b[:k], c[:k], cones, eps=1e-4, max_iters=10000)
    print("max |x - x_oracle| (first %d):" % k, float(np.abs(x[:k].cpu().numpy() - ref["x"]).max()), "oracle iters", ref["iters"][:k])
