"""Ablation probe: times k_forward / k_backward (HIP events inside the engine) under different settings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings

cfgname = sys.argv[1] if len(sys.argv) > 1 else "M"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = P.CONFIGS[cfgname]; n, cones = cfg["n"], cfg["cones"]
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous()
q_t = torch.from_numpy(q_eval).to(dev)
print(eng.launch_info())
eng.set_profiling(True)
def run(tag, reps=3, **kw):
    st = make_settings(kw)
    eng.solve(A_bm, q_t, st); torch.cuda.synchronize(); eng.reset_profile()
    for _ in range(reps): out = eng.solve(A_bm, q_t, st)
    torch.cuda.synchronize()
    ms, nl = eng.profile(0)
    it = out[3].float().mean().item()
    print(f"{tag:40s} fwd {ms:8.3f} ms  mean iters {it:7.1f}  status1 {(out[4]==1).float().mean().item():.3f}")
    return out
run("max_iters=1 normalize=0", max_iters=1, normalize=0)
run("max_iters=1 normalize=1", max_iters=1, normalize=1)
run("max_iters=26", max_iters=26)
run("max_iters=51", max_iters=51)
run("max_iters=101 eps=0", max_iters=101, eps=0.0, eps_infeas=0.0)
run("max_iters=201 eps=0", max_iters=201, eps=0.0, eps_infeas=0.0)
out = run("eps=1e-4", eps=1e-4, max_iters=10000)
x, y, s = out[0], out[1], out[2]
dx = torch.ones_like(x); dy = torch.zeros_like(y)
eng.vjp(A_bm, x, y, s, dx, dy); torch.cuda.synchronize(); eng.reset_profile()
for _ in range(3): eng.vjp(A_bm, x, y, s, dx, dy)
torch.cuda.synchronize()
print("bwd %.3f ms" % eng.profile(1)[0])
