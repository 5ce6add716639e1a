"""Per-iteration cost of k_fwd2 at fixed iteration counts (eps = 0), with and without acceleration: the A/B harness of the timing probes
(CE_ENGINE_SO selects the build).  usage: iter_ab_probe.py [config] [B]"""
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
eng.set_profiling(True)
def run(reps=3, **kw):
    st = make_settings(kw)
    eng.solve(A_bm, q_t, st); torch.cuda.synchronize(); eng.reset_profile()
    for _ in range(reps): out = eng.solve(A_bm, q_t, st)
    torch.cuda.synchronize()
    return eng.profile(0)[0], out
for aa in (0, 10):      # (adaptive_scale off: a rescale is a refactorisation, and a probe build with wrong iterates rescales differently)
    t0, _ = run(max_iters=51, eps=0.0, eps_infeas=0.0, acceleration_lookback=aa, adaptive_scale=0)
    t1, _ = run(max_iters=101, eps=0.0, eps_infeas=0.0, acceleration_lookback=aa, adaptive_scale=0)
    t2, _ = run(max_iters=201, eps=0.0, eps_infeas=0.0, acceleration_lookback=aa, adaptive_scale=0)
    print(f"lookback {aa:2d}: 51 it {t0:.3f} ms, 101 it {t1:.3f} ms, 201 it {t2:.3f} ms, per 100 iterations {2 * (t1 - t0):.3f} / {t2 - t1:.3f} ms")
t, out = run(eps=1e-4, max_iters=10000)
print(f"eps 1e-4: {t:.3f} ms, mean iterations {out[3].float().mean().item():.1f}, solved {(out[4] == 1).float().mean().item():.3f}")
