"""Forward time of the metric configuration cold vs warm-started from the solution of slightly different data (a training loop's
consecutive steps): HIP-event kernel time and mean iterations."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 4096
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
def dev_inputs(A, b, c):
    A_eval, q_eval = tpl.values_from_dense(A, b, c)
    return torch.from_numpy(A_eval).to(dev).t().contiguous(), torch.from_numpy(q_eval).to(dev)
A0, q0 = dev_inputs(A, b, c)
st = make_settings(dict(eps=1e-4, max_iters=10000))
base = eng.solve(A0, q0, st); torch.cuda.synchronize()
eng.set_profiling(True)
rng = np.random.default_rng(1)
rows = []
for rel in (1e-4, 1e-3, 1e-2, 1e-1):
    A2 = A * (1 + rel * rng.standard_normal(A.shape)); b2 = b * (1 + rel * rng.standard_normal(b.shape)); c2 = c * (1 + rel * rng.standard_normal(c.shape))
    A2d, q2d = dev_inputs(A2, b2, c2)
    res = {}
    for tag, warm in (("cold", None), ("warm", (base[0], base[1], base[2]))):
        eng.solve(A2d, q2d, st, warm=warm); torch.cuda.synchronize(); eng.reset_profile()
        for _ in range(5): out = eng.solve(A2d, q2d, st, warm=warm)
        torch.cuda.synchronize()
        res[tag] = dict(ms=eng.profile(0)[0], iters=out[3].float().mean().item(), solved=(out[4] == 1).float().mean().item())
    rows.append(dict(relative_change=rel, **{f"{k}_{kk}": vv for k, v in res.items() for kk, vv in v.items()}))
    print(json.dumps(rows[-1]))
os.makedirs("gpurun_out", exist_ok=True); json.dump(rows, open("gpurun_out/warm_probe.json", "w"), indent=1)
