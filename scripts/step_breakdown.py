"""Where does the wall time of one bench step go?  (host-side view: kernel times from the engine's HIP events vs wall)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 4096
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-4, "max_iters": 10000})
eng = ctx.engine(dev)
for layout in ("batch-minor (reference)", "batch-major view (frontend-native)"):
    if layout.startswith("batch-minor"):
        A_t = torch.from_numpy(A_eval).to(dev).requires_grad_()
    else:
        A_t = torch.from_numpy(A_eval).to(dev).t().contiguous().t().requires_grad_()
    q_t = torch.from_numpy(q_eval).to(dev).requires_grad_()
    def fwd():
        return _CvxpyLayer.apply(None, q_t, A_t, ctx, {}, True, None)
    def step():
        A_t.grad = None; q_t.grad = None
        p, d, info, _ = fwd()
        p.sum().backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    eng.set_profiling(True); eng.reset_profile()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    f, b_, l = eng.profile(0)[0], eng.profile(1)[0], eng.profile(2)
    eng.set_profiling(False)
    t0 = time.perf_counter()
    for _ in range(20):
        with torch.no_grad():
            fwd()
    torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{layout:38s} step {dt:.3f} ms | fwd kernel {f:.3f} bwd kernel {b_:.3f} layout {l[0]:.3f} x{l[1] // 20} | forward-only call {dtf:.3f} ms | unaccounted {dt - f - b_ - l[0] * (l[1] // 20):.3f} ms")
