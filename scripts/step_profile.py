"""Where the host time of one bench step goes (the GPU idles between the forward's status read and the adjoint's launch): cProfile over the bench's own step,
   plus wall-clock stamps  apply() | sum() | backward()  per step.   usage: python scripts/step_profile.py [steps]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 4096
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-4, "max_iters": 10000, "acceleration_lookback": 10})
A_t = torch.from_numpy(A_eval).to(dev).requires_grad_(); q_t = torch.from_numpy(q_eval).to(dev).requires_grad_()
T = np.zeros((steps, 4))
def step(i):
    A_t.grad = None; q_t.grad = None
    t0 = time.perf_counter()
    primal, dual, info, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, {}, True, None)
    t1 = time.perf_counter()
    loss = primal.sum()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    T[i] = (t0, t1, t2, t3)
import warnings; warnings.simplefilter("ignore")
for i in range(20): step(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
d = np.diff(T, axis=1) * 1e6
print("unprofiled: %.4f ms/step;  host us per step (median):  apply() %.1f | sum() %.1f | backward() %.1f | to next apply %.1f" %
      (dt / steps * 1e3, np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median((T[1:, 0] - T[:-1, 3]) * 1e6)))
pr = cProfile.Profile(); pr.enable()
for i in range(steps): step(i)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(28)
print("\n".join(l[:200] for l in s.getvalue().split("\n")[:50]))
