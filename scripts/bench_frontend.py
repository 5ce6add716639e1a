"""Whole-layer forward + backward through cvxpylayers_amd.torch.CvxpyLayer on the metric configuration: parameters A (m, n),
b (m,), c (n,) all batched (B, ...), parameter maps evaluated on the device (ce_parammap_apply), solve, adjoint, gradients back to
the parameters.  Reports ms/step, problems/s and the HIP-event time of the param-map kernels."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.torch import CanonTemplate, CvxpyLayer, VariableRecovery

cfg = P.CONFIGS["M"]; n, cones = cfg["n"], cfg["cones"]; B = 4096
tpl = P.dense_template(n, cones); m = tpl.m
# parameters in user order (A, b, c); canonical vector p = [vec_F(A) (m n), b (m), c (n), 1]
nA = m * n
ptot = nA + m + n
cols = np.repeat(np.arange(n + 1), np.diff(tpl.indptr))
rows_l, cols_l, vals_l = [], [], []
for k in range(tpl.nnz_aug):
    i, j = int(tpl.indices[k]), int(cols[k])
    if j < n:
        rows_l.append(k); cols_l.append(j * m + i); vals_l.append(-1.0)        # A_cvx = -A ; Fortran flattening: index j*m + i
    else:
        rows_l.append(k); cols_l.append(nA + i); vals_l.append(1.0)
A_map = sp.csr_array(sp.coo_array((vals_l, (rows_l, cols_l)), shape=(tpl.nnz_aug, ptot + 1)))
q_map = sp.csr_array(sp.coo_array((np.ones(n), (np.arange(n), nA + m + np.arange(n))), shape=(n + 1, ptot + 1)))
template = CanonTemplate([(m, n), (m,), (n,)], [0, nA, nA + m], A_map, q_map, tpl.problem_data_index, cones,
                         [VariableRecovery(slice(0, n), None, (n,))])
layer = CvxpyLayer(template=template, solver_args={"eps": 1e-4, "max_iters": 10000})
A, b, c = P.generate(n, cones, B, seed=0)
dev = torch.device("cuda", 0)
At = torch.from_numpy(A).to(dev).requires_grad_(); bt = torch.from_numpy(b).to(dev).requires_grad_(); ct = torch.from_numpy(c).to(dev).requires_grad_()

def step():
    At.grad = None; bt.grad = None; ct.grad = None
    (x,) = layer(At, bt, ct)
    x.sum().backward()
    return x
x = step(); torch.cuda.synchronize()
if "--check" in sys.argv:                                  # against the oracle (not under a profiler)
    from oracle import oracle
    ref = oracle.solve_batch(A[:16], b[:16], c[:16], cones, eps=1e-4, max_iters=10000)
    print("max |x - x_oracle| on 16 instances:", float(np.abs(x.detach().cpu().numpy()[:16] - ref["x"]).max()))
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
# forward only / pieces
def timed(fn, reps=20):
    for _ in range(2): fn()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b_.record(); torch.cuda.synchronize()
    return a.elapsed_time(b_) / reps
def fwd_only():
    with torch.no_grad(): layer(At, bt, ct)
t_fwd = timed(fwd_only)
t_flat = timed(lambda: layer._flatten_params((At.detach(), bt.detach(), ct.detach()), (B,)))
# param-map kernel alone
from cvxpylayers_amd.torch.cvxpylayer import _spmm_bm
p_bm = layer._flatten_params((At.detach(), bt.detach(), ct.detach()), (B,))
fwd, bwd = layer._A.on(dev)
for _ in range(3): out = _spmm_bm(fwd, p_bm)
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record()
for _ in range(20): out = _spmm_bm(fwd, p_bm)
e1.record()
for _ in range(20): g = _spmm_bm(bwd, out)
e2.record(); torch.cuda.synchronize()
f_ms, b_ms = e0.elapsed_time(e1) / 20, e1.elapsed_time(e2) / 20
byt = (p_bm.numel() + out.numel()) * 8
res = dict(fwd_only_ms=t_fwd, flatten_ms=t_flat, workload="whole layer (A, b, c batched parameters -> x), metric config, B=4096", ms_per_step=dt * 1e3, problems_per_s=B / dt,
           parammap_fwd_ms=f_ms, parammap_bwd_ms=b_ms, parammap_algorithmic_GBps=byt / (f_ms * 1e-3) / 1e9, parammap_frac_of_8TBps=byt / (f_ms * 1e-3) / 8e12)
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True); json.dump(res, open("gpurun_out/frontend_bench.json", "w"), indent=1)
