import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer
cfg = P.CONFIGS["M"]; n, cones, B = cfg["n"], cfg["cones"], 4096
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-4, "max_iters": 10000})
A_t = torch.from_numpy(A_eval).to(dev).requires_grad_()
q_t = torch.from_numpy(q_eval).to(dev).requires_grad_()
def step():
    A_t.grad = None; q_t.grad = None
    p, d, info, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, {}, True, None)
    p.sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("step ms", (time.perf_counter() - t0) / 20 * 1e3)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=14, max_name_column_width=60))
st = torch.cuda.memory_stats()
print({k: st[k] for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "segment.all.current", "reserved_bytes.all.current", "allocated_bytes.all.peak")})
print(os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"), os.environ.get("PYTORCH_NO_HIP_MEMORY_CACHING"))
