"""Adjoint A/B on one box: the pivoting elimination (k_backward_rt, path per_instance_dense) against the search-free kernel + LSQR re-solve (k_backward_ns + k_sa_lsqr<0>
over the flagged list; the plugin's default path), same solutions, same cotangents.  usage: bwd_ab_probe.py [config] [B] [eps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import _lib, problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings

cfgname = sys.argv[1] if len(sys.argv) > 1 else "M"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
cfg = P.CONFIGS[cfgname]; n, cones = cfg["n"], cfg["cones"]
tpl = P.dense_template(n, cones)
A, b, c = P.generate(n, cones, B, seed=0)
A_eval, q_eval = tpl.values_from_dense(A, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous()
q_t = torch.from_numpy(q_eval).to(dev)
print(eng.launch_info(), "ns variant", _lib.lib().ce_adjoint_ns_variant(eng._h))
x, y, s, iters, status, resid = eng.solve(A_bm, q_t, make_settings(dict(eps=eps, max_iters=10000)))
dx = torch.ones_like(x); dy = torch.zeros_like(y)
eng.set_profiling(True)
res = {}
for tag, kw in (("pivoting k_backward_rt", dict(path="per_instance_dense")), ("search-free k_backward_ns + re-solve", dict(path="per_instance", q_eval=q_t))):
    for _ in range(3): out = eng.vjp(A_bm, x, y, s, dx, dy, **kw)
    torch.cuda.synchronize(); eng.reset_profile()
    for _ in range(10): out = eng.vjp(A_bm, x, y, s, dx, dy, **kw)
    torch.cuda.synchronize()
    ms, nl = eng.profile(1)
    adj = out[2].cpu().numpy()
    res[tag] = out
    print(f"{tag:40s} {ms:8.4f} ms per call ({nl} bracketed launches)   adj_status counts {dict(zip(*np.unique(adj, return_counts=True)))}")
a, b_ = res["pivoting k_backward_rt"], res["search-free k_backward_ns + re-solve"]
reg = (a[2] == 0) & (b_[2] == 0)
dA0, dA1 = a[0].t()[reg], b_[0].t()[reg]
print("regular instances %d of %d: max rel difference of dA %.3e, median %.3e" % (int(reg.sum()), B, float(((dA0 - dA1).abs().amax(dim=1) / (1 + dA0.abs().amax(dim=1))).max()),
      float(((dA0 - dA1).abs().amax(dim=1) / (1 + dA0.abs().amax(dim=1))).median())))
