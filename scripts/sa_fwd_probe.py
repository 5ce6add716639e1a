"""k_sa_fwd timing probe on the BASELINE config 4 shape: fixed iteration counts (eps = 0), kernel time via the engine's HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import ConeEngine, make_settings
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A, b, c, cones, tpl = P.sdp_c4_batch(B, seed=0)
Ab = np.broadcast_to(A, (B,) + A.shape).copy()
A_eval, q_eval = tpl.values_from_dense(Ab, b, c)
dev = torch.device("cuda", 0)
eng = ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, dev)
A_bm = torch.from_numpy(A_eval).to(dev).t().contiguous(); q_t = torch.from_numpy(q_eval).to(dev)
eng.set_profiling(True)
for mi in (1, 26, 101):
    st = make_settings(dict(eps=0.0, eps_infeas=0.0, max_iters=mi))
    eng.solve(A_bm, q_t, st); torch.cuda.synchronize(); eng.reset_profile()
    for _ in range(3): out = eng.solve(A_bm, q_t, st)
    torch.cuda.synchronize()
    print("max_iters", mi, "kernel %.3f ms" % eng.profile(0)[0], eng.last_const_a_kernel, "B", B)
