"""Shared-A paths side by side on BASELINE configs 4 / 5 (plugin boundary): forward and backward time of every implementation
(persistent kernels k_sa_fwd / k_sa_lsqr with their thread-count / product variants, batch-GEMM + batched-LSQR torch path) and the
difference of their solutions and gradients to the first variant.   python scripts/shared_a_probe.py C5 16384 [variants...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer

dev = torch.device("cuda", 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
want = sys.argv[3:]
if cfg == "C5":
    A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=0)
    A1, _ = tpl.values_from_dense(A[None], b[None], c[:1])
    A_dev = torch.from_numpy(A1).to(dev).expand(-1, B).t().contiguous().t()
    q_eval = np.concatenate([c.T, np.zeros((1, B))], axis=0)
else:
    A, b, c, cones, tpl = P.sdp_c4_batch(B, seed=0)
    A1, _ = tpl.values_from_dense(A[None], b[:1], c[:1])
    nnzA = int(tpl.indptr[tpl.n])
    A_dev = torch.from_numpy(A1).to(dev).expand(-1, B).t().contiguous()
    A_dev[:, nnzA:] = torch.from_numpy(b[:, tpl.indices[nnzA:]]).to(dev)          # b is per instance
    A_dev = A_dev.t()
    q_eval = np.concatenate([c.T, np.zeros((1, B))], axis=0)

VARIANTS = {
    "torch":      dict(CE_SA_FWD="0", CE_SA_KERNEL="0"),
    "k256":       dict(CE_SA_FWD="1", CE_SA_KERNEL="1", CE_SA_NT="256"),
    "k512":       dict(CE_SA_FWD="1", CE_SA_KERNEL="1", CE_SA_NT="512"),
    "k1024":      dict(CE_SA_FWD="1", CE_SA_KERNEL="1", CE_SA_NT="1024"),
    "kcsr":       dict(CE_SA_FWD="1", CE_SA_KERNEL="1", CE_SA_SPLIT="0"),
    "default":    dict(),
    # the adjoint kernel: one instance per workgroup (k_sa_lsqr) against NI instances sharing the stream over A_d^T (k_sa_lsqr_mi)
    "ni1":        dict(CE_SA_LSQR_NI="1"),
    "spec0":      dict(CE_SA_LSQR_SPEC="0"),
    "res2":       dict(CE_SA_LSQR_PADLDS="70"),
    "res1":       dict(CE_SA_LSQR_PADLDS="120"),
    "ni2":        dict(CE_SA_LSQR_NI="2"),
    "ni3":        dict(CE_SA_LSQR_NI="3"),
    # the adjoint's LSQR stopping rule: rounds 1-4 (atol = btol = 1e-12, 4 (n + m) iterations) against diffcp's (1e-8, 1e-8, 2 (n + m + 1): the default since round 5)
    "tight":      dict(_args=dict(lsqr_atol=1e-12, lsqr_btol=1e-12, lsqr_iter_lim=4 * (tpl.n + tpl.m))),
}
KEYS = ("CE_SA_FWD", "CE_SA_KERNEL", "CE_SA_NT", "CE_SA_SPLIT", "CE_SA_LSQR_NI", "CE_SA_LSQR_SPEC", "CE_SA_LSQR_PADLDS")
ref = None
res = []
wts = torch.from_numpy(np.random.default_rng(5).standard_normal((tpl.n, B))).to(dev) if True else None
for name in (want or ["torch", "k512", "k256", "default"]):
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update({k: v for k, v in VARIANTS[name].items() if k != "_args"})
    ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={"eps": 1e-4, "max_iters": 20000, "raise_on_error": False, **VARIANTS[name].get("_args", {})})
    A_t = A_dev.detach().requires_grad_(); q_t = torch.from_numpy(q_eval).to(dev).requires_grad_()
    tf = tb = 0.0; reps = 2
    for rep in range(reps + 1):
        A_t.grad = None; q_t.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p, d, info, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, {}, True, None)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (p * (wts if p.shape == wts.shape else wts.t())).sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if rep: tf += (t1 - t0) / reps; tb += (t2 - t1) / reps
    eng = ctx.engine(dev)
    out = dict(variant=name, cfg=cfg, B=B, fwd_ms=tf * 1e3, bwd_ms=tb * 1e3, iters_mean=float(info["iters"].float().mean()), solved=float((info["status"] == 1).float().mean()),
               fwd_kernel=getattr(eng, "last_const_a_kernel", None), lsqr_iters=(float(eng.last_lsqr_iters.float().mean()) if getattr(eng, "last_lsqr_iters", None) is not None else None),
               lsqr_iters_max=(int(eng.last_lsqr_iters.max()) if getattr(eng, "last_lsqr_iters", None) is not None else None))
    x = p.detach(); g = q_t.grad.detach(); gA = A_t.grad.detach()
    out["gq_max"] = float(g.abs().max()); out["gA_max"] = float(gA.abs().max())
    if ref is None: ref = (x.clone(), g.clone(), gA.clone())
    else:
        out["dx_max"] = float((x - ref[0]).abs().max()); out["dgq_rel"] = float((g - ref[1]).abs().max() / (1e-300 + ref[1].abs().max())); out["dgA_rel"] = float((gA - ref[2]).abs().max() / (1e-300 + ref[2].abs().max()))
    if getattr(eng, "last_lsqr_iters", None) is not None:
        li = eng.last_lsqr_iters.double().cpu().numpy()
        out["lsqr_iters_pct"] = [float(v) for v in np.percentile(li, [5, 25, 50, 75, 95])]
        for g_ in (2, 3, 4):
            k_ = (len(li) // g_) * g_
            out[f"lsqr_group{g_}_max_over_mean"] = float(li[:k_].reshape(-1, g_).max(axis=1).mean() / li.mean())
        if name.startswith("ni"):
            if "li_ref" in globals(): out["lsqr_iters_maxdiff"] = float(np.abs(li - li_ref).max())
            else: li_ref = li
    eng.last_lsqr_iters = None
    print(json.dumps(out), flush=True); res.append(out)
os.makedirs("gpurun_out/sap", exist_ok=True)
json.dump(res, open(f"gpurun_out/sap/{cfg}_{B}.json", "w"), indent=1)
