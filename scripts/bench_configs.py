"""Forward + backward throughput of every BASELINE.json configuration on one MI355X (plugin boundary, inputs resident in HBM,
frontend-native batch-major layout), one JSON line per configuration -> profiles/<round>/configs.json.
The headline metric stays bench.py (config M); this file backs the configuration table of DESIGN.md."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvxpylayers_amd import problems as P
from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer

dev = torch.device("cuda", 0)


def portfolio(B, nw=500, kf=50, seed=0):
    rng = np.random.default_rng(seed)
    F = rng.standard_normal((nw, kf)) / np.sqrt(kf) * 0.3
    n = nw + 1; cones = {"z": 1, "l": nw, "q": [kf + 1]}; m = P.cone_rows(cones)
    A = np.zeros((m, n)); b = np.zeros(m)
    A[0, :nw] = 1.0; b[0] = 1.0; A[1:1 + nw, :nw] = -np.eye(nw); A[1 + nw, nw] = -1.0; A[2 + nw:, :nw] = -F.T
    mu = 0.05 + 0.1 * rng.random((B, nw))
    c = np.concatenate([-mu, np.ones((B, 1))], axis=1)
    return A, np.broadcast_to(b, (B, m)).copy(), c, cones


def sdp(B, k=20, neq=20, seed=0):
    rng = np.random.default_rng(seed)
    d = k * (k + 1) // 2; n = d; cones = {"z": neq, "l": 0, "q": [], "s": [k]}; m = neq + d
    A = np.zeros((m, n)); A[:neq] = rng.standard_normal((neq, n)) / np.sqrt(n); A[neq:] = -np.eye(d)
    mk = lambda: P.sym_to_svec(np.stack([(lambda G: G @ G.T / k + 0.1 * np.eye(k))(rng.standard_normal((k, k))) for _ in range(B)]))
    x0 = mk(); y0 = np.concatenate([rng.standard_normal((B, neq)), mk()], axis=1)
    b = x0 @ A.T + np.concatenate([np.zeros((B, neq)), x0], axis=1)
    return A, b, -(y0 @ A), cones


def run(name, tpl, cones, A_eval, q_eval, eps, reps, note, P_eval=None, p_structure=None, min_seconds=1.0, extra_opts=None):
    ctx = MI355_ctx(p_structure, tpl.problem_data_index, cones, options={"eps": eps, "max_iters": 20000, "raise_on_error": False, **({"acceleration_lookback": int(os.environ["CONFIGS_ACCEL"])} if "CONFIGS_ACCEL" in os.environ else {}), **(extra_opts or {})})
    A_t = torch.from_numpy(A_eval).to(dev).t().contiguous().t().requires_grad_()      # (nnz_aug, B) view of batch-major storage
    q_t = torch.from_numpy(q_eval).to(dev).requires_grad_()
    P_t = torch.from_numpy(P_eval).to(dev).requires_grad_() if P_eval is not None else None
    B = A_eval.shape[1]

    def step():
        A_t.grad = None; q_t.grad = None
        if P_t is not None: P_t.grad = None
        p, d, info, _ = _CvxpyLayer.apply(P_t, q_t, A_t, ctx, {}, True, None)
        p.sum().backward()
        return info
    info = step(); info = step(); torch.cuda.synchronize()          # warm-up (kernel loading, allocator, clock ramp)
    t0 = time.perf_counter(); done = 0
    while done < reps or time.perf_counter() - t0 < min_seconds:      # at least `reps` steps AND `min_seconds` of wall time (short timings right
        info = step(); done += 1                                     # after start-up see the clock ramp: profiles/r01/e_configs.json M row)
        if done >= 2000: break
        if done >= reps and done % 4 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / done
    reps = done
    it = info["iters"].float()
    out = dict(config=name, B=B, n=tpl.n, m=tpl.m, cones={k: (v if not isinstance(v, list) else (f"{len(v)}x{v[0]}" if v else "-")) for k, v in cones.items()},
               eps=eps, ms_per_step=dt * 1e3, problems_per_s=B / dt, iters_mean=float(it.mean()), iters_max=float(it.max()),
               solved=float((info["status"] == 1).float().mean()), path=ctx.engine(dev).last_path, reps=reps,
               lsqr_iters_mean=(float(ctx.engine(dev).last_lsqr_iters.float().mean()) if getattr(ctx.engine(dev), "last_lsqr_iters", None) is not None else None),
               acceleration="plugin default (SCS acceleration_lookback 10; engine: one-pair history where the kernel implements it)", note=note)
    print(json.dumps(out), flush=True)
    return out


def native_box_qp(B, nx=50, seed=0):
    """BASELINE config 2 in its native form (problems.native_box_qp_batch): min 1/2 x^T (2 F^T F) x - 2 g^T F x, lo <= x <= hi; same F, g, lo, hi as box_qp_batch"""
    An, bn, qn, _, pst, pv = P.native_box_qp_batch(nx, B, seed)
    return An, bn, qn, pst, np.broadcast_to(pv[:, None], (len(pv), B)).copy()


WANT = [k for k in os.environ.get("CONFIGS", "M,C2,C2Q,C2N,C3,C4,C5,E").split(",") if k]
res = []
for key, note in (("M", "metric configuration"), ("C2", "nonneg cone only (random LP)"), ("C3", "SOCP n=100, 10 SOC(11)")):
    if key not in WANT: continue
    cfg = P.CONFIGS[key]
    tpl = P.dense_template(cfg["n"], cfg["cones"])
    A, b, c = P.generate(cfg["n"], cfg["cones"], 4096, seed=0)
    res.append(run(key, tpl, cfg["cones"], *tpl.values_from_dense(A, b, c), 1e-4, 20 if key != "C2" else 3, note))
if "C2Q" in WANT:
    A, b, c, cones = P.box_qp_batch(50, 4096, seed=0)
    tpl = P.dense_template(A.shape[2], cones, pattern=(A[0] != 0))
    res.append(run("C2Q", tpl, cones, *tpl.values_from_dense(A, b, c), 1e-4, 10, "box QP n=50 in SOC-epigraph form (BASELINE config 2 as DIFFCP sees it); F shared, g/lo/hi batched"))
if "C2N" in WANT:
    An, bn, qn, pst, Pv = native_box_qp(4096)
    conesN = {"z": 0, "l": 100, "q": [], "s": []}
    tplN = P.dense_template(50, conesN, pattern=(An != 0))
    res.append(run("C2N", tplN, conesN, *tplN.values_from_dense(np.broadcast_to(An, (4096,) + An.shape).copy(), bn, qn), 1e-4, 20,
                   "box QP n=50 in NATIVE form (P = 2 F^T F inside the kernels, 100 box rows): BASELINE config 2 (i)", P_eval=Pv, p_structure=pst))
if "E" in WANT:
    ecfg = dict(n=40, cones={"z": 0, "l": 12, "q": [4], "s": [], "ep": 24})      # the cone shape of a 12-sample, 3-feature logistic-regression layer
    tplE = P.dense_template(ecfg["n"], ecfg["cones"])
    Ae, be, ce_ = P.generate(ecfg["n"], ecfg["cones"], 4096, seed=0)
    res.append(run("E", tplE, ecfg["cones"], *tplE.values_from_dense(Ae, be, ce_), 1e-4, 10, "24 exponential cones + nonneg + SOC(4), n=40, m=88 (logistic-regression layer shape), dense random A"))
if "C4" in WANT:
    A, b, c, cones, tpl = P.sdp_c4_batch(1024, seed=0)
    vals4 = tpl.values_from_dense(np.broadcast_to(A, (1024,) + A.shape).copy(), b, c)
    res.append(run("C4", tpl, cones, *vals4, 1e-4, 5, "SDP, one 20x20 PSD cone, A shared (BASELINE config 4); adjoint = LSQR on diffcp's full system with diffcp's stopping rule (plugin default)"))
    res.append(run("C4r", tpl, cones, *vals4, 1e-4, 5, "the same with solver_args adjoint_system='reduced' (r_tau pinned to 0: the system of rounds 1-4)", extra_opts={"adjoint_system": "reduced"}))
if "C5" in WANT:
    Bp = int(os.environ.get("C5_BATCH", "16384"))
    A, b, c, cones, tpl = P.portfolio_c5_batch(Bp, seed=0)
    A1, _ = tpl.values_from_dense(A[None], b[None], c[:1])          # A, b shared: tile one instance's values (the dense (B, m, n) array would be 36 GB)
    vals5 = (np.repeat(A1, Bp, axis=1), np.concatenate([c.T, np.zeros((1, Bp))], axis=0))
    res.append(run("C5", tpl, cones, *vals5, 1e-4, 2, "portfolio n=501, A shared, returns batched (BASELINE config 5), 1 GPU; adjoint = LSQR on diffcp's full system with diffcp's stopping rule (plugin default)", min_seconds=0.0))
    res.append(run("C5r", tpl, cones, *vals5, 1e-4, 2, "the same with solver_args adjoint_system='reduced' (r_tau pinned to 0: the system of rounds 1-4)", min_seconds=0.0, extra_opts={"adjoint_system": "reduced"}))
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/configs.json"
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
