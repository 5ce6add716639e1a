"""bench.py -- forward+backward problems/sec on the BASELINE.json metric configuration.

One "step" = one pass of the hot path over one batch: the plugin boundary call
    primal, dual = _CvxpyLayer.apply(None, q_eval, A_eval, ctx, solver_args, needs_grad=True)
    primal.sum().backward()
on B=4096 synthetic instances of the metric config (n=50, m=100: 20 nonneg rows + 8 SOC(10), dense A; A, b, c all
batched) with q_eval / A_eval already resident in HBM in the reference's batch-minor layout.  The layout pass,
the solve, the status read-back and the adjoint are all inside the timed region.
N>1: `--scaling weak` (default): each rank holds its own --batch instances; `--scaling strong`: --batch is the WHOLE job's batch, sharded contiguously over the
ranks (BASELINE.json: "SOCP n=100, batch=4096, 1->8 GPU batch shard", `--config C3 --scaling strong`; "portfolio n=500, batch=16384 sharded across 8 GPUs",
`--config C5 --batch 16384 --scaling strong`).  The step includes the all-gather of primal/dual (every rank evaluates the same loss on the gathered tensor:
parallel.py's loss="replicated" contract, no collective in the backward); the JSON line carries the exchange on its own (`allgather_ms`) and every rank's own
step time (`ms_per_step_per_rank`).
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (re-exec under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); under an external launcher
(the driver's torchrun line) RANK / LOCAL_RANK / WORLD_SIZE are read from the environment.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# thread placement of the CPU leg (read by the OpenMP runtime when the oracle's library is loaded): one thread per place, neighbours first
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "threads")

from cvxpylayers_amd import problems as P  # noqa: E402
from cvxpylayers_amd.interfaces.mi355_if import MI355_ctx, _CvxpyLayer  # noqa: E402
from cvxpylayers_amd.parallel import gather_solution  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TF = 78.6     # half the 157.3 TF fp32 vector peak


def reference_cpu_baseline(n, cones, solver_args, sample, seed, budget_s=12.0):
    """The TRUE reference CPU path, when it is installed ("every benchmark script must try `import cvxpy, diffcp` and time the
    real layer when available"): diffcp.solve_and_derivative_batch + its adjoint on the same cone programs, all host cores (diffcp's default
    n_jobs = -1), same eps / max_iters / acceleration defaults.  Returns None where cvxpy / diffcp are absent (this image, the GPU box)."""
    try:
        import cvxpy  # noqa: F401  (the reference's own import; canonicalisation is not timed)
        import diffcp
        import scipy.sparse as sp
    except Exception:
        return None
    A, b, c = P.generate(n, cones, sample, seed=seed)
    cone_dict = {"z": int(cones.get("z", 0)), "l": int(cones.get("l", 0)), "q": list(cones.get("q", [])), "s": list(cones.get("s", [])), "ep": int(cones.get("ep", 0)),
                 "p": list(cones.get("p", []))}
    As = [sp.csc_matrix(A[i]) for i in range(sample)]; bs = list(b); cs = list(c); Ks = [cone_dict] * sample
    kw = {k: v for k, v in solver_args.items() if k in ("max_iters", "acceleration_lookback")}
    if "eps" in solver_args:          # SCS 3 has eps_abs / eps_rel and no `eps` (the reference's diffcp_if.py never passes `eps` either): same tolerance on both
        kw["eps_abs"] = kw["eps_rel"] = solver_args["eps"]
    done, passes, t0 = 0, 0, time.perf_counter()
    try:
        while True:
            xs, ys, ss, D, DT = diffcp.solve_and_derivative_batch(As, bs, cs, Ks, mode="lsqr", **kw)
            DT([np.ones(n)] * sample, [np.zeros(len(bs[0]))] * sample, [np.zeros(len(bs[0]))] * sample)
            done += sample; passes += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s or passes >= 200:
                break
    except Exception as e:            # an installed but incompatible stack must not be reported as the reference: fall back to the port, and say so
        print(f"[bench] reference CPU stack present but unusable ({type(e).__name__}: {e}); timing the oracle port instead", file=sys.stderr)
        return None
    return dict(value=done / dt, unit="problems/s", cores=os.cpu_count(), kind="reference",
                note=f"diffcp {getattr(diffcp, '__version__', '?')} solve_and_derivative_batch + adjoint (the reference's own CPU path), n_jobs = all cores",
                sample=f"{passes} passes over {sample} instances of the same workload, {dt:.1f} s of wall time")


def cpu_baseline(n, cones, solver_args, sample, seed, budget_s=12.0):
    """CPU baseline on the host cores of this box.  First choice: the reference's own stack (reference_cpu_baseline, kind "reference"); it is
    not installable in this image or on the GPU box, so what actually runs is the oracle (kind "port": this repository's CPU restatement of
    the diffcp/SCS path, OpenMP over instances).  Bounded sample: whole passes (forward solve + LSQR adjoint, diffcp's default mode) over
    `sample` instances of the same workload are repeated until about `budget_s` seconds of wall time have been spent; value = instances / time."""
    ref = reference_cpu_baseline(n, cones, solver_args, sample, seed, budget_s)
    if ref is not None:
        return ref
    from oracle import oracle
    A, b, c = P.generate(n, cones, sample, seed=seed)
    host = host_cpu_info()
    omp_threads = oracle.num_threads()

    def one_pass(k, nthreads):
        r = oracle.solve_batch(A[:k], b[:k], c[:k], cones, nthreads=nthreads, **solver_args)
        g = oracle.adjoint_batch(A[:k], b[:k], c[:k], cones, r["x"], r["y"], r["s"], np.ones_like(r["x"]), np.zeros_like(r["y"]), mode="lsqr", nthreads=nthreads)
        return r, g

    def rate(nthreads, budget):
        k = min(sample, max(64, 32 * nthreads))
        one_pass(min(k, 2 * nthreads), nthreads)                      # warm (threads, page faults)
        done, passes, t0 = 0, 0, time.perf_counter()
        while True:
            r, g = one_pass(k if nthreads > 1 else min(k, 128), nthreads)
            done += len(r["iters"]); passes += 1
            dt = time.perf_counter() - t0
            if dt >= budget or passes >= 200:
                return done / dt, passes, dt, r, g, len(r["iters"])
    # one thread first (the per-core rate), then the thread counts this box offers: every core the process may run on, and half of them
    # (SMT siblings share the fp64 units: the better of the two is the baseline, and both are reported)
    v1, _, _, _, _, _ = rate(1, min(2.0, budget_s / 6))
    usable = max(1, min(omp_threads, host["affinity_cpus"], int(host["cgroup_cpus"]) if host["cgroup_cpus"] else 1 << 30))
    tried = {}
    for nt in sorted({usable, max(1, usable // 2)}, reverse=True):
        tried[nt] = rate(nt, budget_s / 2 if nt == usable else budget_s / 4)
    best = max(tried, key=lambda k_: tried[k_][0])
    v, passes, dt, r, g, k = tried[best]
    return dict(value=v, unit="problems/s", cores=best, kind="port",
                note="own port, untuned: this repository's plain-C + OpenMP restatement of the SCS / diffcp algorithms (oracle/), not SCS / diffcp themselves",
                sample=f"{passes} passes over {k} instances of the same workload (forward + LSQR adjoint, diffcp's default mode), {dt:.1f} s of wall time",
                single_thread_value=v1, threads_effective=v / v1, parallel_efficiency=v / v1 / best,
                thread_sweep={str(nt): tried[nt][0] for nt in tried}, host=host, omp_max_threads=omp_threads,
                mean_iters=float(r["iters"].mean()), mean_lsqr_iters=float(g["lsqr_iters"].mean()))


def adjoint_roofline(eng, infos, tpl, cones, B, bwd_ms, bwd_bytes):
    """`roofline.backward`: the adjoint kernel against both roofs.  HBM: algorithmic bytes (8(nnzA + 2n + 3m) read + 8(nnzA + m + n) written per instance)
    over its HIP-event time.  fp64: the flops the elimination EXECUTES, from the active sets of the last solutions (neq equality rows, KW weighted rows, nf = n - neq):
    row elimination 2 neq^2 (n + 1), null-space transform 2 KW neq (n + 1), reduced Hessian 2 KW (nf + 1)^2, sweep 2 nf^2 (nf + 1), products behind it 4 KW n."""
    from cvxpylayers_amd import _lib
    n, m = tpl.n, tpl.m
    z, l, q = int(cones.get("z", 0)), int(cones.get("l", 0)), list(cones.get("q", []))
    out = {"kernel": "adjoint (k_backward_ns: search-free null-space elimination)" if _lib.lib().ce_adjoint_ns_variant(eng._h) >= 0 else "adjoint (k_backward_rt / k_backward)",
           "hbm": {"achieved": bwd_bytes * B / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": bwd_bytes * B},
           "ms": bwd_ms}
    out["hbm"]["frac"] = out["hbm"]["achieved"] / HBM_PEAK_GBS
    try:
        sol = getattr(eng, "_last_solution", None)
        if sol is not None and q is not None and not cones.get("s") and not cones.get("ep"):
            v = (sol[1] - sol[2])                                   # y - s of the most recent solve
            neq = torch.full((v.shape[0],), float(z), dtype=torch.float64, device=v.device) + (v[:, z:z + l] > 0).sum(dim=1)
            kw = torch.zeros_like(neq)
            off = z + l
            for d in q:
                t0, zz = v[:, off], v[:, off + 1:off + d]
                nz = zz.norm(dim=1)
                inside = nz <= t0; bnd = (~inside) & ~(nz <= -t0)
                neq = neq + inside * float(d) + bnd * 1.0
                kw = kw + bnd * float(d)
                off += d
            nf = (n - neq).clamp_min(0.0)
            fl = 2 * neq ** 2 * (n + 1) + 2 * kw * neq * (n + 1) + 2 * kw * (nf + 1) ** 2 + 2 * nf ** 2 * (nf + 1) + 4 * kw * n
            tf = float(fl.sum().item()) / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
            out["fp64"] = {"achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TF, "executed_flops_per_launch": float(fl.sum().item()),
                           "mean_equality_rows": float(neq.mean().item()), "mean_weighted_rows": float(kw.mean().item()), "mean_reduced_order": float(nf.mean().item())}
    except Exception as e:          # (introspection only: never fail the benchmark line)
        out["fp64"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def host_cpu_info():
    """What the CPU leg may use on this box: the affinity mask, the cgroup quota, the logical CPU count (so that `cores` can be checked)."""
    info = dict(cpu_count=os.cpu_count(), affinity_cpus=len(os.sched_getaffinity(0)), cgroup_cpu_max=None, cgroup_cpus=None, omp_env={k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")})
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            info["cgroup_cpu_max"] = " ".join(txt)
            if f.endswith("cpu.max") and txt[0] != "max":
                info["cgroup_cpus"] = float(txt[0]) / float(txt[1])
            elif f.endswith("quota_us") and int(txt[0]) > 0:
                info["cgroup_cpus"] = int(txt[0]) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        info["cpu_model"] = model[0] if model else None
    except Exception:
        info["cpu_model"] = None
    return info


def pmc_traffic(kernel_short):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*/?_pmc_summary.json,
    made by scripts/gpu_round.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md).
    Returns (bytes, source) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*pmc_summary.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k.startswith(kernel_short) and "FETCH_SIZE_bytes_per_launch" in v and "WRITE_SIZE_bytes_per_launch" in v:
                return v["FETCH_SIZE_bytes_per_launch"] + v["WRITE_SIZE_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, None


def sq_limiter(kernel_short):
    """What the committed SQ-counter pass (profiles/*/?_sq_summary.json, scripts/gpu_r3.sh sq) says binds the dominant kernel: share of wave
    time with a VALU instruction in flight, share spent waiting, LDS bank conflicts.  The kernel touches HBM once and iterates in LDS / registers,
    so neither the HBM nor the MFMA roofline binds it; these counters are the evidence."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*sq_summary.json")))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k.startswith(kernel_short) and isinstance(v, dict):
                keep = {kk: vv for kk, vv in v.items() if kk in ("valu_busy", "wait_any", "wait_inst_any", "active_inst_any", "lds_conflict", "mfma_util", "simd_valu_util", "cu_lds_inst_util", "cu_lds_array_util", "resident_waves_per_cu")}      # shares of wave time
                return dict(source=os.path.relpath(f, ROOT), kernel=k, counters=keep,
                            reading="SQ counters of this file. Per unit (simd_valu_util, cu_lds_*_util; scripts/sq_summary.py): a SIMD has a VALU instruction in its pipe for about half of the launch, a CU's LDS is busy "
                                    "for about half of it, with ~10 of 12 wave slots occupied on average (tail of the launch) -- neither pipe is saturated. Per wave (valu_busy, wait_*): a wave issues VALU work for a fifth of its "
                                    "time and waits (barriers, LDS operand reads, dependent chains) for most of the rest; three workgroups per CU overlap these chains but do not hide them. The kernel is bound by the latency of "
                                    "one workgroup's barrier-separated phases together with the two half-loaded pipes they share, not by HBM (touched once) or MFMA; about a sixth of its VALU instructions are the fp64 FMAs the flop count knows, hence the fp64 fraction")
    return None


def shard_sizes(total, world):
    """contiguous shards of a batch of `total` instances (parallel.shard_bounds)"""
    from cvxpylayers_amd.parallel import shard_bounds
    return [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]


def workload_dims(config):
    if config == "C5":
        return 501, 552
    cfg = P.CONFIGS[config]
    return cfg["n"], P.cone_rows(cfg["cones"])


def build_workload(config, B, seeds, dev):
    """The rotating batches of one rank: [(A_eval (nnz_aug, B), q_eval (n + 1, B))] resident in HBM, the template and the cones.  M / C3 (and any dense entry of
    problems.CONFIGS): A, b, c all batched, seeds as given.  C5: BASELINE config 5 (portfolio n = 501, m = 552; A and b SHARED, mu per instance): one value matrix
    for all slots, the objectives differ."""
    if config == "C5":
        batches, A1t = [], None
        for sd in seeds:
            A, b, c, cones, tpl = P.portfolio_c5_batch(B, seed=sd)
            if A1t is None:
                A1, _ = tpl.values_from_dense(A[None], b[None], c[:1])
                A1t = torch.from_numpy(A1[:, 0]).to(dev)[None, :].repeat(B, 1).contiguous().t().requires_grad_()      # (nnz_aug, B) view of batch-major storage: what the frontend hands over
            q = torch.from_numpy(np.ascontiguousarray(np.concatenate([c.T, np.zeros((1, B))], axis=0))).to(dev).requires_grad_()
            batches.append((A1t, q))
        return tpl, cones, batches, f"config C5: portfolio n={tpl.n} m={tpl.m} (1 zero + 500 nonneg + SOC(51)), A and b shared, mu batched"
    cfg = P.CONFIGS[config]
    n, cones = cfg["n"], cfg["cones"]
    tpl = P.dense_template(n, cones)
    batches = []
    for sd in seeds:
        A, b, c = P.generate(n, cones, B, seed=sd)
        A_eval, q_eval = tpl.values_from_dense(A, b, c)
        batches.append((torch.from_numpy(A_eval).to(dev).requires_grad_(),      # (nnz_aug, B) batch-minor, as the reference's frontend hands it over
                        torch.from_numpy(q_eval).to(dev).requires_grad_()))
    return tpl, cones, batches, (f"config {config}: n={n} m={tpl.m} cones l={cones.get('l', 0)} q={cones.get('q', [])} dense A (nnzA={tpl.nnzA}), A,b,c batched")


def dry_run_ranks(args):
    """`--dry-run-ranks N`: the N>1 plumbing of this script on CPU (no GPU, backend gloo) -- self-spawn under torch.distributed.run with the
    127.0.0.1 rendezvous, RANK / LOCAL_RANK / WORLD_SIZE from the environment, barrier-bracketed timing with the MAX over ranks, the fused
    differentiable all-gather of (primal | dual) rows, rank 0 printing ONE JSON line.  The step is a stand-in (no solver runs: value is null)."""
    world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    n, m = workload_dims(args.config)
    Btot = min(args.batch, 64 * world if args.scaling == "strong" else 64)
    sizes = shard_sizes(Btot, world) if args.scaling == "strong" else [Btot] * world          # strong: the job's batch split contiguously (ragged when it does not divide)
    B = sizes[rank]
    # K rotating batches per rank, as in the real run (slot k of rank r is seeded r + world * k there; here it is FILLED with that number + 1): every rank must
    # visit the same slot at the same step, or the gathered batch would mix slots
    K_rot = max(1, args.rotate)
    xs = [torch.full((B, n), float(rank + world * k + 1), dtype=torch.float64, requires_grad=True) for k in range(K_rot)]
    y = torch.zeros((B, m), dtype=torch.float64)
    ok = True
    dist.barrier(); t0 = time.perf_counter()
    for step_no in range(args.steps):
        slot = step_no % K_rot
        x = xs[slot]
        x.grad = None
        primal, dual = gather_solution(x * 1.0, y, sizes if args.scaling == "strong" else None)
        primal.sum().backward()
        want = n * sum(sizes[r] * (r + world * slot + 1) for r in range(world))          # every rank contributed the SAME slot, with its own shard size
        ok = ok and tuple(primal.shape) == (sum(sizes), n) and float(primal.sum()) == want and bool((x.grad == 1.0).all())
    dist.barrier(); dt = time.perf_counter() - t0
    tdt = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
    okt = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64); dist.all_reduce(okt, op=dist.ReduceOp.MIN); ok = bool(okt.item() == 1.0)
    if rank == 0:
        print(json.dumps({"metric": "dry run of the multi-rank plumbing (no solver)", "value": None, "unit": "problems/s", "n_gpus": 0, "ranks": world, "backend": "gloo", "scaling": args.scaling,
                          "config": args.config, "shard_sizes": sizes,
                          "steps": args.steps, "rotating_batches": K_rot, "ms_per_step": 1e3 * float(tdt.item()) / max(args.steps, 1), "dry_run": True, "gather_ok": bool(ok),
                          "local_rank_env": os.environ.get("LOCAL_RANK"), "master_addr": os.environ.get("MASTER_ADDR")}))
    dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: ~0.7 s, long enough to be past the clock ramp)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--config", default="M", help="M (the metric configuration), C3 (SOCP n=100), C5 (portfolio n=501, shared A), or any dense entry of problems.CONFIGS")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"), help="N > 1: weak = --batch instances PER RANK; strong = --batch instances for the whole job, sharded over the ranks")
    ap.add_argument("--extras", type=int, default=1, help="1: also time (outside the headline) the README-recommended solver settings and the asynchronous forward of raise_on_error=False")
    ap.add_argument("--eps", type=float, default=1e-4, help="eps_abs=eps_rel (SCS default 1e-4)")
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--dispatch-history", type=int, default=1, help="longest-first dispatch from the previous step's iteration counts (the plugin's default); 0: index order")
    ap.add_argument("--rotate", type=int, default=4, help="K distinct seeded batches resident in HBM, visited round-robin by the warm-up and the timed loop (K = 1: one batch "
                                                          "re-solved every step, which lets the plugin's history heuristics replay instead of predict)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--accel", type=int, default=10,
                    help="acceleration_lookback handed to both sides.  10 (default) = SCS's own default, which diffcp forwards: type-I Anderson "
                         "acceleration every 10 iterations (the engine keeps a one-pair history, the CPU oracle the full lookback: same iteration "
                         "counts within 2.5 %%, profiles/r02/aa_memory.json); 0 = plain iteration.")
    ap.add_argument("--dry-run-ranks", type=int, default=0, help="N > 0: exercise the multi-rank launch / gather plumbing with N CPU ranks over gloo (no GPU, no solver)")
    args = ap.parse_args()

    if args.dry_run_ranks > 0:
        if "WORLD_SIZE" not in os.environ:
            args.gpus = args.dry_run_ranks          # same self-spawn path as --gpus N
        else:
            return dry_run_ranks(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare: create the N ranks (one process per GPU) and let rank 0 of that job print the JSON line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):      # one node (the contract): bootstrap on the loopback interface, no interface / IB probing
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")                            # (RCCL's one-rank bootstrap took 3 s or 31 s depending on the box without it; data moves over xGMI P2P either way)
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    sizes = shard_sizes(args.batch, world) if (args.scaling == "strong" and world > 1) else [args.batch] * world
    B = sizes[rank]
    solver_args = {"eps": args.eps, "max_iters": 10000, "acceleration_lookback": args.accel}
    # K distinct batches (seeds rank, rank + world, ...: no two ranks and no two slots share one), all resident in HBM before anything is timed.  The loop visits
    # them round-robin, so what the plugin remembers of the previous call (iteration counts -> dispatch order, the largest adjoint system -> tile) belongs to a
    # DIFFERENT batch of the same distribution: the heuristics have to predict, as in a training loop over fresh mini-batches.
    K_rot = max(1, args.rotate)
    tpl, cones, batches, workload_desc = build_workload(args.config, B, [rank + world * k for k in range(K_rot)], dev)
    n = tpl.n
    ctx = MI355_ctx(None, tpl.problem_data_index, cones, options={**solver_args, "dispatch_history": bool(args.dispatch_history)})
    eng = ctx.engine(dev)
    step_no = [0]
    last_info = [None] * K_rot          # the most recent `info` of every slot (iteration counts of ALL rotating batches enter the flop count)

    def step(fixed=None):
        slot = (step_no[0] % K_rot) if fixed is None else fixed
        A_t, q_t = batches[slot]
        step_no[0] += 1
        A_t.grad = None
        q_t.grad = None
        primal, dual, info, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, {}, True, None)
        if world > 1:
            primal, dual = gather_solution(primal, dual, sizes if args.scaling == "strong" else None)      # one fused RCCL all-gather per step (ragged shards are padded)
        primal.sum().backward()
        last_info[slot] = info
        return info

    # clock ramp: a run as short as the driver's (--steps 20 --warmup 5: 60 ms of GPU work) would be timed on a GPU that has not reached its steady clocks yet (docs/ROUND_NOTES_r3_r4.md: short measurements right
    # after process start see the ramp).  Untimed steps until 0.3 s have passed, in ADDITION to the W warm-up steps of the contract; reported as `prewarm_steps`.
    prewarm_steps = 0
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw < 0.3) if world == 1 else (prewarm_steps < 100):      # (several ranks: a FIXED count -- every step holds a collective, the ranks must agree on their number)
        info = step(); prewarm_steps += 1
        if prewarm_steps % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        info = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.set_profiling(2)          # HIP events around the FORWARD launches of the timed region only (the roofline kernel); the other kernels' averages come from a few extra steps below:
    eng.reset_profile()           # every bracketed launch costs two event records of host time in front of it, and the adjoint's sit in the gap in which the GPU waits for the host
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = [1e3 * dt / args.steps]
    if world > 1:
        tall = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt], dtype=torch.float64, device=dev))
        per_rank_ms = [1e3 * float(t.item()) / args.steps for t in tall]
        dt = max(float(t.item()) for t in tall)          # the MAX over ranks is the job's time
    fwd_ms, nf = eng.profile(0)
    eng.set_profiling(12)         # adjoint + layout launches: bracketed on a few untimed steps
    eng.reset_profile()
    for _ in range(max(8, min(args.steps, 40))):
        step()
    torch.cuda.synchronize()
    bwd_ms, nb = eng.profile(1)
    lay_ms, nl = eng.profile(2)
    eng.set_profiling(False)
    # Two more timings of the same step, outside the headline (one rank only): `index_order` = the rotation with the dispatch history switched off, `replay` = ONE
    # batch re-solved every step with the history on (the previous call's iteration counts and largest adjoint system are then exact: what rounds 1-4 reported).
    def timed(ns, fixed=None):
        for _ in range(3):
            step(fixed)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(ns):
            step(fixed)
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) * 1e3 / ns
    other_ms = replay_ms = None
    if world == 1:
        ns_other = max(8, min(args.steps, 40))
        eng.set_dispatch_history(not args.dispatch_history)
        other_ms = timed(ns_other)
        eng.set_dispatch_history(True)
        replay_ms = timed(ns_other, fixed=0)
        eng.set_dispatch_history(bool(args.dispatch_history))
    # Outside the headline (one rank): (1) the solver settings the reference's README recommends for accurate gradients (eps 1e-8, max_iters 10000,
    # acceleration_lookback 0: /root/reference README "solver_args" section) on the same rotating batches; (2) the same step with raise_on_error=False -- the
    # caller waives the raise-from-forward contract, the plugin then never waits for the device (failure masking happens on the device): what the one contractual
    # host round trip of the default path costs.
    extras = {}
    if world == 1 and args.extras:
        def timed_with(sa, ns):
            def st():
                A_t, q_t = batches[step_no[0] % K_rot]; step_no[0] += 1
                A_t.grad = None; q_t.grad = None
                primal, dual, info2, _ = _CvxpyLayer.apply(None, q_t, A_t, ctx, sa, True, None)
                primal.sum().backward()
                return info2
            for _ in range(4):
                info2 = st()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(ns):
                info2 = st()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) * 1e3 / ns, info2
        ns_x = max(8, min(args.steps, 40))
        import warnings as _w
        with _w.catch_warnings():
            _w.simplefilter("ignore")
            ms_rd, info_rd = timed_with({"eps": 1e-8, "max_iters": 10000, "acceleration_lookback": 0}, ns_x)
            ms_as, _ = timed_with({"raise_on_error": False}, ns_x)
            ms_sy, _ = timed_with({}, ns_x)
        torch.cuda.synchronize()
        extras["readme_recommended_settings"] = {"solver_args": {"eps": 1e-8, "max_iters": 10000, "acceleration_lookback": 0}, "ms_per_step": ms_rd, "problems_per_s": B / (ms_rd * 1e-3),
                                                 "mean_iters": float(info_rd["iters"].float().mean().item()), "solved_fraction": float((info_rd["status"] == 1).float().mean().item()),
                                                 "source": "reference README (solver_args for accurate derivatives); same rotating batches, same step"}
        extras["async_forward"] = {"solver_args": {"raise_on_error": False}, "ms_per_step": ms_as, "ms_per_step_default_same_loop": ms_sy, "problems_per_s": B / (ms_as * 1e-3),
                                   "note": "raise_on_error=False waives the reference's raise-from-forward contract: no host round trip in forward (failed instances are masked on the "
                                           "device, the outcome is reported by the next call); the difference to ms_per_step_default_same_loop is what that round trip costs"}
    allgather_ms = None
    if world > 1:         # the exchange step on its own: one fused RCCL all-gather of (B, n + m) rows per step (HIP events on this stream)
        with torch.no_grad():
            pr = torch.zeros((B, n), dtype=torch.float64, device=dev); du = torch.zeros((B, tpl.m), dtype=torch.float64, device=dev)
            gsz = sizes if args.scaling == "strong" else None
            for _ in range(5):
                gather_solution(pr, du, gsz)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                gather_solution(pr, du, gsz)
            e1.record(); e1.synchronize()
            allgather_ms = e0.elapsed_time(e1) / 50

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = sum(sizes) * args.steps / dt
        m, nnzA = tpl.m, tpl.nnzA
        cfgname = args.config
        slots = [li for li in last_info if li is not None]
        iters = torch.stack([li["iters"] for li in slots]).cpu().numpy().astype(np.float64)          # (slots visited, B)
        solved = float(np.mean([float((li["status"] == 1).float().mean().item()) for li in slots]))
        # algorithmic HBM bytes per instance: forward 8(nnzA+m+n) read + 8(n+2m) written
        fwd_bytes = 8 * (nnzA + m + n) + 8 * (n + 2 * m)
        bwd_bytes = 8 * (nnzA + 2 * n + 3 * m) + 8 * (nnzA + m + n)
        ach = fwd_bytes * B / (fwd_ms * 1e-3) / 1e9 if fwd_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic("k_fwd")
        # algorithmic fp64 flops of the forward kernel: setup m n^2 + n^3/3, per iteration 4 nnzA + 2 n^2 + 10(n+m)
        flops = B * (m * n * n + n ** 3 / 3.0) + float(iters.sum(axis=1).mean()) * (4 * nnzA + 2 * n * n + 10 * (n + m))
        vtf = flops / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
        out = {
            "metric": ("forward+backward problems/sec, batch=4096 n=50 m=100 SOC" if (args.config == "M" and B == 4096) else
                       f"forward+backward problems/sec, batch={B} n={n} m={m} config {args.config}"), "value": value, "unit": "problems/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm_steps, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_desc + (f", B={B} per GPU" if args.scaling == "weak" or world == 1 else f", B={args.batch} for the job, shards {sizes}") +
                                   f", eps_abs=eps_rel={args.eps}, max_iters=10000, "
                                   + (f"Anderson acceleration (SCS default: acceleration_lookback={args.accel}, interval 10; engine: one-pair history)" if args.accel > 0 else "acceleration off")
                                   + f"; {K_rot} distinct seeded batches resident in HBM, visited round-robin (warm-up and timed steps)" +
                                   "; step = plugin forward (layout pass + solve + status) + backward (adjoint VJP of sum(x))",
                       "batch_per_gpu": B, "rotating_batches": K_rot, "parallelism": f"batch-shard x{world}", "acceleration_lookback": int(args.accel),
                       "solved_fraction": solved, "mean_iters": float(iters.mean())},
            "roofline": {"bound": "valu-issue/latency", "kernel": "forward (k_fwd2 / k_forward_rt / k_forward): the longest kernel of the step",
                         "achieved": vtf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": vtf / FP64_VALU_PEAK_TF,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_flops_per_launch": flops,
                         "note": "the binding roof is what the SQ counters say (`limiter`): the fp64 vector pipe's issue rate and the latency of dependent chains / barriers / LDS, "
                                 "so achieved / peak are algorithmic fp64 flops against the fp64 VALU peak.  The iteration lives in LDS and registers and HBM is touched once by "
                                 "construction: the HBM side is kept under `hbm` (algorithmic bytes / kernel time against 8 TB/s) and `traffic` (PMC bytes per launch)",
                         "limiter": sq_limiter("k_fwd"),
                         "hbm": {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": fwd_bytes * B,
                                 "traffic_over_algorithmic": (traffic / (fwd_bytes * B)) if traffic else None}},
            "kernels_ms": {"k_forward": fwd_ms, "k_backward": bwd_ms, "k_transpose": lay_ms, "launches": [nf, nb, nl],
                           "bwd_algorithmic_GBps": bwd_bytes * B / (bwd_ms * 1e-3) / 1e9 if bwd_ms > 0 else 0.0},
            "host_gap_ms": ms_per_step - (fwd_ms + bwd_ms + 2 * lay_ms) if args.config != "C5" else None,
            "iters": {"mean": float(iters.mean()), "max": float(iters.max())},
            "dispatch": {"history": bool(args.dispatch_history), "ms_per_step_rotating_with_history": ms_per_step if args.dispatch_history else other_ms,
                         "ms_per_step_rotating_index_order": other_ms if args.dispatch_history else ms_per_step,
                         "ms_per_step_replay_one_batch": replay_ms,
                         "note": "the plugin records every call's iteration counts and dispatches the next call's workgroups longest-first ONLY when the history has been predictive "
                                 "(>= 70 % of the instances in the same check interval as the call before: decided on the device); the adjoint's first tile is sized by the previous call's "
                                 "largest system.  `value` is measured on rotating batches: the history does not predict there and the index order runs; `replay` re-solves one batch "
                                 "(history exact, applied from the third call on) and is reported for comparison with rounds 1-4 only"},
            "launch": eng.launch_info(),
        }
        out["roofline"]["backward"] = adjoint_roofline(eng, slots, tpl, cones, B, bwd_ms, bwd_bytes)
        out["ms_per_step_per_rank"] = per_rank_ms
        out.update(extras)
        if allgather_ms is not None:
            out["allgather_ms"] = allgather_ms
            out["allgather_bytes_per_rank"] = 8 * B * (n + m)
            out["allgather_share_of_step"] = allgather_ms / ms_per_step
        if not args.no_cpu and world == 1 and args.config != "C5":          # (the CPU leg draws dense instances of problems.CONFIGS: not config 5's shared-A portfolio)
            out["cpu_baseline"] = cpu_baseline(n, cones, solver_args, args.cpu_sample, seed=0)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
