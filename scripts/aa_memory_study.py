"""Anderson-acceleration memory study on the CPU oracle (test infrastructure): iteration counts of the plain iteration, the
engine's one-pair history (aa_mem = 1) and SCS's default lookback 10 (and 5) on every BASELINE configuration, eps 1e-4.
Writes profiles/r02/aa_memory.json.   python scripts/aa_memory_study.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cvxpylayers_amd import problems as P
from oracle import oracle

out = {}
def run(name, A, b, c, cones, B):
    row = {}
    for mem in (0, 1, 5, 10):
        t0 = time.time()
        r = oracle.solve_batch(A, b, c, cones, eps=1e-4, max_iters=20000, acceleration_lookback=mem, acceleration_interval=10)
        row[f"mem{mem}"] = dict(mean_iters=float(r["iters"].mean()), max_iters=int(r["iters"].max()), solved=float((r["status"] == 1).mean()), seconds=round(time.time() - t0, 2))
    row["B"] = B
    row["mem1_vs_mem10"] = row["mem1"]["mean_iters"] / row["mem10"]["mean_iters"]
    out[name] = row
    print(name, {k: (v["mean_iters"] if isinstance(v, dict) else v) for k, v in row.items()}, flush=True)

B = 256
cfg = P.CONFIGS["M"]; A, b, c = P.generate(cfg["n"], cfg["cones"], B, seed=0); run("M  (metric: n=50 m=100 SOC)", A, b, c, cfg["cones"], B)
A, b, c, cones = P.box_qp_batch(50, B, seed=0); run("C2 (box QP n=50, epigraph form)", A, b, c, cones, B)
cfg = P.CONFIGS["C3"]; A, b, c = P.generate(cfg["n"], cfg["cones"], 128, seed=0); run("C3 (SOCP n=100)", A, b, c, cfg["cones"], 128)
A, b, c, cones, tpl = P.sdp_c4_batch(32, seed=0); run("C4 (SDP 20x20)", np.broadcast_to(A, (32,) + A.shape).copy(), b, c, cones, 32)
A, b, c, cones, tpl = P.portfolio_c5_batch(16, seed=0); run("C5 (portfolio n=501)", np.broadcast_to(A, (16,) + A.shape).copy(), np.broadcast_to(b, (16,) + b.shape).copy(), c, cones, 16)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02", "aa_memory.json"), "w"), indent=1)
