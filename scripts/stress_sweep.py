"""Randomised sweep of the engine against the CPU oracle: tests/stress_kit.py (what it checks) run for as many shapes / seeds as asked.
usage: stress_sweep.py [n_shapes] [seed] [B] [ext|shared]      (GPU box; exit code 1 when a check fails)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
mode = sys.argv[4] if len(sys.argv) > 4 else ""          # "ext": PSD / exponential / power cones too; "shared": shared-A templates (k_sa_fwd / k_sa_lsqr)
if mode == "shared": os.environ["CE_CONST_A"] = "1"
from stress_kit import sweep, sweep_shared
t0 = time.time()
fails, notes, _ = sweep_shared(n_shapes, seed0, B) if mode == "shared" else sweep(n_shapes, seed0, B, ext=(mode == "ext"))
print(f"{n_shapes} shapes (seed {seed0}, B {B}): {len(fails)} failed, {len(notes)} with notes (iteration counts / inaccurate statuses), {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
