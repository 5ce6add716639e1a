import os
import sys


def _cpu_budget() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (the GPU box: 256 CPUs visible, a quota of 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


# BEFORE torch / the oracle load an OpenMP runtime: with 256 CPUs visible and a quota of 16, every parallel region (the oracle's, torch's CPU ops between GPU launches) starts 256 threads whose idle
# spinning burns the cgroup's quota and gets the MAIN thread throttled -- tests then run 10-25 x slower on some boxes (round 5: the same test 0.4 s on one box, 9.9 s on another; two passes of the suite
# stopped making progress for 10-20 minutes inside tests that interleave small CPU ops with GPU launches).  Thread counts = the budget, idle workers sleep instead of spinning.
os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_budget()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("MKL_NUM_THREADS", os.environ["OMP_NUM_THREADS"])
os.environ.setdefault("OPENBLAS_NUM_THREADS", os.environ["OMP_NUM_THREADS"])

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run via gpurun / driver round-end)")
