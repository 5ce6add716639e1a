"""Generates tests/golden/*.npz  --  run from the repo root:  python tests/golden/make_golden.py

The reference (cvxpylayers -> diffcp 1.1.4 -> scs 3.2.9) cannot be imported in this image (cvxpy,
diffcp and scs are not installed and there is no network) and stores no golden vectors of its own, so these
fixtures are NOT outputs of the reference: they are outputs of the repository's CPU oracle
(oracle/cone_oracle.c, itself pinned on the reference's closed-form known answers in
tests/test_oracle_known_answers.py), frozen at tight tolerance (eps = 1e-10, where the optimum no longer depends
on the iteration path) so that (i) the oracle cannot drift silently and (ii) the GPU tests have a fixed target
that does not depend on the oracle being rebuilt on the GPU box.
Inputs are regenerated from the seeds by cvxpylayers_amd.problems.generate; only outputs are stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvxpylayers_amd import problems as P  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = {
    # name: (n, cones, B, seed)
    "metric_M": (50, {"z": 0, "l": 20, "q": [10] * 8}, 16, 0),
    "mixed_small": (10, {"z": 3, "l": 8, "q": [5, 4]}, 8, 3),
    "socp_C3": (100, {"z": 0, "l": 10, "q": [11] * 10}, 4, 2),
    "sdp_small": (6, {"z": 2, "l": 3, "q": [4], "s": [3]}, 8, 5),
    "sdp_C4_lite": (36, {"z": 6, "l": 0, "q": [], "s": [8]}, 4, 6),
    "expcone_mixed": (8, {"z": 2, "l": 4, "q": [4], "s": [], "ep": 3}, 8, 1),
    "expcone_logreg_shape": (40, {"z": 0, "l": 12, "q": [4], "s": [], "ep": 24}, 4, 0),
    "powcone_mixed": (8, {"z": 1, "l": 3, "q": [3], "s": [], "ep": 1, "p": [0.3, -0.6, 0.5]}, 8, 3),
}
# quadratic-objective cases: name -> (n, cones, B, seed); P = G G^T / n + 0.1 I from the same seed (stored in the fixture)
QP_CASES = {
    "qp_mixed": (10, {"z": 2, "l": 6, "q": [4]}, 8, 7),
}
ONLY_NEW = "--only-new" in sys.argv      # keep the fixtures already committed byte-identical


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, (n, cones, B, seed) in CASES.items():
        if ONLY_NEW and os.path.exists(os.path.join(here, name + ".npz")):
            continue
        A, b, c = P.generate(n, cones, B, seed=seed)
        r = oracle.solve_batch(A, b, c, cones, eps=1e-10, max_iters=200000)
        assert (r["status"] == 1).all(), (name, r["status"])
        rng = np.random.default_rng(seed + 100)
        dx = rng.standard_normal(r["x"].shape)
        dy = rng.standard_normal(r["y"].shape)
        g = oracle.adjoint_batch(A, b, c, cones, r["x"], r["y"], r["s"], dx, dy, mode="dense")
        np.savez_compressed(os.path.join(here, name + ".npz"), n=n, B=B, seed=seed,
                            z=cones.get("z", 0), l=cones.get("l", 0), q=np.asarray(cones.get("q", []), dtype=np.int64),
                            s=np.asarray(cones.get("s", []), dtype=np.int64), ep=int(cones.get("ep", 0)),
                            p=np.asarray(cones.get("p", []), dtype=np.float64),
                            x=r["x"], y=r["y"], sl=r["s"], dx=dx, dy=dy, dA=g["dA"], db=g["db"], dc=g["dc"])
        print(name, "iters", r["iters"].max(), "file", name + ".npz")
    for name, (n, cones, B, seed) in QP_CASES.items():
        if ONLY_NEW and os.path.exists(os.path.join(here, name + ".npz")):
            continue
        A, b, c = P.generate(n, cones, B, seed=seed)
        rng = np.random.default_rng(seed + 200)
        G = rng.standard_normal((B, n, n)); Pm = G @ G.transpose(0, 2, 1) / n + 0.1 * np.eye(n)
        r = oracle.solve_batch(A, b, c, cones, P=Pm, eps=1e-10, max_iters=200000)
        assert (r["status"] == 1).all(), (name, r["status"])
        dx = rng.standard_normal(r["x"].shape); dy = rng.standard_normal(r["y"].shape)
        g = oracle.adjoint_batch(A, b, c, cones, r["x"], r["y"], r["s"], dx, dy, P=Pm, mode="dense")
        np.savez_compressed(os.path.join(here, name + ".npz"), n=n, B=B, seed=seed, z=cones.get("z", 0), l=cones.get("l", 0),
                            q=np.asarray(cones.get("q", []), dtype=np.int64), s=np.asarray([], dtype=np.int64), ep=0, p=np.asarray([], dtype=np.float64),
                            P=Pm, x=r["x"], y=r["y"], sl=r["s"], dx=dx, dy=dy, dA=g["dA"], db=g["db"], dc=g["dc"], dP=g["dP"])
        print(name, "iters", r["iters"].max(), "file", name + ".npz")


if __name__ == "__main__":
    main()
