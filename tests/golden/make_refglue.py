"""Generates tests/golden/refglue_<case>.npz by running the REFERENCE's own frontend + DIFFCP plugin code (tests/ref_glue.py:
/root/reference/src executed unchanged, cvxpy / diffcp stubbed, diffcp's arithmetic done by the CPU oracle) on the seeded cases
of tests/ref_cases.py.  Needs /root/reference (build container only).      python tests/golden/make_refglue.py

Each file holds: the parameters, the layer outputs and the parameter gradients of loss = sum_k <out_k, weight_k>, and the tensors
that crossed the plugin boundary in that run (q_eval, A_eval, primal, dual and their gradients)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_cases  # noqa: E402
import ref_glue  # noqa: E402

if __name__ == "__main__":
    args = {**ref_cases.SOLVER_ARGS, "mode": "dense"}
    with ref_glue.reference_modules() as ns:
        for name, make in ref_cases.CASES.items():
            out = ref_glue.run_case(ns, make(), args)
            np.savez_compressed(os.path.join(HERE, f"refglue_{name}.npz"), **out)
            print(name, {k: v.shape for k, v in out.items() if k.startswith(("out", "grad"))})
