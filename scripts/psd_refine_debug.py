"""GPU debug: accuracy of the PSD projection per step of a drifting sequence, refinement vs Jacobi-only (warm & 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_psd_refine as T
for k in (2, 20, 33):
    for mode in (1, 3):
        rng = np.random.default_rng(k)
        eng, tpl = T._engine(k)
        B = 48
        G = rng.standard_normal((B, k, k)); S = 0.5 * (G + np.swapaxes(G, 1, 2))
        Vst = torch.zeros((B, 1, k * k), dtype=torch.float64, device="cuda")
        errs = []
        for step, pert in enumerate([None, 0.5, 0.2, 0.1, 0.03, 0.03, 0.01, 0.01, 3e-3, 1e-3, 1e-3, 1e-4, 1e-5, 1e-7, 0.0, 1e-3]):
            if pert is not None:
                G = rng.standard_normal((B, k, k)); S = S + pert * 0.5 * (G + np.swapaxes(G, 1, 2))
            X = T._project(eng, tpl, S, Vst, warm=(mode if step > 0 else 0))
            e = np.abs(X - T._exact(S)).reshape(B, -1).max(1) / np.abs(S).max()
            errs.append("%.1e" % e.max())
        print("k", k, "mode", "refine" if mode == 1 else "jacobi", " ".join(errs))
