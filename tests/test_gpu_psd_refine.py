"""The PSD projection of the shared-A kernels, alone (C ABI: ce_ca_psd_mfma -> psd_project_refine, ce_psd_mfma.h): a sequence of symmetric
matrices that drift like ADMM iterates (large steps first, then smaller and smaller) is projected call after call with the eigenvector state
carried over, and every result is compared with numpy's eigh projection.  Covers the three regimes of the routine -- cold start (Jacobi
sweeps), warm Jacobi fall-back (perturbation larger than the eigenvalue gaps), refinement steps on the matrix cores -- plus repeated
eigenvalues, semidefinite and zero matrices, and PSD orders on both sides of the 16 x 16 MFMA tile."""
import ctypes as C

import numpy as np
import pytest
import torch

from cvxpylayers_amd import _lib, problems as P

pytestmark = pytest.mark.gpu


def _engine(k, n=3):
    from cvxpylayers_amd.interfaces.mi355_if import ConeEngine
    cones = {"z": 1, "l": 2, "q": [3], "s": [k]}
    tpl = P.dense_template(n, cones)
    return ConeEngine(tpl.indices, tpl.indptr, tpl.n, tpl.m, cones, torch.device("cuda", 0)), tpl


def _project(eng, tpl, S, Vst, warm):
    """S (B, k, k) symmetric -> Pi_PSD(S) through the kernel; Vst: the caller-owned eigenvector state."""
    B, k, _ = S.shape
    l = tpl.n + tpl.m + 1; lp = l + (l & 1)
    off = tpl.n + 1 + 2 + 3                                      # PSD block after zero / nonneg / SOC rows
    U = torch.zeros((B, lp), dtype=torch.float64, device="cuda")
    U[:, off:off + k * (k + 1) // 2] = torch.from_numpy(P.sym_to_svec(S)).cuda()
    active = torch.ones(B, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().ce_ca_psd_mfma(eng._h, B, lp, U.data_ptr(), Vst.data_ptr(), int(warm), active.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ce_ca_psd_mfma")
    torch.cuda.synchronize()
    return P.svec_to_sym(U[:, off:off + k * (k + 1) // 2].cpu().numpy(), k)


def _exact(S):
    w, V = np.linalg.eigh(S)
    return (V * np.maximum(w, 0.0)[:, None, :]) @ np.swapaxes(V, 1, 2)


@pytest.mark.parametrize("k", [2, 5, 13, 16, 20, 33])
def test_projection_tracks_a_drifting_sequence(k):
    rng = np.random.default_rng(k)
    eng, tpl = _engine(k)
    B = 48
    G = rng.standard_normal((B, k, k)); S = 0.5 * (G + np.swapaxes(G, 1, 2))
    Vst = torch.zeros((B, 1, k * k), dtype=torch.float64, device="cuda")
    worst = 0.0
    for step, pert in enumerate([None, 0.5, 0.2, 0.1, 0.03, 0.03, 0.01, 0.01, 3e-3, 1e-3, 1e-3, 1e-4, 1e-5, 1e-7, 0.0, 1e-3]):
        if pert is not None:
            G = rng.standard_normal((B, k, k)); S = S + pert * 0.5 * (G + np.swapaxes(G, 1, 2))
        X = _project(eng, tpl, S, Vst, warm=step > 0)
        err = np.abs(X - _exact(S)).max() / np.abs(S).max()
        worst = max(worst, err)
        assert err <= 2e-13, (k, step, pert, err)
        V = Vst.cpu().numpy().reshape(B, k, k)                  # the state stays an orthonormal eigenbasis
        assert np.abs(V @ np.swapaxes(V, 1, 2) - np.eye(k)).max() <= 1e-11, (k, step)


def test_projection_handles_repeated_zero_and_definite_spectra():
    k = 20
    rng = np.random.default_rng(0)
    eng, tpl = _engine(k)
    Q = np.linalg.qr(rng.standard_normal((6, k, k)))[0]
    spectra = [np.r_[np.ones(7), -np.ones(13)],                 # two clusters
               np.r_[np.full(5, 2.0), np.zeros(10), np.full(5, -3.0)],      # semidefinite part
               np.zeros(k),                                     # zero matrix
               np.linspace(0.1, 2.0, k),                        # positive definite: projection = identity map
               -np.linspace(0.1, 2.0, k),                       # negative definite: projection = 0
               np.r_[1.0, 1.0 + 1e-9, 1.0 - 1e-9, -np.linspace(0.5, 1, k - 3)]]   # nearly repeated
    S = np.stack([(Q[i] * w[None, :]) @ Q[i].T for i, w in enumerate(spectra)]); S = 0.5 * (S + np.swapaxes(S, 1, 2))
    Vst = torch.zeros((len(spectra), 1, k * k), dtype=torch.float64, device="cuda")
    for step, pert in enumerate([None, 1e-2, 1e-4, 1e-8, 0.0, 0.0]):
        if pert:
            G = rng.standard_normal(S.shape); S = S + pert * 0.5 * (G + np.swapaxes(G, 1, 2))
        X = _project(eng, tpl, S, Vst, warm=step > 0)
        assert np.abs(X - _exact(S)).max() <= 3e-13 * max(np.abs(S).max(), 1.0), (step, pert)
