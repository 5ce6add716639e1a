"""Constant-A forward path: A batch-invariant, only b and c vary over the batch.

This is the case of most "parameters in the right-hand side / objective" layers (control, portfolio: BASELINE config 5) and
the one the reference's Moreau plugin special-cases (moreau_if.py:234-256, PA_is_constant).  The three matrix products of
the SCS-style iteration then are the SAME matrix for every instance, so they are done for the whole batch as fp64 GEMMs on the
MFMA pipe (torch.matmul -> rocBLAS, 53 TFLOP/s measured for these shapes on MI355X):

    T   = rho_x W_x - W_y A^                 (B x m) . (m x n)
    P_x = ((T Q) * 1/(rho_x + scale_b Lam)) Q^T        A^^T D0 A^ = Q Lam Q^T factored ONCE per call, so each instance keeps
    Q_y = P_x A^^T                                      its own adaptive scale (D_y = scale_b * D0)

and the per-instance remainder (tau-tilde, cone projections, relaxed update, termination / adaptive-scale logic) runs in
the HIP kernels of csrc/ce_const_a.h through the C ABI (ce_ca_step / ce_ca_check / ce_ca_finish).  Algorithm, constants and
order of operations are those of the per-instance kernels and of oracle/cone_oracle.c (SCS 3 restated); converged instances are
frozen at their own iteration.  Cones: zero / nonnegative / second-order.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from cvxpylayers_amd import _lib

CONVERGED_INTERVAL = 25
NUM_RUIZ_PASSES, NUM_L2_PASSES = 25, 1
MIN_SCALE, MAX_SCALE = 1e-4, 1e4
TAU_FACTOR, ZERO_CONE_FACTOR = 10.0, 1000.0


def _clamp_scale(v: torch.Tensor) -> torch.Tensor:
    return torch.where(v < MIN_SCALE, torch.ones_like(v), torch.clamp(v, max=MAX_SCALE))


def is_constant_A(A_bm: torch.Tensor, nnzA: int) -> bool:
    """True when the A part of the value rows is identical for every instance (one pass over the values, one sync)."""
    if A_bm.shape[0] == 1:
        return False
    return bool((A_bm[:, :nnzA] == A_bm[0:1, :nnzA]).all().item())


def solve_const_a(eng, A_bm: torch.Tensor, q_eval: torch.Tensor, settings):
    """eng: ConeEngine; A_bm (B, nnz_aug) batch-major values of [A_cvx | b_cvx]; q_eval (n+1, B).  Returns x, y, s, iters, status, resid."""
    L = _lib.lib()
    dev = A_bm.device
    n, m = eng.n, eng.m
    B = A_bm.shape[0]
    l = n + m + 1
    lp = l + (l & 1)                         # even row pitch
    indices, indptr = eng._indices, eng._indptr
    nnzA = int(indptr[n])
    cols = np.repeat(np.arange(n + 1), np.diff(indptr))
    f64 = dict(dtype=torch.float64, device=dev)
    # ---- dense A (solver form: A = -A_cvx), b (B, m), c (B, n)
    A = torch.zeros((m, n), **f64)
    A[torch.from_numpy(indices[:nnzA].astype(np.int64)).to(dev), torch.from_numpy(cols[:nnzA].astype(np.int64)).to(dev)] = -A_bm[0, :nnzA]
    b = torch.zeros((B, m), **f64)
    if eng.nnz_aug > nnzA:
        b[:, torch.from_numpy(indices[nnzA:].astype(np.int64)).to(dev)] = A_bm[:, nnzA:]
    c = q_eval[:n].t().to(torch.float64).contiguous()
    cone = eng.cone_dict
    z, nl, qs = int(cone.get("z", 0)), int(cone.get("l", 0)), [int(v) for v in cone.get("q", [])]
    # ---- equilibration of the one shared matrix (25 Ruiz passes + 1 l2 pass, row scalings averaged inside SOC blocks)
    D = torch.ones(m, **f64); E = torch.ones(n, **f64)
    if settings.normalize:
        blk = torch.full((m,), -1, dtype=torch.int64, device=dev)
        off = z + nl
        for k, d in enumerate(qs):
            blk[off:off + d] = k
            off += d
        soc_rows = (blk >= 0).nonzero().flatten()
        cnt = torch.tensor(qs, **f64) if qs else None
        for p in range(NUM_RUIZ_PASSES + NUM_L2_PASSES):
            if p >= NUM_RUIZ_PASSES:
                Dt, Et = A.norm(dim=1), A.norm(dim=0)
            else:
                Dt, Et = A.abs().amax(dim=1), A.abs().amax(dim=0)
            if qs:
                avg = torch.zeros(len(qs), **f64).index_add_(0, blk[soc_rows], Dt[soc_rows]) / cnt
                Dt = Dt.clone(); Dt[soc_rows] = avg[blk[soc_rows]]
            Dt = 1.0 / torch.sqrt(_clamp_scale(Dt)); Et = 1.0 / torch.sqrt(_clamp_scale(Et))
            A = Dt[:, None] * A * Et[None, :]
            D = D * Dt; E = E * Et
    At = A.t().contiguous()
    nrm_b0 = b.abs().amax(dim=1) if m else torch.zeros(B, **f64)
    nrm_c0 = c.abs().amax(dim=1)
    bh = b * D; ch = c * E
    if settings.normalize:
        sigma = 1.0 / _clamp_scale(torch.maximum(bh.abs().amax(dim=1), ch.abs().amax(dim=1)))
    else:
        sigma = torch.ones(B, **f64)
    bh = (bh * sigma[:, None]).contiguous(); ch = (ch * sigma[:, None]).contiguous()
    # ---- one eigendecomposition serves every instance and every rescale:  A^T D0 A = Q Lam Q^T
    d0 = torch.ones(m, **f64); d0[:z] = ZERO_CONE_FACTOR
    lam, Q = torch.linalg.eigh(At @ (d0[:, None] * A))
    lam = torch.clamp(lam, min=0.0)
    Qt = Q.t().contiguous()
    rho_x, alpha = float(settings.rho_x), float(settings.alpha)
    scale = torch.full((B,), float(settings.scale), **f64)
    state = {}

    def refresh():
        """g = (R + M)^{-1} h, h.g, phi and the per-instance diagonal of G for the current scales."""
        dyv = scale[:, None] * d0[None, :]
        a = (dyv * bh) @ A
        Dinv = 1.0 / (rho_x + scale[:, None] * lam[None, :])
        gx = (((ch - a) @ Q) * Dinv) @ Qt
        pk = (((ch + a) @ Q) * Dinv) @ Qt
        gy = dyv * (gx @ At + bh)
        hg = (ch * gx).sum(dim=1) + (bh * gy).sum(dim=1)
        Bc = scale.shape[0]
        G = torch.zeros((Bc, lp), **f64); PHI = torch.zeros((Bc, lp), **f64)
        G[:, :n] = gx; G[:, n:n + m] = gy
        PHI[:, :n] = rho_x * pk; PHI[:, n:n + m] = bh - pk @ At
        state.update(Dinv=Dinv, G=G, PHI=PHI, inv_den=(1.0 / (TAU_FACTOR + hg)).contiguous())

    # The working set is COMPACTED as instances finish: rows of converged instances are written back and dropped once fewer
    # than half of the current rows are active, so the tail of slow instances does not pay for GEMMs over the whole batch.
    B0 = B
    out_x = torch.empty((B0, n), **f64); out_y = torch.empty((B0, m), **f64); out_s = torch.empty((B0, m), **f64)
    out_iters = torch.zeros(B0, dtype=torch.int32, device=dev); out_status = torch.zeros(B0, dtype=torch.int32, device=dev)
    out_resid = torch.full((B0, 3), float("nan"), **f64)
    rows = torch.arange(B0, device=dev)              # original index of every current row
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    h = eng._h
    max_iters = int(settings.max_iters)

    refresh()
    W = torch.zeros((B, lp), **f64); W[:, l - 1] = 1.0
    UT = torch.zeros((B, lp), **f64); U = torch.zeros((B, lp), **f64)
    active = torch.ones(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    resid = torch.full((B, 3), float("nan"), **f64)
    sum_log = torch.zeros(B, **f64)
    n_log = torch.zeros(B, dtype=torch.int32, device=dev)
    last_sc = torch.zeros(B, dtype=torch.int32, device=dev)
    rescaled = torch.zeros(B, dtype=torch.int32, device=dev)

    def write_back(sel_rows=None):
        """un-normalise the current rows (ce_ca_finish) and store those selected (default: all) at their original positions"""
        Bc = W.shape[0]
        x = torch.empty((Bc, n), **f64); y = torch.empty((Bc, m), **f64); sv = torch.empty((Bc, m), **f64)
        st, itc = status.clone(), iters.clone()
        _lib.check(L.ce_ca_finish(h, Bc, lp, max_iters, W.data_ptr(), UT.data_ptr(), U.data_ptr(), D.data_ptr(), E.data_ptr(), bh.data_ptr(),
                                  ch.data_ptr(), sigma.data_ptr(), scale.data_ptr(), active.data_ptr(), st.data_ptr(), itc.data_ptr(),
                                  x.data_ptr(), y.data_ptr(), sv.data_ptr(), stream), "ce_ca_finish")
        k = slice(None) if sel_rows is None else sel_rows
        dst = rows[k]
        out_x[dst] = x[k]; out_y[dst] = y[k]; out_s[dst] = sv[k]; out_iters[dst] = itc[k]; out_status[dst] = st[k]; out_resid[dst] = resid[k]

    for it in range(max_iters):
        check = (it % CONVERGED_INTERVAL) == 0
        last = it + 1 >= max_iters
        Bc = W.shape[0]
        Wx, Wy = W[:, :n], W[:, n:n + m]
        T = torch.addmm(Wx, Wy, A, beta=rho_x, alpha=-1.0)             # rho_x w_x - A^T w_y      (B, n)
        PX = ((T @ Q) * state["Dinv"]) @ Qt                              # p_x = G_b t
        QY = PX @ At                                                     # A p_x                    (B, m)
        _lib.check(L.ce_ca_step(h, Bc, lp, W.data_ptr(), UT.data_ptr(), U.data_ptr(), PX.data_ptr(), PX.stride(0), QY.data_ptr(),
                                QY.stride(0), state["G"].data_ptr(), state["PHI"].data_ptr(), scale.data_ptr(),
                                state["inv_den"].data_ptr(), active.data_ptr(), int(not (check or last)),
                                int(((it + 1) % CONVERGED_INTERVAL) == 0), alpha, stream), "ce_ca_step")
        if check:
            AX = U[:, :n] @ At
            ATY = U[:, n:n + m] @ A
            _lib.check(L.ce_ca_check(h, Bc, lp, it, C.byref(settings), W.data_ptr(), UT.data_ptr(), U.data_ptr(), AX.data_ptr(), AX.stride(0),
                                     ATY.data_ptr(), ATY.stride(0), D.data_ptr(), E.data_ptr(), bh.data_ptr(), ch.data_ptr(),
                                     sigma.data_ptr(), nrm_b0.data_ptr(), nrm_c0.data_ptr(), scale.data_ptr(), sum_log.data_ptr(),
                                     n_log.data_ptr(), last_sc.data_ptr(), active.data_ptr(), status.data_ptr(), iters.data_ptr(),
                                     resid.data_ptr(), rescaled.data_ptr(), stream), "ce_ca_check")
            n_active, n_resc = torch.stack([active.sum(), rescaled.sum()]).tolist()      # the one host sync per 25 iterations
            if n_active == 0:
                break
            if n_resc:
                refresh()
                rescaled.zero_()
            if 2 * n_active <= Bc and Bc > 64:          # compact
                done_rows = (active == 0).nonzero().flatten()
                keep = (active != 0).nonzero().flatten()
                write_back(done_rows)
                rows = rows[keep]
                W, UT, U = W[keep].contiguous(), UT[keep].contiguous(), U[keep].contiguous()
                bh, ch, sigma, nrm_b0, nrm_c0 = bh[keep].contiguous(), ch[keep].contiguous(), sigma[keep].contiguous(), nrm_b0[keep].contiguous(), nrm_c0[keep].contiguous()
                scale, sum_log, n_log, last_sc = scale[keep].contiguous(), sum_log[keep].contiguous(), n_log[keep].contiguous(), last_sc[keep].contiguous()
                active, status, iters, resid, rescaled = active[keep].contiguous(), status[keep].contiguous(), iters[keep].contiguous(), resid[keep].contiguous(), rescaled[keep].contiguous()
                for key in ("Dinv", "G", "PHI", "inv_den"):
                    state[key] = state[key][keep].contiguous()
    write_back()
    return out_x, out_y, out_s, out_iters, out_status, out_resid
