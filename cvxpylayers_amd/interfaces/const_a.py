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


class _PsdBlock:
    """svec <-> symmetric matrix for one PSD block (lower triangle, column-major, sqrt(2) off-diagonals; torch/cvxpylayer.py:201-222)."""

    def __init__(self, k: int, off: int, dev):
        self.k, self.off, self.d = k, off, k * (k + 1) // 2
        ri, cj = [], []
        for j in range(k):
            for i in range(j, k):
                ri.append(i); cj.append(j)
        self.ri = torch.tensor(ri, device=dev); self.cj = torch.tensor(cj, device=dev)
        diag = self.ri == self.cj
        one = torch.ones(len(ri), dtype=torch.float64, device=dev)      # (python scalars in torch.where would make float32 constants)
        self.to_mat = torch.where(diag, one, one * (2.0 ** -0.5))
        self.to_vec = torch.where(diag, one, one * (2.0 ** 0.5))

    def smat(self, v):
        S = torch.zeros(v.shape[0], self.k, self.k, dtype=torch.float64, device=v.device)
        val = v * self.to_mat
        S[:, self.ri, self.cj] = val
        S[:, self.cj, self.ri] = val
        return S

    def svec(self, S):
        return S[:, self.ri, self.cj] * self.to_vec


def _psd_blocks(cone, dev):
    z, nl, qs = int(cone.get("z", 0)), int(cone.get("l", 0)), [int(v) for v in cone.get("q", [])]
    off = z + nl + sum(qs)
    out = []
    for k in [int(v) for v in cone.get("s", [])]:
        out.append(_PsdBlock(k, off, dev)); off += k * (k + 1) // 2
    return out


def is_constant_A(A_bm: torch.Tensor, nnzA: int) -> bool:
    """True when the A part of the value rows is identical for every instance (one pass over the values, one sync)."""
    if A_bm.shape[0] == 1:
        return False
    return bool((A_bm[:, :nnzA] == A_bm[0:1, :nnzA]).all().item())


def solve_const_a(eng, A_bm: torch.Tensor, q_eval: torch.Tensor, settings, warm=None):
    """eng: ConeEngine; A_bm (B, nnz_aug) batch-major values of [A_cvx | b_cvx]; q_eval (n+1, B).  Returns x, y, s, iters, status, resid."""
    import os, time
    _timing = os.environ.get("CE_CA_TIMING") == "1"
    def _tick(tag, _t=[None]):
        if _timing:
            torch.cuda.synchronize(); now = time.perf_counter()
            if _t[0] is not None: print(f"[const_a] {tag}: {(now - _t[0]) * 1e3:.1f} ms")
            _t[0] = now
    _tick("start")
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
    b = torch.zeros((B, m), **f64)
    if eng.nnz_aug > nnzA:
        b[:, torch.from_numpy(indices[nnzA:].astype(np.int64)).to(dev)] = A_bm[:, nnzA:]
    c = q_eval[:n].t().to(torch.float64).contiguous()
    cone = eng.cone_dict
    z, nl, qs = int(cone.get("z", 0)), int(cone.get("l", 0)), [int(v) for v in cone.get("q", [])]
    psd = _psd_blocks(cone, dev)
    ntri = int(cone.get("ep", 0)) + len(cone.get("p", []))       # exponential / power cone triples (after the PSD blocks)
    # Everything derived from the shared matrix alone (its equilibration: 26 passes of small torch kernels, ~4 ms of launches) is kept on the
    # engine and reused while the caller keeps handing over the same values -- the usual case: A is a constant of the layer.
    A_vals0 = A_bm[0, :nnzA]
    cache = getattr(eng, "_ca_cache", None)
    if cache is not None and cache["normalize"] == bool(settings.normalize) and cache["A0"].shape == A_vals0.shape and torch.equal(cache["A0"], A_vals0):
        A, D, E = cache["A"], cache["D"], cache["E"]
    else:
        cache = None
        A = torch.zeros((m, n), **f64)
        A[torch.from_numpy(indices[:nnzA].astype(np.int64)).to(dev), torch.from_numpy(cols[:nnzA].astype(np.int64)).to(dev)] = -A_vals0
        # ---- equilibration of the one shared matrix (25 Ruiz passes + 1 l2 pass, row scalings averaged inside SOC blocks)
        D = torch.ones(m, **f64); E = torch.ones(n, **f64)
        if settings.normalize:
            blk = torch.full((m,), -1, dtype=torch.int64, device=dev)
            off = z + nl
            blocks = qs + [pb.d for pb in psd] + [3] * ntri      # row scalings are averaged inside SOC / PSD blocks and exp / power triples alike
            for k, d in enumerate(blocks):
                blk[off:off + d] = k
                off += d
            soc_rows = (blk >= 0).nonzero().flatten()
            cnt = torch.tensor(blocks, **f64) if blocks else None
            for p in range(NUM_RUIZ_PASSES + NUM_L2_PASSES):
                if p >= NUM_RUIZ_PASSES:
                    Dt, Et = A.norm(dim=1), A.norm(dim=0)
                else:
                    Dt, Et = A.abs().amax(dim=1), A.abs().amax(dim=0)
                if blocks:
                    avg = torch.zeros(len(blocks), **f64).index_add_(0, blk[soc_rows], Dt[soc_rows]) / cnt
                    Dt = Dt.clone(); Dt[soc_rows] = avg[blk[soc_rows]]
                Dt = 1.0 / torch.sqrt(_clamp_scale(Dt)); Et = 1.0 / torch.sqrt(_clamp_scale(Et))
                A = Dt[:, None] * A * Et[None, :]
                D = D * Dt; E = E * Et
        eng._ca_cache = dict(A0=A_vals0.clone(), normalize=bool(settings.normalize), A=A, D=D, E=E)
    _tick("extract + equilibrate")
    At = A.t().contiguous()
    nrm_b0 = b.abs().amax(dim=1) if m else torch.zeros(B, **f64)
    nrm_c0 = c.abs().amax(dim=1)
    bh = b * D; ch = c * E
    if settings.normalize:
        sigma = 1.0 / _clamp_scale(torch.maximum(bh.abs().amax(dim=1), ch.abs().amax(dim=1)))
    else:
        sigma = torch.ones(B, **f64)
    bh = (bh * sigma[:, None]).contiguous(); ch = (ch * sigma[:, None]).contiguous()
    # ---- persistent one-kernel path (ce_shared_a_fwd.h): A = (rows with one entry) + (r <= 64 rows with several): the reduced KKT matrix
    #      is diagonal + rank r and is applied by the Woodbury identity; every iterate of an instance stays in LDS, no host round trips.
    #      Taken whenever the template has that shape, the Woodbury form is stable and the iterates fit LDS (measured against the batch-GEMM
    #      path below: 28 vs 40 ms at BASELINE config 4, 420 vs 480 ms at config 5 with B = 16384); CE_SA_FWD=0 disables.
    _saf = os.environ.get("CE_SA_FWD")
    if _saf != "0":
        split = eng._ca_cache.get("split")
        if split is None:
            row_nnz = np.bincount(indices[:nnzA], minlength=m)
            drows_np = np.nonzero(row_nnz >= 2)[0].astype(np.int32)
            r_d = int(len(drows_np))
            RP = 16 if r_d <= 16 else (32 if r_d <= 32 else 64)
            split = dict(r_d=r_d, RP=RP, stable=False)
            if r_d <= 64:
                srow = (row_nnz == 1)
                ent_rows = indices[:nnzA].astype(np.int64); ent_cols = cols[:nnzA].astype(np.int64)
                sing = srow[ent_rows]                                          # structural entries that sit in single-entry rows
                srow_col_np = np.full(m, -1, dtype=np.int32); srow_col_np[drows_np] = -2 - np.arange(r_d, dtype=np.int32); srow_col_np[ent_rows[sing]] = ent_cols[sing]      # >= 0 column of a singleton row, -2 - a: dense row in slot a, -1 empty row
                order = np.argsort(ent_cols[sing], kind="stable")
                scol_row_np = ent_rows[sing][order].astype(np.int32)
                scol_ptr_np = np.concatenate([[0], np.cumsum(np.bincount(ent_cols[sing], minlength=n))]).astype(np.int32)
                ti32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
                drow_t, srow_col_t, scol_ptr_t, scol_row_t = ti32(drows_np if r_d else np.zeros(1)), ti32(srow_col_np), ti32(scol_ptr_np), ti32(scol_row_np if len(scol_row_np) else np.zeros(1))
                srow_val = torch.zeros(m, **f64)
                if sing.any():
                    rs_t = torch.from_numpy(ent_rows[sing]).to(dev); cs_t = torch.from_numpy(ent_cols[sing]).to(dev)
                    srow_val[rs_t] = A[rs_t, cs_t]
                AdT = torch.zeros((n, RP), **f64)
                if r_d:
                    AdT[:, :r_d] = A[torch.from_numpy(drows_np.astype(np.int64)).to(dev), :].t()
                d0s = torch.ones(m, **f64); d0s[:z] = ZERO_CONE_FACTOR
                gs = torch.zeros(n, **f64)
                if sing.any():
                    gs.index_add_(0, cs_t, d0s[rs_t] * srow_val[rs_t] ** 2)
                # Woodbury is only stable when the diagonal part carries weight in EVERY column (each variable sits in some single-entry row:
                # bounds, identity blocks); a column without one has Dg_j = rho_x = 1e-6 and the formula cancels catastrophically
                split.update(stable=bool((gs.min() >= 1e-2).item()) if n else False, drow_t=drow_t, srow_col_t=srow_col_t, scol_ptr_t=scol_ptr_t,
                             scol_row_t=scol_row_t, srow_val=srow_val, AdT=AdT, gs=gs)
            eng._ca_cache["split"] = split
        r_d, RP, stable = split["r_d"], split["RP"], split["stable"]
        if r_d <= 64 and stable:
            drow_t, srow_col_t, scol_ptr_t, scol_row_t, srow_val, AdT, gs = (split[k] for k in ("drow_t", "srow_col_t", "scol_ptr_t", "scol_row_t", "srow_val", "AdT", "gs"))
        if r_d <= 64 and stable:
            xo = torch.empty((B, n), **f64); yo = torch.empty((B, m), **f64); so = torch.empty((B, m), **f64)
            it_o = torch.empty(B, dtype=torch.int32, device=dev); st_o = torch.empty(B, dtype=torch.int32, device=dev); rs_o = torch.empty((B, 3), **f64)
            wx = wy = ws = None
            settings.warm_start = 0
            if warm is not None:
                wx, wy, ws = (t.detach().to(device=dev, dtype=torch.float64).contiguous() for t in warm)
                settings.warm_start = 1
            ptr = lambda t: t.data_ptr() if t is not None else None
            rc = L.ce_solve_shared_a(eng._h, B, r_d, RP, AdT.data_ptr(), drow_t.data_ptr(), srow_col_t.data_ptr(), srow_val.data_ptr(), scol_ptr_t.data_ptr(),
                                     scol_row_t.data_ptr(), gs.data_ptr(), D.data_ptr(), E.data_ptr(), bh.data_ptr(), ch.data_ptr(), sigma.data_ptr(),
                                     nrm_b0.data_ptr(), nrm_c0.data_ptr(), C.byref(settings), ptr(wx), ptr(wy), ptr(ws), xo.data_ptr(), yo.data_ptr(),
                                     so.data_ptr(), it_o.data_ptr(), st_o.data_ptr(), rs_o.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if rc == 0:
                eng._note_acceleration(settings, honoured=True, path="shared-A forward kernel")     # k_sa_fwd implements it (one-pair history, like k_fwd2)
                eng.last_const_a_kernel = "k_sa_fwd"
                _tick("k_sa_fwd (enqueued)")
                return xo, yo, so, it_o, st_o, rs_o
            if rc not in (-2, -3):
                _lib.check(rc, "ce_solve_shared_a")
    eng.last_const_a_kernel = "batch GEMM"
    eng._note_acceleration(settings, honoured=False, path="constant-A batch-GEMM path")
    # ---- one eigendecomposition serves every instance and every rescale:  A^T D0 A = Q Lam Q^T
    d0 = torch.ones(m, **f64); d0[:z] = ZERO_CONE_FACTOR
    lam, Q = torch.linalg.eigh(At @ (d0[:, None] * A))
    lam = torch.clamp(lam, min=0.0)
    Qt = Q.t().contiguous()
    _tick("eigh")
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
        new = dict(Dinv=Dinv, G=G, PHI=PHI, inv_den=(1.0 / (TAU_FACTOR + hg)).contiguous())
        for key, val in new.items():      # in place when possible: a captured HIP graph keeps pointing at these buffers
            if key in state and state[key].shape == val.shape:
                state[key].copy_(val)
            else:
                state[key] = val

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
    if warm is not None:       # SCS warm start u = (x^, y^, 1), v = (0, s^, 0); the iteration map's fixed point has w = u + R^-1 v
        wx0, wy0, ws0 = (t.to(device=dev, dtype=torch.float64) for t in warm)
        Wx = sigma[:, None] * wx0 / E[None, :]
        Wy = sigma[:, None] * wy0 / D[None, :] + sigma[:, None] * D[None, :] * ws0 * (scale[:, None] * d0[None, :])
        ok = (torch.isfinite(Wx).all(dim=1) & torch.isfinite(Wy).all(dim=1))[:, None]
        W[:, :n] = torch.where(ok, Wx, W[:, :n]); W[:, n:n + m] = torch.where(ok, Wy, W[:, n:n + m])
    UT = torch.zeros((B, lp), **f64); U = torch.zeros((B, lp), **f64)
    roots = torch.zeros((B, max(ntri, 1)), **f64)        # per-cone root of the previous iteration (exp / power projections)
    maxs = max([pb.k for pb in psd], default=0)
    use_mfma_psd = bool(psd) and maxs <= 39 and os.environ.get("CE_PSD_MFMA", "1") != "0"
    psdV = torch.zeros((B, max(len(psd), 1), max(maxs * maxs, 1)), **f64)        # eigenvectors of every PSD block (state of ce_ca_psd_mfma)
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

    def iteration(it, strm):
        """GEMMs + elementwise kernels of iteration `it` (everything but the check), enqueued on stream handle `strm`"""
        check = (it % CONVERGED_INTERVAL) == 0
        last = it + 1 >= max_iters
        Bc = W.shape[0]
        T = torch.addmm(W[:, :n], W[:, n:n + m], A, beta=rho_x, alpha=-1.0)      # rho_x w_x - A^T w_y      (B, n)
        PX = ((T @ Q) * state["Dinv"]) @ Qt                                        # p_x = G_b t
        QY = PX @ At                                                               # A p_x                    (B, m)
        norm_after = int(((it + 1) % CONVERGED_INTERVAL) == 0)
        _lib.check(L.ce_ca_step(h, Bc, lp, W.data_ptr(), UT.data_ptr(), U.data_ptr(), PX.data_ptr(), PX.stride(0), QY.data_ptr(),
                                QY.stride(0), state["G"].data_ptr(), state["PHI"].data_ptr(), scale.data_ptr(),
                                state["inv_den"].data_ptr(), active.data_ptr(), int(not (check or last) and not psd and not ntri),
                                norm_after, alpha, strm), "ce_ca_step")
        if psd or ntri:
            # PSD blocks: the step kernel leaves the cone input in U; project it in place (ce_ca_psd: workgroup-parallel Jacobi,
            # ~600x faster than batched rocSOLVER eigh at 20x20), then the relaxed update / renormalisation the kernel skipped
            if psd:
                if use_mfma_psd:      # matrix cores: eigen-refinement warm-started from the previous iteration's eigenvectors (self-correcting
                    #                   orthogonality: no restart at check iterations any more)
                    _lib.check(L.ce_ca_psd_mfma(h, Bc, lp, U.data_ptr(), psdV.data_ptr(), int(it > 0), active.data_ptr(), strm), "ce_ca_psd_mfma")
                else:
                    _lib.check(L.ce_ca_psd(h, Bc, lp, U.data_ptr(), active.data_ptr(), strm), "ce_ca_psd")
            if ntri:
                _lib.check(L.ce_ca_triples(h, Bc, lp, U.data_ptr(), roots.data_ptr(), active.data_ptr(), strm), "ce_ca_triples")
            if not (check or last):
                _lib.check(L.ce_ca_update(h, Bc, lp, W.data_ptr(), UT.data_ptr(), U.data_ptr(), active.data_ptr(), norm_after, alpha, strm), "ce_ca_update")

    # The 24 iterations between two checks are always the same launch sequence on the same buffers: they are captured once in a
    # HIP graph and replayed (the eager Python loop is host-bound for small batches: ~10 launches per iteration).  The graph is
    # re-captured only when a compaction replaces the buffers.
    _tick("refresh + state")
    use_graph = os.environ.get("CE_CA_GRAPH", "1") != "0"
    graph = None
    it = 0
    while it < max_iters:
        check = (it % CONVERGED_INTERVAL) == 0
        if not check:
            nblk = CONVERGED_INTERVAL - (it % CONVERGED_INTERVAL)            # iterations up to (not including) the next check
            if use_graph and (it % CONVERGED_INTERVAL) == 1 and it + nblk < max_iters:
                if graph is None:
                    torch.cuda.synchronize(dev)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        cs_ = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                        for j in range(nblk):
                            iteration(1 + j, cs_)
                graph.replay()                                 # (capture records, it does not execute)
                it += nblk
                continue
            iteration(it, stream)
            it += 1
            continue
        last = it + 1 >= max_iters
        Bc = W.shape[0]
        iteration(it, stream)
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
                W, UT, U, roots, psdV = W[keep].contiguous(), UT[keep].contiguous(), U[keep].contiguous(), roots[keep].contiguous(), psdV[keep].contiguous()
                bh, ch, sigma, nrm_b0, nrm_c0 = bh[keep].contiguous(), ch[keep].contiguous(), sigma[keep].contiguous(), nrm_b0[keep].contiguous(), nrm_c0[keep].contiguous()
                scale, sum_log, n_log, last_sc = scale[keep].contiguous(), sum_log[keep].contiguous(), n_log[keep].contiguous(), last_sc[keep].contiguous()
                active, status, iters, resid, rescaled = active[keep].contiguous(), status[keep].contiguous(), iters[keep].contiguous(), resid[keep].contiguous(), rescaled[keep].contiguous()
                for key in ("Dinv", "G", "PHI", "inv_den"):
                    state[key] = state[key][keep].contiguous()
                graph = None                                  # buffers replaced: capture again
        it += 1
    _tick("iterations")
    write_back()
    _tick("write back")
    return out_x, out_y, out_s, out_iters, out_status, out_resid


# ======================================================================================================================
# Adjoint for the constant-A path: batched LSQR on the reduced system  N r = (dx, DPi dy),  N(r_x, r_y) = (-A^T r_y, DPi(A r_x - r_y) + r_y)
# (the tau row / column of diffcp's M^T r = dz is dropped: dA, db, dc do not depend on the null component, r_tau = 0; this
# is the same reduced system the per-instance kernels eliminate directly).  diffcp itself solves its system with LSQR
# (mode="lsqr" is its default); here every operator application is a GEMM over the batch plus elementwise cone derivatives.
# ======================================================================================================================
def _dproj(v, h, z, nl, qs, psd=(), psd_eig=None, tri=None):
    """DPi_{K*}(v) h, blockwise: zero rows (dual cone free) -> h; nonneg -> h [v > 0]; SOC -> closed form (cone_oracle.c dproj_soc)."""
    out = h.clone()
    if nl:
        out[:, z:z + nl] = h[:, z:z + nl] * (v[:, z:z + nl] > 0)
    off = z + nl
    for d in qs:
        vb, hb = v[:, off:off + d], h[:, off:off + d]
        if d == 1:
            out[:, off] = hb[:, 0] * (vb[:, 0] >= 0)
        else:
            t, zz = vb[:, :1], vb[:, 1:]
            nz = zz.norm(dim=1, keepdim=True)
            zh = (zz * hb[:, 1:]).sum(dim=1, keepdim=True)
            nzs = torch.clamp(nz, min=1e-300)
            o0 = (nz * hb[:, :1] + zh) / (2 * nzs)
            oz = (zz * hb[:, :1] + (t + nz) * hb[:, 1:] - t * zz * zh / (nzs * nzs)) / (2 * nzs)
            mid = torch.cat([o0, oz], dim=1)
            inside = nz <= t
            polar = (nz <= -t) & ~inside
            out[:, off:off + d] = torch.where(inside, hb, torch.where(polar, torch.zeros_like(hb), mid))
        off += d
    for pb, (Uv, Bm) in zip(psd, psd_eig or ()):      # DPi(V)[H] = U (B o (U^T H U)) U^T
        Hm = pb.smat(h[:, pb.off:pb.off + pb.d])
        out[:, pb.off:pb.off + pb.d] = pb.svec(Uv @ (Bm * (Uv.transpose(1, 2) @ Hm @ Uv)) @ Uv.transpose(1, 2))
    if tri is not None:       # exponential / power triples: 3x3 Jacobians J (B, ntri, 3, 3) from ce_ca_triple_jac (symmetric)
        off0, J = tri
        ntri = J.shape[1]
        hb = h[:, off0:off0 + 3 * ntri].reshape(-1, ntri, 3, 1)
        out[:, off0:off0 + 3 * ntri] = (J @ hb).reshape(-1, 3 * ntri)
    return out


def _psd_eig(v, psd):
    """eigenvectors of smat(v_c) and the divided-difference matrix B (1: both positive, 0: both non-positive, l+/(l+ - l-): mixed)"""
    out = []
    for pb in psd:
        w, Uv = torch.linalg.eigh(pb.smat(v[:, pb.off:pb.off + pb.d]))
        wi, wj = w[:, :, None], w[:, None, :]
        pi, pj = torch.clamp(wi, min=0.0), torch.clamp(wj, min=0.0)
        den = wi - wj
        Bm = torch.where((wi > 0) & (wj > 0), torch.ones_like(den), torch.where((wi <= 0) & (wj <= 0), torch.zeros_like(den), (pi - pj) / torch.where(den == 0, torch.ones_like(den), den)))
        out.append((Uv, Bm))
    return out


def vjp_const_a(eng, A_bm, x, y, s, dx, dy, batch_minor_out=False, atol=1e-8, btol=1e-8, iter_lim=0, q_eval=None, conlim=1e8):
    """Returns dA (nnz_aug, B), dq (n+1, B), adj_status (B,) in the boundary convention (diffcp_if.py:91-92).
    atol / btol / iter_lim: LSQR's stopping rule; the defaults are diffcp's (1e-8, 1e-8, 2 N with N = n + m + 1: what diffcp_if.py:86 runs and
    oracle/cone_oracle.c:85,712 restates); the plugin forwards solver_args["lsqr_atol" / "lsqr_btol" / "lsqr_iter_lim"] (mi355_if.lsqr_rule).
    q_eval (n+1, B): the forward call's objective values.  With them the one-kernel LSQR solves diffcp's FULL (n + m + 1) system (tau row and column: b and c
    enter); without them r_tau is pinned to 0 -- the same gradients wherever the system is regular, a different minimum-norm element on degenerate faces."""
    if iter_lim <= 0:
        iter_lim = 2 * (eng.n + eng.m + 1)
    dev = A_bm.device
    n, m, B = eng.n, eng.m, A_bm.shape[0]
    import os as _os
    # The one-kernel LSQR (ce_shared_a.h).  With the singleton / dense-row split of A (at most 64 rows with several entries) its products are
    # balanced and it beats the batched implementation below at every size measured (BASELINE config 5, B = 16384: 0.26 s against 0.73 s);
    # without the split its CSR / CSC products serialise on the longest row and lose at very large batch x nnz (1.05 s there), hence the
    # work threshold for that case only.  CE_SA_KERNEL=1 / 0 forces / disables.
    _sa = _os.environ.get("CE_SA_KERNEL")
    _row_nnz = np.bincount(eng._indices[:eng.nnzA], minlength=m) if eng.nnzA else np.zeros(m, dtype=np.int64)
    _has_split = int((_row_nnz >= 2).sum()) <= 64
    if _sa != "0" and (_sa == "1" or _has_split or B * max(eng.nnzA, 1) <= (1 << 26)):
        # one kernel, one workgroup per instance (ce_shared_a.h); falls through to the batched torch implementation when the template
        # has exponential / power cones or the LSQR vectors of an instance do not fit LDS
        f64_ = dict(dtype=torch.float64, device=dev)
        dA_bm = torch.empty((B, eng.nnz_aug), **f64_); dq = torch.empty((n + 1, B), **f64_)
        adj = torch.empty((B,), dtype=torch.int32, device=dev); its = torch.empty((B,), dtype=torch.int32, device=dev)
        xc, yc, sc_, dxc, dyc = (t.to(torch.float64).contiguous() for t in (x, y, s, dx, dy))
        if q_eval is not None:
            qd = q_eval.detach().to(device=dev, dtype=torch.float64)
            q_args = (qd.data_ptr(), qd.stride(0), qd.stride(1))
        else:
            q_args = (None, 0, 0)
        assert A_bm.stride(1) == 1
        rc = _lib.lib().ce_vjp_shared_a(eng._h, B, A_bm.data_ptr(), A_bm.stride(0), *q_args, xc.data_ptr(), yc.data_ptr(), sc_.data_ptr(), dxc.data_ptr(), dyc.data_ptr(),
                                        dA_bm.data_ptr(), dq.data_ptr(), B, 1, adj.data_ptr(), its.data_ptr(), atol, btol, float(conlim), int(iter_lim),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc == 0:
            eng.last_lsqr_iters = its
            return (dA_bm.t().contiguous() if batch_minor_out else dA_bm.t()), dq, adj
        if rc not in (-2, -3):
            _lib.check(rc, "ce_vjp_shared_a")
    indices, indptr = eng._indices, eng._indptr
    nnzA = eng.nnzA
    cols_np = np.repeat(np.arange(n + 1), np.diff(indptr))
    f64 = dict(dtype=torch.float64, device=dev)
    rows_t = torch.from_numpy(indices.astype(np.int64)).to(dev)
    cols_t = torch.from_numpy(cols_np.astype(np.int64)).to(dev)
    A = torch.zeros((m, n), **f64)
    A[rows_t[:nnzA], cols_t[:nnzA]] = -A_bm[0, :nnzA]
    At = A.t().contiguous()
    cone = eng.cone_dict
    z, nl, qs = int(cone.get("z", 0)), int(cone.get("l", 0)), [int(v) for v in cone.get("q", [])]
    v = y - s
    psd = _psd_blocks(cone, dev)
    peig = _psd_eig(v, psd)
    ntri = int(cone.get("ep", 0)) + len(cone.get("p", []))
    tri = None
    if ntri:
        vc = v.contiguous()
        J = torch.empty((B, ntri, 3, 3), **f64)
        _lib.check(_lib.lib().ce_ca_triple_jac(eng._h, B, vc.data_ptr(), vc.stride(0), J.data_ptr(),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ce_ca_triple_jac")
        tri = (m - 3 * ntri, 0.5 * (J + J.transpose(2, 3)))

    # diffcp's FULL (n + m + 1) system when the call's q_eval is at hand (tau row and column: c from q_eval, b from the value rows), like the one-kernel LSQR above
    # (round 6, ADVICE: this fallback used to drop them and to stop without the conlim / machine-precision tests -- the answer depended on which path ran);
    # q_eval None: r_tau pinned to 0 (adjoint_system="reduced")
    TAU = q_eval is not None
    if TAU:
        cvec = q_eval.detach().to(**f64)[:n].t().contiguous()                       # (B, n)
        bvec = torch.zeros((B, m), **f64)
        if eng.nnz_aug > nnzA:
            bvec[:, rows_t[nnzA:]] = A_bm[:, nnzA:].to(torch.float64)
    else:
        cvec, bvec = torch.zeros((B, n), **f64), torch.zeros((B, m), **f64)

    def N(rx, ry, rt):          # (B,n),(B,m),(B,) -> same
        return (-(ry @ A) - cvec * rt[:, None], _dproj(v, rx @ At - bvec * rt[:, None] - ry, z, nl, qs, psd, peig, tri) + ry,
                (cvec * rx).sum(dim=1) + (bvec * ry).sum(dim=1))

    def NT(px, py, pt):
        q = _dproj(v, py, z, nl, qs, psd, peig, tri)
        return q @ A + cvec * pt[:, None], -(px @ At) + bvec * pt[:, None] - q + py, -(cvec * px).sum(dim=1) - (bvec * q).sum(dim=1)

    def nrm(a, b_, t_):
        return torch.sqrt((a * a).sum(dim=1) + (b_ * b_).sum(dim=1) + t_ * t_)

    x64, y64 = x.to(torch.float64), y.to(torch.float64)
    bx, by = dx.to(torch.float64), _dproj(v, dy.to(torch.float64), z, nl, qs, psd, peig, tri)
    bt = -((x64 * bx).sum(dim=1) + (y64 * dy.to(torch.float64)).sum(dim=1)) if TAU else torch.zeros(B, **f64)      # dz_tau = -(x.dx + y.dy)
    # LSQR (Paige & Saunders), batched; rows that met a stopping test are frozen
    bnorm = nrm(bx, by, bt)
    live = bnorm > 0
    safe = lambda t: torch.where(t > 0, t, torch.ones_like(t))
    beta = bnorm.clone()
    ux, uy, ut = bx / safe(beta)[:, None], by / safe(beta)[:, None], bt / safe(beta)
    vx, vy, vt = NT(ux, uy, ut)
    if not TAU:
        vt = torch.zeros_like(vt)
    alfa = nrm(vx, vy, vt)
    vx, vy, vt = vx / safe(alfa)[:, None], vy / safe(alfa)[:, None], vt / safe(alfa)
    wx, wy, wt = vx.clone(), vy.clone(), vt.clone()
    rx, ry, rt = torch.zeros_like(bx), torch.zeros_like(by), torch.zeros(B, **f64)
    rhobar, phibar = alfa.clone(), beta.clone()
    anorm = torch.zeros(B, **f64); ddnorm = torch.zeros(B, **f64); xxnorm = torch.zeros(B, **f64)
    zz = torch.zeros(B, **f64); cs2 = -torch.ones(B, **f64); sn2 = torch.zeros(B, **f64)
    live = live & (alfa * beta > 0)
    itn_lim = int(iter_lim)
    ctol = 1.0 / conlim if conlim and conlim > 0 else 0.0
    tau_on = 1.0 if TAU else 0.0

    def lsqr_iter(st):
        """one LSQR iteration as a pure function of the state tuple (so that a block of them can be captured in a HIP graph); stopping tests as in
        oracle/cone_oracle.c lsqr_MT / scipy: atol, btol, conlim and the three machine-precision tests"""
        ux, uy, ut, vx, vy, vt, wx, wy, wt, rx, ry, rt, alfa, rhobar, phibar, anorm, ddnorm, xxnorm, zz, cs2, sn2, live = st
        tx, ty, tt = N(vx, vy, vt)
        ux, uy, ut = tx - alfa[:, None] * ux, ty - alfa[:, None] * uy, tau_on * (tt - alfa * ut)
        beta = nrm(ux, uy, ut)
        ux, uy, ut = ux / safe(beta)[:, None], uy / safe(beta)[:, None], ut / safe(beta)
        anorm = torch.sqrt(anorm * anorm + alfa * alfa + beta * beta)
        tx, ty, tt = NT(ux, uy, ut)
        vx, vy, vt = tx - beta[:, None] * vx, ty - beta[:, None] * vy, tau_on * (tt - beta * vt)
        alfa = nrm(vx, vy, vt)
        vx, vy, vt = vx / safe(alfa)[:, None], vy / safe(alfa)[:, None], vt / safe(alfa)
        rho = torch.sqrt(rhobar * rhobar + beta * beta)
        cs, sn = rhobar / safe(rho), beta / safe(rho)
        theta = sn * alfa; rhobar = -cs * alfa; phi = cs * phibar; phibar = sn * phibar; tau = sn * phi
        t1, t2 = phi / safe(rho), -theta / safe(rho)
        lm = live.to(torch.float64)
        ddnorm = ddnorm + ((wx * wx).sum(dim=1) + (wy * wy).sum(dim=1) + wt * wt) / safe(rho * rho)
        rx = rx + (lm * t1)[:, None] * wx; ry = ry + (lm * t1)[:, None] * wy; rt = rt + lm * t1 * wt
        wx, wy, wt = vx + t2[:, None] * wx, vy + t2[:, None] * wy, vt + t2 * wt
        delta = sn2 * rho; gambar = -cs2 * rho; rhs = phi - delta * zz; zbar = rhs / safe(gambar.abs()) * torch.sign(gambar)
        xnorm = torch.sqrt(xxnorm + zbar * zbar)
        gamma = torch.sqrt(gambar * gambar + theta * theta); cs2 = gambar / safe(gamma); sn2 = theta / safe(gamma); zz = rhs / safe(gamma); xxnorm = xxnorm + zz * zz
        rnorm = phibar
        arnorm = alfa * tau.abs()
        test1 = rnorm / safe(bnorm); test2 = arnorm / (anorm * rnorm + 1e-300)
        test3 = 1.0 / (anorm * torch.sqrt(ddnorm) + 1e-300); tt1 = test1 / (1.0 + anorm * xnorm / safe(bnorm))
        rtol = btol + atol * anorm * xnorm / safe(bnorm)
        live = live & ~((test1 <= rtol) | (test2 <= atol) | (test3 <= ctol) | (1.0 + test3 <= 1.0) | (1.0 + test2 <= 1.0) | (1.0 + tt1 <= 1.0))
        return (ux, uy, ut, vx, vy, vt, wx, wy, wt, rx, ry, rt, alfa, rhobar, phibar, anorm, ddnorm, xxnorm, zz, cs2, sn2, live)

    state = [ux, uy, ut, vx, vy, vt, wx, wy, wt, rx, ry, rt, alfa, rhobar, phibar, anorm, ddnorm, xxnorm, zz, cs2, sn2, live]
    BLK = 16
    import os
    graph = None
    if os.environ.get("CE_CA_GRAPH", "1") != "0":
        # the loop is host-bound for small batches (~60 small launches per iteration): capture BLK iterations once and replay
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            st = tuple(state)
            for _ in range(BLK):
                st = lsqr_iter(st)
            for dst, src in zip(state, st):
                dst.copy_(src)
    itn = 0
    while itn < itn_lim:
        if graph is not None:
            graph.replay()
        else:
            st = tuple(state)
            for _ in range(BLK):
                st = lsqr_iter(st)
            state = list(st)
        itn += BLK
        if not bool(state[-1].any().item()):
            break
    rx, ry, rt, live = state[9], state[10], state[11], state[-1]
    adj = live.to(torch.int32)           # 1: LSQR hit its iteration limit for this instance
    # outputs: dA_ij = x_j r_y,i - y_i r_x,j ; db = y r_tau - r_y ; dc = x r_tau - r_x, packed as [-dA.data, db[b_idx]], [dc, 0]  (cone_oracle.c adjoint_one)
    K = eng.nnz_aug
    dA_bm = torch.empty((B, K), **f64)
    ra, ca = rows_t[:nnzA], cols_t[:nnzA]
    dA_bm[:, :nnzA] = -(x64[:, ca] * ry[:, ra] - y64[:, ra] * rx[:, ca])
    if K > nnzA:
        rb = rows_t[nnzA:]
        dA_bm[:, nnzA:] = y64[:, rb] * rt[:, None] - ry[:, rb]
    dq = torch.zeros((n + 1, B), **f64)
    dq[:n] = (x64 * rt[:, None] - rx).t()
    dA = dA_bm.t().contiguous() if batch_minor_out else dA_bm.t()
    return dA, dq, adj
