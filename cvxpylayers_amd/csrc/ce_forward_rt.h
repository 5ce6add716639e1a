// ce_forward_rt.h -- register-tiled forward kernel (the hot path for LDS-sized instances).
//
// One instance per 512-thread workgroup (8 wave64).  The operators of the ADMM iteration are distributed
// so that every product is a short dot product followed by a DPP butterfly inside a group of CH lanes
// (no LDS round trip for partial sums, no extra barrier):
//   at1[T1] : REGISTER column tile of A-hat.  thread (j = tid / CH1, c = tid % CH1) holds A[c + CH1*k][j]   -> A^T v
//   gt [TG] : REGISTER tile of G = (rho_x I + A^T Dy A)^{-1} (symmetric): same thread holds G[c + CH1*k][j] -> G v
//   A-hat rows stay in LDS (leading dimension = 4 mod 8 -> conflict-free interleaved reads):
//             thread (i = tid / CH2, c = tid % CH2) reads A[i][c + CH2*k]                                   -> A v
// Tiles and LDS vectors are zero padded, so the products need no predication.
// Ruiz equilibration runs on register tiles; Gauss-Jordan inversion of the reduced KKT matrix runs on the gt
// tile with one barrier per pivot.  4 workgroup barriers per iteration.
// All LDS vectors sit at COMPILE-TIME offsets (template VP) so that accesses are `ds_read ... offset:imm` from a
// handful of base registers, and rarely used solver scalars are parked in LDS: both keep the kernel at <=128 VGPRs
// (2 workgroups = 16 waves per CU).
// Algorithm identical to ce_forward_generic.h / oracle/cone_oracle.c (same iterates up to summation order).
#pragma once

constexpr int NT2 = 512;
constexpr int NW2 = NT2 / 64;
constexpr int SOC_SMALL = 32;   // cones up to this size: every row thread recomputes its cone's norm (no extra barrier)
constexpr int RT_NVEC = 14;
constexpr int RT_EXTRA = NW2 * 8 + NW2 + 16;   // red, wpart, scalars

// (dpp_mov, group_reduce, dpp_mov_rows, wave_reduce_dpp: ce_common.h)

template <int K, int NWV>
__device__ __forceinline__ void block_reduce_n(double (&v)[K], unsigned maxmask, double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = ((maxmask >> k) & 1u) ? wave_reduce_dpp<true>(v[k]) : wave_reduce_dpp<false>(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[wid * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double t[NWV];          // the waves' partial results, requested together (the plain chain waited for each of them: NWV LDS round trips in series behind the barrier)
#pragma unroll
        for (int w = 0; w < NWV; w++) t[w] = red[w * K + k];
        double a = t[0];
#pragma unroll
        for (int w = 1; w < NWV; w++) a = ((maxmask >> k) & 1u) ? fmax(a, t[w]) : a + t[w];
        v[k] = a;
    }
    __syncthreads();
}

// register tile . LDS vector, interleaved assignment: sum_k tile[k] * vec[CH*k]  (vec already offset by the lane's c)
template <int CH, int TT>
__device__ __forceinline__ double tile_dot(const double (&tile)[TT], const double *vec) {
    double a0 = 0, a1 = 0;
#pragma unroll
    for (int k = 0; k + 1 < TT; k += 2) {
        a0 = fma(tile[k], vec[CH * k], a0);
        a1 = fma(tile[k + 1], vec[CH * (k + 1)], a1);
    }
    if constexpr (TT & 1) a0 = fma(tile[TT - 1], vec[CH * (TT - 1)], a0);
    return group_reduce<CH, false>(a0 + a1);
}
// LDS row . LDS vector (row pads are zero).  Loads are issued in two batches ahead of their FMAs (latency hiding
// without holding 2*TT doubles live at once).
template <int CH, int TT>
__device__ __forceinline__ double row_dot(const double *row, const double *vec, bool valid) {
    double a0 = 0, a1 = 0;
    if (valid) {
        constexpr int H = (TT + 1) / 2;
        {
            double r[H], v[H];
#pragma unroll
            for (int k = 0; k < H; k++) { r[k] = row[CH * k]; v[k] = vec[CH * k]; }
#pragma unroll
            for (int k = 0; k < H; k++) { if (k & 1) a1 = fma(r[k], v[k], a1); else a0 = fma(r[k], v[k], a0); }
        }
        {
            double r[TT - H], v[TT - H];
#pragma unroll
            for (int k = 0; k < TT - H; k++) { r[k] = row[CH * (H + k)]; v[k] = vec[CH * (H + k)]; }
#pragma unroll
            for (int k = 0; k < TT - H; k++) { if (k & 1) a1 = fma(r[k], v[k], a1); else a0 = fma(r[k], v[k], a0); }
        }
    }
    return group_reduce<CH, false>(a0 + a1);
}

template <int CH1, int T1, int TG, int CH2, int T2, int VP, int WPE>
__global__ void __launch_bounds__(NT2, WPE)
k_forward_rt(DevT T, ce_settings S, const double *__restrict__ Avals, const double *__restrict__ qv, long sqk, long sqb,
             double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ so, int *__restrict__ iters_o,
             int *__restrict__ status_o, double *__restrict__ resid_o, double *aa_ws) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // ---- compile-time LDS layout (doubles)
    constexpr int O_BV = 0 * VP, O_CV = 1 * VP, O_DV = 2 * VP, O_EV = 3 * VP, O_G = 4 * VP, O_W = 5 * VP, O_UT = 6 * VP,
                  O_U = 7 * VP, O_ZB = 8 * VP, O_PHI = 9 * VP, O_TV = 10 * VP, O_PX = 11 * VP, O_S1 = 12 * VP, O_S2 = 13 * VP,
                  O_RED = RT_NVEC * VP, O_WP = O_RED + NW2 * 8, O_SC = O_WP + NW2, O_A = O_SC + 16;
    // parked scalars (uniform values, written redundantly by every thread with the same value)
    enum { SC_NB0 = 0, SC_NC0, SC_SIGMA, SC_SUMLOG, SC_RP, SC_RD, SC_GAP };
    double *const sc = sm + O_SC;
    double *const red = sm + O_RED;
    double *const A = sm + O_A;

    const int tid = threadIdx.x, inst = blockIdx.x;
    const int n = T.n, m = T.m, l = n + m + 1, lda = T.lda, nq = T.nq, z = T.z;
    const int j1 = tid / CH1, c1 = tid % CH1;      // column tiling (A^T, G)
    const int i2 = tid / CH2, c2 = tid % CH2;      // row tiling (A)
    const bool own1 = (c1 == 0) && (j1 < n);
    const bool own2 = (c2 == 0) && (i2 < m);
    const bool rowok = i2 < m;
    const double *Arow = A + (rowok ? i2 : 0) * lda + c2;
    const int e = tid;                              // element of (x, y, tau) this thread owns in the elementwise phases

    for (int i = tid; i < RT_NVEC * VP + RT_EXTRA; i += NT2) sm[i] = 0.0;
    for (int i = tid; i < m * lda; i += NT2) A[i] = 0.0;

    const bool row_is_nonneg = own2 && (i2 >= z) && (i2 < z + T.l);
    int soc_r0 = -1, soc_d = 0;
    if (e >= n && e < n + m) {
        const int c = T.rowcone[e - n];
        if (c >= 0) { soc_r0 = T.qoff[c]; soc_d = T.qoff[c + 1] - soc_r0; }
    }
    __syncthreads();
    // ---------------------------------------------------------------- load
    {
        const double *vals = Avals + (size_t)inst * T.nnz_aug;
        for (int k = tid; k < T.nnz_aug; k += NT2) {
            const double val = vals[k];
            const int r = T.rowidx[k], c = T.colidx[k];
            if (c < n) A[r * lda + c] = -val; else sm[O_BV + r] = val;
        }
        for (int j = tid; j < n; j += NT2) { sm[O_CV + j] = qv[j * sqk + inst * sqb]; sm[O_EV + j] = 1.0; }
        for (int i = tid; i < m; i += NT2) sm[O_DV + i] = 1.0;
    }
    __syncthreads();
    {
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT2) r[0] = fmax(r[0], fabs(sm[O_BV + i]));
        for (int j = tid; j < n; j += NT2) r[1] = fmax(r[1], fabs(sm[O_CV + j]));
        block_reduce_n<2, NW2>(r, 3u, red);
        sc[SC_NB0] = r[0]; sc[SC_NC0] = r[1]; sc[SC_SIGMA] = 1.0;
    }
    // ---- register tiles
    double at1[T1], gt[TG];
#pragma unroll
    for (int k = 0; k < T1; k++) { const int r = c1 + CH1 * k; at1[k] = (j1 < n && r < m) ? A[r * lda + j1] : 0.0; }

    // ---------------------------------------------------------------- equilibration on register tiles
    if (S.normalize) {
        double a2[T2];    // row tile, live only here
#pragma unroll
        for (int k = 0; k < T2; k++) a2[k] = rowok ? Arow[CH2 * k] : 0.0;      // row pads are zero
        for (int pass = 0; pass < NUM_RUIZ_PASSES + NUM_L2_PASSES; pass++) {
            const bool l2 = pass >= NUM_RUIZ_PASSES;
            const int oEt = (pass & 1) ? O_S2 : O_S1;     // ping-pong: column scaling of this pass
            const int oDt = (pass & 1) ? O_PX : O_TV;     // ping-pong: final row scaling of this pass
            double cn = 0, rn = 0;
            if (l2) {
#pragma unroll
                for (int k = 0; k < T1; k++) cn = fma(at1[k], at1[k], cn);
#pragma unroll
                for (int k = 0; k < T2; k++) rn = fma(a2[k], a2[k], rn);
                cn = sqrt(group_reduce<CH1, false>(cn)); rn = sqrt(group_reduce<CH2, false>(rn));
            } else {
#pragma unroll
                for (int k = 0; k < T1; k++) cn = fmax(cn, fabs(at1[k]));
#pragma unroll
                for (int k = 0; k < T2; k++) rn = fmax(rn, fabs(a2[k]));
                cn = group_reduce<CH1, true>(cn); rn = group_reduce<CH2, true>(rn);
            }
            if (own1) sm[oEt + j1] = 1.0 / sqrt(clamp_scale(cn));
            if (own2) sm[O_ZB + i2] = rn;                 // raw row norms
            __syncthreads();
            if (own2) {
                double a = rn;
                const int c = T.rowcone[i2];
                if (c >= 0) {   // block-average inside the SOC so the scaled cone is still the cone
                    const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
                    a = 0; for (int i = r0; i < r1; i++) a += sm[O_ZB + i];
                    a /= (double)(r1 - r0);
                }
                sm[oDt + i2] = 1.0 / sqrt(clamp_scale(a));
            }
            __syncthreads();
            {
                const double ej = sm[oEt + j1];           // pad entries are 0
#pragma unroll
                for (int k = 0; k < T1; k++) at1[k] *= sm[oDt + c1 + CH1 * k] * ej;
                __builtin_amdgcn_sched_barrier(0);        // keep the two scaling sweeps apart: peak register pressure of the kernel is here
                const double di = sm[oDt + i2];
#pragma unroll
                for (int k = 0; k < T2; k++) a2[k] *= di * sm[oEt + c2 + CH2 * k];
                if (own1) sm[O_EV + j1] *= ej;
                if (own2) sm[O_DV + i2] *= di;
            }
            // no barrier: the next pass writes the other ping-pong buffers (and ZB, last read before the barrier above)
        }
        __syncthreads();
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT2) { const double v = sm[O_BV + i] * sm[O_DV + i]; sm[O_BV + i] = v; r[0] = fmax(r[0], fabs(v)); }
        for (int j = tid; j < n; j += NT2) { const double v = sm[O_CV + j] * sm[O_EV + j]; sm[O_CV + j] = v; r[1] = fmax(r[1], fabs(v)); }
        block_reduce_n<2, NW2>(r, 3u, red);
        const double sigma = 1.0 / clamp_scale(fmax(r[0], r[1]));
        sc[SC_SIGMA] = sigma;
        for (int i = tid; i < m; i += NT2) sm[O_BV + i] *= sigma;
        for (int j = tid; j < n; j += NT2) sm[O_CV + j] *= sigma;
        for (int i = tid; i < VP; i += NT2) { sm[O_S1 + i] = 0.0; sm[O_S2 + i] = 0.0; sm[O_TV + i] = 0.0; sm[O_PX + i] = 0.0; sm[O_ZB + i] = 0.0; }
        if (rowok) {   // equilibrated A back to LDS
#pragma unroll
            for (int k = 0; k < T2; k++) { if (c2 + CH2 * k < n) A[i2 * lda + c2 + CH2 * k] = a2[k]; }
        }
        __syncthreads();
    }

    double scale = S.scale, hg = 0;
    const double rho_x = S.rho_x, rtau = TAU_FACTOR, alpha = S.alpha;
    auto dyv = [&](int i) -> double { return (i < z) ? ZERO_CONE_FACTOR * scale : scale; };   // 1 / r_y

    // ---- (re)factor:  gt <- tile of (rho_x I + A^T Dy A)^{-1};  g, h.g, phi.   Clobbers TV, PX, S1, S2.
    auto refactor = [&]() {
#pragma unroll
        for (int k = 0; k < TG; k++) gt[k] = 0.0;
        if (j1 < n) {   // S[a][j1], a = c1 + CH1*k  (row pads of A are zero and lda >= CH1*TG)
            const double *r = A;
            for (int i = 0; i < m; i++, r += lda) {
                const double aj = r[j1] * dyv(i);
#pragma unroll
                for (int k = 0; k < TG; k++) gt[k] = fma(r[c1 + CH1 * k], aj, gt[k]);
            }
#pragma unroll
            for (int k = 0; k < TG; k++) if (c1 + CH1 * k == j1) gt[k] += rho_x;
        }
        // Gauss-Jordan inversion, one barrier per pivot; pivot row / column published through LDS (double buffered):
        //   col[a] = G[a][kp] from threads j1 == kp ;  row[j] = G[kp][j] from threads c1 == kp % CH1 (tile slot kp / CH1)
        // pads of the buffers stay zero, so pad tile entries stay exactly zero without guards.
        if (j1 == 0) {
#pragma unroll
            for (int kk = 0; kk < TG; kk++) sm[O_TV + c1 + CH1 * kk] = gt[kk];
        }
        if (c1 == 0 && j1 < n) sm[O_S1 + j1] = gt[0];
        __syncthreads();
        int kp = 0;
#pragma unroll
        for (int slot = 0; slot < TG; slot++) {
            for (int rr = 0; rr < CH1 && kp < n; rr++, kp++) {
                const int oc = (kp & 1) ? O_PX : O_TV, orow = (kp & 1) ? O_S2 : O_S1;
                const int ocn = (kp & 1) ? O_TV : O_PX, orown = (kp & 1) ? O_S1 : O_S2;
                const double pinv = 1.0 / sm[orow + kp];
                const double rj = sm[orow + j1];
                const bool colthread = (j1 == kp);
#pragma unroll
                for (int kk = 0; kk < TG; kk++) {
                    const int a = c1 + CH1 * kk;
                    const double ca = sm[oc + a];
                    double v;
                    if (a == kp) v = colthread ? pinv : rj * pinv;
                    else if (colthread) v = -ca * pinv;
                    else v = fma(-ca * pinv, rj, gt[kk]);
                    gt[kk] = v;
                }
                // publish pivot kp+1
                const int kn = kp + 1;
                if (kn < n) {
                    if (j1 == kn) {
#pragma unroll
                        for (int kk = 0; kk < TG; kk++) sm[ocn + c1 + CH1 * kk] = gt[kk];
                    }
                    if (j1 < n) {
                        if (rr + 1 < CH1) { if (c1 == rr + 1) sm[orown + j1] = gt[slot]; }
                        else { if (c1 == 0) sm[orown + j1] = gt[(slot + 1 < TG) ? slot + 1 : slot]; }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < VP; i += NT2) { sm[O_TV + i] = 0.0; sm[O_PX + i] = 0.0; sm[O_S1 + i] = 0.0; sm[O_S2 + i] = 0.0; }
        // at1 is a register cache of (LDS-resident) A-hat: re-reading it here splits its live range around the
        // register-hungry inversion above, so the allocator never spills the tile inside the iteration loop.
#pragma unroll
        for (int k = 0; k < T1; k++) { const int r = c1 + CH1 * k; at1[k] = (j1 < n && r < m) ? A[r * lda + j1] : 0.0; }
        __syncthreads();
        for (int i = tid; i < m; i += NT2) sm[O_TV + i] = dyv(i) * sm[O_BV + i];
        __syncthreads();
        {
            const double a = tile_dot<CH1, T1>(at1, sm + O_TV + c1);
            if (own1) { const double cj = sm[O_CV + j1]; sm[O_S1 + j1] = cj - a; sm[O_S2 + j1] = cj + a; }   // rhs for g_x ; k = c + A^T Dy b
        }
        __syncthreads();
        {
            const double gx = tile_dot<CH1, TG>(gt, sm + O_S1 + c1), gk = tile_dot<CH1, TG>(gt, sm + O_S2 + c1);
            if (own1) { sm[O_G + j1] = gx; sm[O_PX + j1] = gk; }
        }
        __syncthreads();
        double r[1] = {0};
        {
            const double agx = row_dot<CH2, T2>(Arow, sm + O_G + c2, rowok), agk = row_dot<CH2, T2>(Arow, sm + O_PX + c2, rowok);
            if (own2) {
                const double bi = sm[O_BV + i2];
                const double gy = dyv(i2) * (agx + bi);
                sm[O_G + n + i2] = gy; r[0] += bi * gy;
                sm[O_PHI + n + i2] = bi - agk;
            }
            if (own1) { r[0] += sm[O_CV + j1] * sm[O_G + j1]; sm[O_PHI + j1] = rho_x * sm[O_PX + j1]; }
        }
        block_reduce_n<1, NW2>(r, 0u, red);
        hg = r[0];
        for (int i = tid; i < VP; i += NT2) { sm[O_TV + i] = 0.0; sm[O_PX + i] = 0.0; sm[O_S1 + i] = 0.0; sm[O_S2 + i] = 0.0; }
        __syncthreads();
    };
    auto phiw_partials = [&]() {
        double a = (e < l - 1) ? sm[O_PHI + e] * sm[O_W + e] : 0.0;
        a = wave_reduce_dpp<false>(a);
        if ((tid & 63) == 0) sm[O_WP + (tid >> 6)] = a;
    };

    if (e < l) sm[O_W + e] = (e == l - 1) ? 1.0 : 0.0;    // cold start (refactor() does not touch W)
    __syncthreads();

    int status = 0, iter = 0, last_scale_iter = 0, n_log = 0;
    const bool big_soc = T.maxq > SOC_SMALL;
    bool resume = false;     // true: the iteration interrupted by a rescale still owes its relaxed update

    // cone projection of element e from ZB (pre-projection values); small cones: recomputed by every row thread
    auto project_e = [&](int e) -> double {
        double ue = sm[O_ZB + e];
        if (soc_d > 1 && !big_soc) {
            const double *zc = sm + O_ZB + n + soc_r0;
            const double t0 = zc[0];
            double q0 = 0, q1 = 0; int k = 1;
            for (; k + 1 < soc_d; k += 2) { q0 = fma(zc[k], zc[k], q0); q1 = fma(zc[k + 1], zc[k + 1], q1); }
            if (k < soc_d) q0 = fma(zc[k], zc[k], q0);
            const double nz = sqrt(q0 + q1);
            const bool first = (e - n) == soc_r0;
            if (nz <= t0) { /* inside */ }
            else if (nz <= -t0) ue = 0.0;
            else { const double c0 = 0.5 * (t0 + nz); ue = first ? c0 : ue * (c0 / nz); }
        } else if (soc_d == 1) ue = fmax(ue, 0.0);
        return ue;
    };

    // relaxed update w += alpha (u - ut) and the per-wave partials of phi . w for the next iteration
    auto relaxed_update = [&](int e) {
        double a = 0;
        if (e < l) {
            const double we = sm[O_W + e] + alpha * (sm[O_U + e] - sm[O_UT + e]);
            sm[O_W + e] = we;
            if (e < l - 1) a = sm[O_PHI + e] * we;
        }
        a = wave_reduce_dpp<false>(a);
        if ((e & 63) == 0) sm[O_WP + (e >> 6)] = a;
        __syncthreads();
    };

    // Anderson acceleration (one secant pair, residual safeguard: the algorithm of k_fwd2 / k_sa_fwd / k_forward and of the oracle with aa_mem = 1); the four history
    // vectors of the instance live in global memory (aa_ws [B][4][lp]), entry e is only ever touched by thread e
    const int lp_aa = l + (l & 1);
    bool aa_on = S.acceleration_lookback > 0 && aa_ws != nullptr, aa_pending = false, aa_stale = false;
    const int aa_int = S.acceleration_interval > 0 ? S.acceleration_interval : 10;
    int aa_iter = 0, aa_rej = 0;
    double aa_normg = 0, aa_hs = 1.0;
    double *const aaXP = aa_ws ? aa_ws + (size_t)blockIdx.x * 4 * lp_aa : nullptr, *const aaFP = aaXP + lp_aa, *const aaFS = aaFP + lp_aa, *const aaWP = aaFS + lp_aa;

    // Outer loop: (re)factor, then iterate until convergence / iteration limit / a rescale request.  Keeping refactor()
    // out of the hot loop keeps its (large, fully unrolled) code and register pressure away from the iteration.
    for (bool done = false; !done;) {
    refactor();
    if (resume) { relaxed_update(tid); resume = false; iter++; }
    else { phiw_partials(); __syncthreads(); }
    for (;;) {
        if (iter >= S.max_iters) { done = true; break; }
        // Re-derive the thread coordinates from an opaque copy of the thread id: they become iteration-local values
        // (a handful of VALU ops) instead of kernel-lived registers that the allocator would spill to scratch.
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int j1 = t_ / CH1, c1 = t_ % CH1, i2 = t_ / CH2, c2 = t_ % CH2, e = t_;
        const bool own1 = (c1 == 0) && (j1 < n), own2 = (c2 == 0) && (i2 < m), rowok = i2 < m;
        const bool row_is_nonneg = own2 && (i2 >= z) && (i2 < z + T.l);
        const double *Arow = A + (rowok ? i2 : 0) * lda + c2;
        const bool check = (iter % CONVERGED_INTERVAL) == 0;
        const bool last = iter + 1 >= S.max_iters;
        if (aa_on) {
            bool w_changed = false;
            if (aa_pending) {      // safeguard: residual of the map at the accelerated point against the residual before the step
                double rs[1] = {0};
                if (e < l) { const double dd = aaWP[e] - sm[O_W + e]; rs[0] = dd * dd; }
                block_reduce_n<1, NW2>(rs, 0u, red);
                if (!(sqrt(rs[0]) <= aa_normg)) {
                    if (e < l) sm[O_W + e] = aaFS[e] * aa_hs;
                    aa_iter = 0; w_changed = true;
                    if (++aa_rej >= AA_MAX_REJECT) aa_on = false;
                }
                aa_pending = false;
            }
            if (aa_on && iter > 0 && iter % aa_int == 0 && !aa_stale) {
                if (aa_iter > 0) {
                    double rr[5] = {0, 0, 0, 0, 0}, xv = 0, fv = 0, fp = 0;
                    if (e < l) {
                        xv = aaWP[e]; fv = sm[O_W + e]; fp = aaFP[e] * aa_hs;
                        const double gv = xv - fv, xp = aaXP[e] * aa_hs, sv = xv - xp, yv = gv - (xp - fp);
                        rr[0] = sv * sv; rr[1] = yv * yv; rr[2] = sv * yv; rr[3] = sv * gv; rr[4] = gv * gv;
                    }
                    block_reduce_n<5, NW2>(rr, 0u, red);
                    const double mm = rr[2] + 1e-8 * sqrt(rr[0]) * sqrt(rr[1]), gam = rr[3] / mm;
                    const bool ok = fabs(mm) > 1e-300 && fabs(gam) < 1e10;
                    if (e < l) { aaXP[e] = xv; aaFP[e] = fv; if (ok) { aaFS[e] = fv; sm[O_W + e] = fv - gam * (fv - fp); } }
                    aa_hs = 1.0;
                    if (ok) { aa_normg = sqrt(rr[4]); aa_pending = true; w_changed = true; } else aa_iter = 0;
                } else {
                    if (e < l) { aaXP[e] = aaWP[e]; aaFP[e] = sm[O_W + e]; }
                    aa_hs = 1.0;
                }
                aa_iter++;
            }
            if (w_changed) {      // phi . w of the new input (the renormalisation below recomputes it again on check iterations)
                __syncthreads();
                { double a_ = (e < l - 1) ? sm[O_PHI + e] * sm[O_W + e] : 0.0; a_ = wave_reduce_dpp<false>(a_); if ((e & 63) == 0) sm[O_WP + (e >> 6)] = a_; }
                __syncthreads();
            }
        }
        if (check && iter > 0) {   // keep the homogeneous iterate in range
            const double we = (e < l) ? sm[O_W + e] : 0.0;
            double r[1] = {we * we};
            block_reduce_n<1, NW2>(r, 0u, red);
            const double nw = sqrt(r[0]);
            if (nw > 0 && e < l) sm[O_W + e] = we * (sqrt((double)l) / nw);
            if (nw > 0 && aa_on) { const double f = sqrt((double)l) / nw; aa_hs *= f; aa_normg *= f; }      // the stored history scales with w (lazily)
            __syncthreads();
            { double a_ = (e < l - 1) ? sm[O_PHI + e] * sm[O_W + e] : 0.0; a_ = wave_reduce_dpp<false>(a_); if ((e & 63) == 0) sm[O_WP + (e >> 6)] = a_; }
            __syncthreads();
        }
        if (aa_on && (aa_pending || (iter + 1) % aa_int == 0)) { aa_stale = false; if (e < l) aaWP[e] = sm[O_W + e]; }      // input of this iteration, where the next one needs it
        // P1a: t = rho_x w_x - A^T w_y
        {
            const double a = tile_dot<CH1, T1>(at1, sm + O_W + n + c1);
            if (own1) sm[O_TV + j1] = rho_x * sm[O_W + j1] - a;
        }
        __syncthreads();
        // P1b: p_x = G t
        {
            const double a = tile_dot<CH1, TG>(gt, sm + O_TV + c1);
            if (own1) sm[O_PX + j1] = a;
        }
        __syncthreads();
        // P2: q = A p_x ; tau-tilde ; u-tilde ; cone input
        {
            double numer = rtau * sm[O_W + l - 1];
#pragma unroll
            for (int k = 0; k < NW2; k++) numer += sm[O_WP + k];
            const double tau_t = numer / (rtau + hg);
            const double q = row_dot<CH2, T2>(Arow, sm + O_PX + c2, rowok);
            if (own2) {
                const int ee = n + i2;
                const double we = sm[O_W + ee];
                const double py = we + dyv(i2) * q;
                const double ute = py - tau_t * sm[O_G + ee];
                double ze = 2 * ute - we;
                if (row_is_nonneg && ze < 0) ze = 0;
                sm[O_UT + ee] = ute; sm[O_ZB + ee] = ze;
            }
            if (e < n) {
                const double ute = sm[O_PX + e] - tau_t * sm[O_G + e];
                sm[O_UT + e] = ute; sm[O_ZB + e] = 2 * ute - sm[O_W + e];
            }
            if (e == NT2 - 1) { sm[O_UT + l - 1] = tau_t; sm[O_ZB + l - 1] = fmax(0.0, 2 * tau_t - sm[O_W + l - 1]); }
        }
        __syncthreads();
        if (big_soc) {   // large cones: one leader per cone computes (c0, f) -> S1/S2, then rows apply (uniform branch)
            for (int c = e; c < nq; c += NT2) {
                const int r0 = n + T.qoff[c], r1 = n + T.qoff[c + 1];
                const double t0 = sm[O_ZB + r0]; double nz = 0;
                for (int k = r0 + 1; k < r1; k++) nz = fma(sm[O_ZB + k], sm[O_ZB + k], nz);
                nz = sqrt(nz);
                double c0, f;
                if (r1 - r0 == 1) { c0 = fmax(t0, 0.0); f = 0.0; }
                else if (nz <= t0) { c0 = t0; f = 1.0; }
                else if (nz <= -t0) { c0 = 0.0; f = 0.0; }
                else { c0 = 0.5 * (t0 + nz); f = c0 / nz; }
                sm[O_S1 + c] = c0; sm[O_S2 + c] = f;
            }
            __syncthreads();
            if (soc_d > 0) { const int c = T.rowcone[e - n]; sm[O_ZB + e] = ((e - n) == soc_r0) ? sm[O_S1 + c] : sm[O_S2 + c] * sm[O_ZB + e]; }
            __syncthreads();
            for (int c = e; c < nq; c += NT2) { sm[O_S1 + c] = 0.0; sm[O_S2 + c] = 0.0; }
        }
        if (!check && !last) {
            // P3 (fast path): project, relaxed update, phi.w partials
            double a = 0;
            if (e < l) {
                const double ue = project_e(e);
                const double we = sm[O_W + e] + alpha * (ue - sm[O_UT + e]);
                sm[O_U + e] = ue; sm[O_W + e] = we;
                if (e < l - 1) a = sm[O_PHI + e] * we;
            }
            a = wave_reduce_dpp<false>(a);
            if ((e & 63) == 0) sm[O_WP + (e >> 6)] = a;
            __syncthreads();
            iter++;
            continue;
        }
        // ---- slow path (every CONVERGED_INTERVAL iterations, and the last one)
        if (e < l) sm[O_U + e] = project_e(e);
        __syncthreads();
        bool stop = false, rescale = false;
        if (check) {
            const double ax_raw = row_dot<CH2, T2>(Arow, sm + O_U + c2, rowok);        // A-hat x-hat   (valid in row groups)
            const double aty_raw = tile_dot<CH1, T1>(at1, sm + O_U + n + c1);          // A-hat^T y-hat (valid in column groups)
            const double tau = fabs(sm[O_U + l - 1]);
            const double isg = 1.0 / sc[SC_SIGMA];
            double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rp, nax, ns, naxs, rd, naty (max) ; ctx, bty (sum)
            if (own2) {
                const int i = i2;
                const double sc_ = isg / sm[O_DV + i];
                const double ax = ax_raw * sc_;
                const double uy = sm[O_U + n + i];
                const double sh = (uy + sm[O_W + n + i] - 2 * sm[O_UT + n + i]) / dyv(i) * sc_;
                const double bt = sm[O_BV + i] * tau * sc_;
                r[0] = fabs(ax + sh - bt); r[1] = fabs(ax); r[2] = fabs(sh); r[3] = fabs(ax + sh);
                r[7] = sm[O_BV + i] * uy * isg * isg;
            }
            if (own1) {
                const int j = j1;
                const double sc_ = isg / sm[O_EV + j];
                const double aty = aty_raw * sc_;
                const double cj = sm[O_CV + j];
                r[4] = fabs(aty + cj * tau * sc_); r[5] = fabs(aty);
                r[6] = cj * sm[O_U + j] * isg * isg;
            }
            block_reduce_n<8, NW2>(r, 0x3Fu, red);
            const double rp = r[0], nax = r[1], ns = r[2], naxs = r[3], rd = r[4], naty = r[5], ctx = r[6], bty = r[7];
            const double nrm_b0 = sc[SC_NB0], nrm_c0 = sc[SC_NC0];
            if (tau > 0) {
                const double res_pri = rp / tau, res_dual = rd / tau, gap = fabs(ctx + bty) / tau;
                sc[SC_RP] = res_pri; sc[SC_RD] = res_dual; sc[SC_GAP] = gap;
                const double prl = fmax(fmax(nrm_b0 * tau, ns), nax) / tau, drl = fmax(nrm_c0 * tau, naty) / tau;
                const double grl = fmax(fabs(ctx), fabs(bty)) / tau;
                if (res_pri <= S.eps_abs + S.eps_rel * prl && res_dual <= S.eps_abs + S.eps_rel * drl &&
                    gap <= S.eps_abs + S.eps_rel * grl) { status = 1; stop = true; }
            }
            if (!stop && bty < 0 && naty / (-bty) <= S.eps_infeas) { status = -2; stop = true; }
            if (!stop && ctx < 0 && naxs / (-ctx) <= S.eps_infeas) { status = -1; stop = true; }
            if (!stop && S.adaptive_scale && iter > 0) {
                const double dp = fmax(fmax(nax, ns), nrm_b0 * tau), dd = fmax(naty, nrm_c0 * tau);
                const double rel_p = rp / (dp > 0 ? dp : 1), rel_d = rd / (dd > 0 ? dd : 1);
                if (rel_p > 0 && rel_d > 0 && isfinite(rel_p) && isfinite(rel_d)) {
                    const double sum_log = sc[SC_SUMLOG] + log(rel_p) - log(rel_d); n_log++;
                    __syncthreads();                 // everyone has read SC_SUMLOG before it is rewritten
                    sc[SC_SUMLOG] = sum_log;
                    const double factor = sqrt(exp(sum_log / n_log));
                    if (iter - last_scale_iter >= RESCALING_MIN_ITERS) {
                        const double ns2 = fmin(fmax(scale * factor, MIN_SCALE_VALUE), MAX_SCALE_VALUE);
                        if (ns2 != scale && (factor > sqrt(10.0) || factor < 1.0 / sqrt(10.0))) {
                            // keep (s, kappa):  w_y+ = rsk_y / r_y+ + 2 ut_y - u_y
                            const double dy_ratio = ns2 / scale;
                            if (e >= n && e < l - 1) {
                                const double ue = sm[O_U + e], ute = sm[O_UT + e];
                                const double d0 = ue + sm[O_W + e] - 2 * ute;
                                sm[O_W + e] = d0 * dy_ratio + 2 * ute - ue;
                            }
                            n_log = 0; last_scale_iter = iter; scale = ns2; aa_iter = 0; aa_pending = false; aa_stale = true;
                            __syncthreads();
                            sc[SC_SUMLOG] = 0.0;
                            rescale = true;
                        }
                    }
                }
            }
        }
        if (stop) { done = true; break; }
        if (last) { iter++; done = true; break; }
        if (rescale) { resume = true; break; }      // -> refactor() with the new scale, then finish this iteration
        relaxed_update(e);
        iter++;
    }
    }

    const double tau = fabs(sm[O_U + l - 1]);
    const double sigma = sc[SC_SIGMA];
    if (status == 0) {   // ran out of iterations (SCS set_unfinished)
        const double kap = fabs(rtau * (sm[O_U + l - 1] + sm[O_W + l - 1] - 2 * sm[O_UT + l - 1]));
        double r[2] = {0, 0};
        const double isg = 1.0 / sigma;
        if (e < n) r[0] = sm[O_CV + e] * sm[O_U + e] * isg * isg;
        else if (e < l - 1) r[1] = sm[O_BV + e - n] * sm[O_U + e] * isg * isg;
        block_reduce_n<2, NW2>(r, 0u, red);
        if (tau > kap) status = 2; else if (r[1] < r[0]) status = -7; else status = -6;
    }
    // ---------------------------------------------------------------- write back (un-normalise)
    {
        const bool solved = (status == 1 || status == 2);
        const bool infeas = (status == -2 || status == -7);
        const double it = solved ? 1.0 / (sigma * tau) : 1.0 / sigma;
        for (int j = tid; j < n; j += NT2) xo[(size_t)inst * n + j] = infeas ? NAN : sm[O_EV + j] * sm[O_U + j] * it;
        for (int i = tid; i < m; i += NT2) {
            const double uy = sm[O_U + n + i], di = sm[O_DV + i];
            const double sh = (uy + sm[O_W + n + i] - 2 * sm[O_UT + n + i]) / dyv(i);
            yo[(size_t)inst * m + i] = (solved || infeas) ? di * uy * it : NAN;
            so[(size_t)inst * m + i] = infeas ? NAN : sh / di * it;
        }
        if (tid == 0) {
            iters_o[inst] = iter; status_o[inst] = status;
            if (resid_o) { resid_o[3 * inst] = sc[SC_RP]; resid_o[3 * inst + 1] = sc[SC_RD]; resid_o[3 * inst + 2] = sc[SC_GAP]; }
        }
    }
}
