// ce_backward_ns.h -- SEARCH-FREE structured direct adjoint (round 6; hot path of the default adjoint for plain-cone templates).
//
// Same system as ce_backward_rt.h (diffcp's adjoint M^T r = dz with r_tau pinned to 0, reduced cone block by cone block to the saddle system
//        H r_x - B^T mu = f,   B r_x = d_B,        H = sum_c theta_c A_z^T (I - zh zh^T) A_z  (symmetric PSD),   B = the neq equality rows),
// but it is no longer eliminated as ONE (n + neq)-order matrix with a partial-pivoting search per pivot (59 serial pivots x 2.1 k cycles = 128 k of the
// 214 k cycles of an instance of the metric configuration, profiles/r05/z_setup_phase_cycles.log).  Null-space elimination instead:
//   1. B (neq x n, neq ~ 9) is brought to reduced row-echelon form with COLUMN pivoting by ONE wave, in place in LDS (the equality rows are rows of A;
//      a boundary cone's e_y row overwrites the cone's t-row, which H does not need): B P = [I  R],  x_piv = d~ - R x_free.  The only pivot searches left
//      are these neq wave-local ones (DPP butterflies, no workgroup barrier); the multipliers of the elimination stay in the pivot columns (the
//      transposed product B_1^-T g that yields mu is their reverse replay).
//   2. the rows that make up H are transformed to the null-space basis, a~_k = Z^T a_k = a_k[free] - R^T a_k[piv], in place;
//   3. the reduced Hessian Z^T H Z = sum_k w_k a~_k a~_k^T (order nf = n - neq ~ 41, with the right-hand side Z^T (f - H x_p) as one more column) is
//      accumulated on the MATRIX CORES in the accumulator layout of k_fwd2's S formation (half the k-rows and a quarter of the tiles of round 5's H);
//   4. it is symmetric positive definite on a regular instance: solved by k_fwd2's blocked SWEEP on the matrix cores (v_mfma_f64_16x16x4_f64 rank-4
//      updates, diagonal 4 x 4 pivot blocks, one barrier per block of four, NO search): ~11 blocks instead of 59 pivots;
//   5. x_piv, then mu = B_1^-T (H r_x - f)[piv], then r_y and the outputs exactly as ce_backward_rt.h.
// Rank deficiency (a redundant equality row, a reduced Hessian that is singular on the null space: the degenerate active sets of LP-like programs) shows up
// as a vanishing pivot in step 1 or 4: the variable is dropped (pivot inverse 0, finite numbers), the instance is flagged (adj_status bit 2 = 4) and appended to
// the device-side list whose instances the LSQR kernel behind this launch re-solves with diffcp's own method (cone_engine.hip ce_vjp_qp) -- this kernel is only
// launched when that re-solve is armed, so its answer on such instances is never the final one.
//
// Plain cones (zero / nonnegative / second-order), linear objective.  PSD / exponential / power cones and quadratic objectives keep k_backward_rt.
#pragma once

// pitch of the sweep's row buffers: 16 mod 32 doubles (conflict-free) with at least 16 doubles of gap behind the 16 NTILE entries of a row (the gaps of the
// eight buffer rows hold the ORIGINAL diagonal of the reduced Hessian: the rank tolerance of a pivot is relative to its own diagonal entry)
__host__ __device__ constexpr int bwd_ns_ldp(int NTILE) { return (16 * NTILE + 16) % 32 == 16 ? 16 * NTILE + 16 : 16 * NTILE + 32; }
__host__ __device__ inline int bwd_ns_kwmax(int m) { return ((m + 4) & ~3) + 4; }                                            // capacity of the weighted-row list (<= m entries + at least one pad, a multiple of 4)
__host__ __device__ constexpr int bwd_ns_nsl(int NTILE) { return (16 * NTILE - 3 + 63) / 64; }                               // 64-lane slots that hold the columns 0 .. n (n = right-hand side)
__host__ __device__ inline int bwd_ns_union_doubles(int n, int m, int nqs, int NTILE) {
    const int npad = n + (n & 1);
    int a = nqs * npad + 2 * (64 * bwd_ns_nsl(NTILE) + 2), b = 8 * bwd_ns_ldp(NTILE), c = m + (m & 1) + npad;      // {a_z, the row elimination's two publication buffers} | the sweep's row buffers | {q, g}
    int r = a > b ? a : b;
    return r > c ? r : c;
}
__host__ __device__ inline size_t bwd_ns_lds_bytes_of(int n, int m, int nq, int NTILE, int NTHR) {
    const int nqs = nq > 0 ? nq : 1, kw = bwd_ns_kwmax(m), npad = n + (n & 1);
    size_t d = (size_t)m * n + 1 + 2 * (size_t)m /* vv, dv */ + 3 * (size_t)npad /* rx, fvec, dB */ + 5 * (size_t)nqs + 2 * (size_t)kw /* tvec, wgt */ +
               (size_t)(NTHR / 64) * 8 + (size_t)nqs /* qaz */ + 1 + (size_t)bwd_ns_union_doubles(n, m, nqs, NTILE);
    size_t i = 2 * (size_t)m + 3 * (size_t)nqs + 4 * (size_t)n + 2 * (size_t)kw + (size_t)(NTHR / 64) + 1 + 8;
    return d * 8 + i * 4;
}

#ifdef CE_TIMING
#define NS_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) tstamp[i] = __builtin_readcyclecounter(); } while (0)
#define NS_SUB(i) do { if (threadIdx.x == 0) tsub[i] = __builtin_readcyclecounter(); } while (0)          // thread 0's clock, no barrier
#else
#define NS_STAMP(i) do { } while (0)
#define NS_SUB(i) do { } while (0)
#endif

template <int NTILE, int NTHR>
__global__ void __launch_bounds__(NTHR, (NTHR == 256 ? 3 : 1))
k_backward_ns(DevT T, const double *__restrict__ Avals, const double *__restrict__ xg, const double *__restrict__ yg, const double *__restrict__ sg,
              const double *__restrict__ dxg, const double *__restrict__ dyg, double *__restrict__ dAo, double *__restrict__ dqo, long sdqk, long sdqb,
              int *__restrict__ adj_status, int *__restrict__ fix) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int NWB = NTHR / 64, LDP = bwd_ns_ldp(NTILE), NSL = bwd_ns_nsl(NTILE), NCOLP = 64 * NSL, PUBP = NCOLP + 2;
    constexpr int NLOC = (16 * NTILE - 4 + NWB - 1) / NWB;          // equality rows per wave (rows are dealt cyclically to the waves)
    static_assert(NWB >= NTILE, "one wave per 16-row strip of the reduced system");
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, inst = blockIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, lc = lane & 15;
    const int n = T.n, m = T.m, lda = T.lda, nq = T.nq, z = T.z;
    const int nqs = nq > 0 ? nq : 1, npad = n + (n & 1), KWMAX = bwd_ns_kwmax(m);

    // ---- LDS carve
    double *p = sm;
    double *A = p; p += m * lda; p += (p - sm) & 1;
    double *vv = p; p += m;          // v = y - s ; later r_y
    double *dv = p; p += m;          // dy, then d = DPi dy ; later y (outputs)
    double *rx = p; p += npad;
    double *fvec = p; p += npad;     // f = dx + sum over boundary cones [...]; its free entries become Z^T f ; later x (outputs)
    double *dB = p; p += npad;       // right-hand side of the equalities -> d~ -> x_piv ; finally the multipliers mu
    double *cinfo = p; p += 5 * nqs; // per cone: lambda, |z|, e_y.d, e_s.d, theta
    double *tvec = p; p += KWMAX;    // per weighted row: t_k = a_k . x_p ; later q_k = a_k . r_x
    double *wgt = p; p += KWMAX;     // per weighted row: its weight in H (theta_c for the z-rows of cone c, -theta_c for a_z)
    double *red = p; p += NWB * 8;
    double *qaz = p; p += nqs;       // a_z . r_x per cone
    p += (p - sm) & 1;
    double *U = p; p += bwd_ns_union_doubles(n, m, nqs, NTILE);
    double *az = U;                  // a_z = A_z^T z-hat per boundary cone (pitch npad); dead after the Gram
    double *pub = U + nqs * npad;    // row elimination: two publication buffers {scaled pivot row (64 NSL), inverse pivot, pivot column}
    double *Rbuf = U;                // the sweep's two buffers of four rows
    double *qv2 = U;                 // (A r_x)_i for the rows of boundary cones, after the sweep
    double *mu = U + m + (m & 1);    // g = (H r_x - f)[piv] by equality index, after the sweep (the multipliers themselves end in dB)
    int *ip = (int *)p;
    int *rkind = ip; ip += m;
    int *eqrow = ip; ip += m;
    int *ckind = ip; ip += nqs;
    int *ceq = ip; ip += nqs;
    int *cbase = ip; ip += nqs;      // first entry of cone c in the weighted-row list
    int *erow = ip; ip += n;         // equality e -> offset of its row in sm
    int *pcol = ip; ip += n;         // equality e -> pivot column (-1: redundant row, dropped)
    int *cmap = ip; ip += n;         // column j -> free index f (>= 0) or -1 (pivot column)
    int *fcol = ip; ip += n;         // free index f -> column ; before the row elimination: equality e -> source (row index >= 0, or -1 - cone)
    int *esrc = fcol;
    int *wrow = ip; ip += KWMAX;     // weighted row -> offset of its row in sm
    int *wsrc = ip; ip += KWMAX;     // weighted row -> row index of A (>= 0) or -1 - cone (a_z)
    int *wcnt = ip; ip += NWB + 1;
    int *misc = ip; ip += 8;         // [0] n_eq, [1] nf, [2] flags, [3] KW

#ifdef CE_TIMING
    __shared__ long long tstamp[16], tsub[16];
    if (threadIdx.x < 16) tsub[threadIdx.x] = 0;
#endif
    NS_STAMP(0);
    // ---- load: the instance's values scattered into dense solver form A = -A_cvx (b is not needed: r_tau is pinned)
    {
        constexpr int LU = NTHR == 256 ? 20 : 12;
        const double *vals = Avals + (size_t)inst * T.nnz_aug;
        const int nnz = T.nnzA;
        double v0[LU]; int r0[LU], c0[LU];
#pragma unroll
        for (int u = 0; u < LU; u++) { const int k = tid + u * NTHR, kc = k < nnz ? k : 0; v0[u] = vals[kc]; r0[u] = T.rowidx[kc]; c0[u] = k < nnz ? T.colidx[kc] : -1; }
        for (int i = tid; i < m * lda; i += NTHR) A[i] = 0.0;
        for (int i = tid; i < m; i += NTHR) { vv[i] = yg[(size_t)inst * m + i] - sg[(size_t)inst * m + i]; dv[i] = dyg[(size_t)inst * m + i]; }
        if (tid < 8) misc[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < LU; u++) if (c0[u] >= 0) A[r0[u] * lda + c0[u]] = -v0[u];
        for (int kb = LU * NTHR; kb < nnz; kb += LU * NTHR) {
#pragma unroll
            for (int u = 0; u < LU; u++) { const int k = kb + tid + u * NTHR, kc = k < nnz ? k : 0; v0[u] = vals[kc]; r0[u] = T.rowidx[kc]; c0[u] = k < nnz ? T.colidx[kc] : -1; }
#pragma unroll
            for (int u = 0; u < LU; u++) if (c0[u] >= 0) A[r0[u] * lda + c0[u]] = -v0[u];
        }
        __syncthreads();
    }
    NS_STAMP(1);
    // ---- classify + d = DPi(v) dy in ONE pass: 16 lanes per cone (a row per lane, sums by DPP butterflies inside the 16-lane row), nonnegative rows beside them.
    //      (One thread per cone walked its rows in dependent loops: 8 of 256 threads busy, ~25 k cycles for the two phases.)  dv holds dy: transformed in place.
    for (int i = tid; i < z + T.l; i += NTHR) { const bool eq = (i < z || vv[i] > 0); rkind[i] = eq ? RK_EQ : RK_FREE; if (!eq) dv[i] = 0.0; }
    for (int c0 = 0; c0 < nq; c0 += NTHR / 16) {
        const int c = c0 + (tid >> 4), l16 = tid & 15;
        const bool cv = c < nq;
        const int r0 = cv ? T.qoff[c] : 0, r1 = cv ? T.qoff[c + 1] : 0, d = r1 - r0;
        const double t0 = cv ? vv[r0] : 0.0, h0 = cv ? dv[r0] : 0.0;
        double nz2 = 0.0, zh = 0.0;
        for (int i = r0 + 1 + l16; i < r1; i += 16) { const double w = vv[i]; nz2 = fma(w, w, nz2); zh = fma(w, dv[i], zh); }
        nz2 = group_reduce<16, false>(nz2); zh = group_reduce<16, false>(zh);
        const double nz = sqrt(nz2);
        int kind; double lam = 0.0;
        if (d == 1) kind = t0 >= 0 ? 0 : 1;
        else if (nz <= t0) kind = 0; else if (nz <= -t0) kind = 1; else { kind = 2; lam = (t0 + nz) / (2 * nz); }
        double zd = 0.0;
        const double i2n = kind == 2 ? 1.0 / (2 * nz) : 0.0, cz = kind == 2 ? t0 * zh / (nz * nz) : 0.0;
        const int rk = kind == 0 ? RK_EQ : (kind == 1 ? RK_FREE : RK_SOCB);
        for (int i = r0 + 1 + l16; i < r1; i += 16) {
            rkind[i] = rk;
            if (kind == 1) dv[i] = 0.0;
            else if (kind == 2) { const double w = vv[i]; const double di = (w * h0 + (t0 + nz) * dv[i] - w * cz) * i2n; dv[i] = di; zd = fma(w, di, zd); }
        }
        zd = group_reduce<16, false>(zd);
        if (kind == 2) {
            // u_i: the weight of row i of this cone in  f - dx = sum_c [ a_s (e_s.d) + (A_c^T d - a_y (e_y.d) - a_s (e_s.d)) / (1 - lam) ] = A^T u  (a_y, a_s are
            // combinations of the cone's rows: a_y = (a_0 + a_z) / sqrt 2, a_s = (a_0 - a_z) / sqrt 2, a_z = A_z^T z-hat), so that f is ONE pass over the rows
            // together with a_z, not a pass of its own behind a_z and a barrier
            const double d0 = (nz * h0 + zh) * i2n, zdn = zd / nz, eyd = (d0 + zdn) * M_SQRT1_2, esd = (d0 - zdn) * M_SQRT1_2;
            const double il = 1.0 / (1 - lam), k0c = (esd * (1 - il) - eyd * il) * M_SQRT1_2, kzc = (-esd * (1 - il) - eyd * il) * M_SQRT1_2, inz = 1.0 / nz;
            for (int i = r0 + 1 + l16; i < r1; i += 16) tvec[i] = fma(il, dv[i], vv[i] * inz * kzc);
            if (l16 == 0) tvec[r0] = fma(il, d0, k0c);
        }
        if (cv && l16 == 0) {
            rkind[r0] = rk; ckind[c] = kind; cinfo[5 * c] = lam; cinfo[5 * c + 1] = nz; cinfo[5 * c + 4] = lam / (1 - lam);
            if (kind == 1) dv[r0] = 0.0;
            else if (kind == 2) {
                const double d0 = (nz * h0 + zh) * i2n;
                dv[r0] = d0;
                const double zdn = zd / nz;
                cinfo[5 * c + 2] = (d0 + zdn) * M_SQRT1_2;   // e_y . d
                cinfo[5 * c + 3] = (d0 - zdn) * M_SQRT1_2;   // e_s . d
            }
        }
    }
    __syncthreads();
    NS_STAMP(2);
    // ---- equality numbering: ballot prefix sums (rows in order, then one e_y row per boundary cone); weighted-row list offsets
    {
        int base = 0;
        for (int i0 = 0; i0 < m; i0 += NTHR) {
            const int i = i0 + tid;
            const bool f = (i < m) && (rkind[i] == RK_EQ);
            const unsigned long long bal = __ballot(f);
            if (lane == 0) wcnt[wave] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wave; w++) off += wcnt[w];
            int tot = 0;
            for (int w = 0; w < NWB; w++) tot += wcnt[w];
            if (i < m) {
                const int e = f ? off + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
                eqrow[i] = e;
                if (f && e < n) { esrc[e] = i; erow[e] = (int)(A - sm) + i * lda; }
            }
            base += tot;
            __syncthreads();
        }
        // one lane per cone: the boundary cones before it give its equality index and the start of its weighted rows
        for (int c = tid; c < nq; c += NTHR) {
            int ne = base, kw = 0;
            for (int c2 = 0; c2 < c; c2++) { if (ckind[c2] == 2) { ne++; kw += T.qoff[c2 + 1] - T.qoff[c2]; } }
            if (ckind[c] == 2) {
                if (ne < n) { esrc[ne] = -1 - c; erow[ne] = (int)(A - sm) + T.qoff[c] * lda; }      // (the cone's t-row will hold a_y)
                ceq[c] = ne; cbase[c] = kw;
                ne++; kw += T.qoff[c + 1] - T.qoff[c];
            } else { ceq[c] = -1; cbase[c] = -1; }
            if (c == nq - 1) { misc[0] = ne; misc[3] = kw; }
        }
        if (nq == 0 && tid == 0) { misc[0] = base; misc[3] = 0; }
        __syncthreads();
    }
    const int neq = misc[0], KW = misc[3], KW4 = (KW + 4) & ~3;          // (at least one pad entry: entry KW stands for f in the null-space transform)
    if (neq > n) {   // more active rows than variables: rank deficient by counting (flagged, zero gradient; the LSQR launch behind this kernel serves it)
        for (int k = tid; k < T.nnz_aug; k += NTHR) dAo[(size_t)inst * T.nnz_aug + k] = 0.0;
        for (int j = tid; j <= n; j += NTHR) dqo[j * sdqk + inst * sdqb] = 0.0;
        if (tid == 0) { if (adj_status) adj_status[inst] = 2; if (fix) fix[1 + atomicAdd(fix, 1)] = inst; }
        return;
    }
    // ---- f = dx + A^T u  and  a_z = A_z^T z-hat  in ONE pass over the rows of the boundary cones (4 lanes per column, each takes every 4th cone; fixed summation order)
    for (int j0 = 0; j0 < n; j0 += NTHR / 4) {
        const int j = j0 + (tid >> 2), part = tid & 3;
        double acc = 0;
        if (j < n) {
            for (int c = part; c < nq; c += 4) {
                if (ckind[c] != 2) continue;
                const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
                double g = 0, g1 = 0, a = 0, a1 = 0;
                for (int i = r0; i < r1; i += 4) {
                    double av[4], uv[4], wv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int iu = min(i + u, r1 - 1); av[u] = A[iu * lda + j]; uv[u] = tvec[iu]; wv[u] = vv[iu]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const bool in = i + u < r1;
                        const double uu = in ? uv[u] : 0.0, ww = (in && i + u > r0) ? wv[u] : 0.0;
                        if (u & 1) { g1 = fma(av[u], uu, g1); a1 = fma(av[u], ww, a1); } else { g = fma(av[u], uu, g); a = fma(av[u], ww, a); }
                    }
                }
                az[c * npad + j] = (a + a1) / cinfo[5 * c + 1];
                acc += g + g1;
            }
        }
        acc = group_reduce<4, false>(acc);
        if (j < n && part == 0) fvec[j] = acc + dxg[(size_t)inst * n + j];
    }
    __syncthreads();
    // ---- the e_y row of every boundary cone overwrites the cone's t-row (H does not contain the t-row: theta A_c^T (I - e_y e_y^T - e_s e_s^T) A_c =
    //      theta (A_z^T A_z - a_z a_z^T)); right-hand sides of the equalities; the weighted-row list {z-rows (theta), a_z (-theta)} per boundary cone
    for (int idx = tid; idx < nq * n; idx += NTHR) {
        const int c = idx / n, j = idx - c * n;
        if (ckind[c] != 2) continue;
        const int r0 = T.qoff[c];
        A[r0 * lda + j] = (A[r0 * lda + j] + az[c * npad + j]) * M_SQRT1_2;
    }
    for (int e = tid; e < neq; e += NTHR) { const int src = esrc[e]; dB[e] = src >= 0 ? dv[src] : cinfo[5 * (-1 - src) + 2]; }
    for (int i = tid; i < m; i += NTHR) {
        const int c = (i >= z + T.l && nq > 0) ? T.rowcone[i] : -1;
        if (c >= 0 && ckind[c] == 2) {
            const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
            const double th = cinfo[5 * c + 4];
            if (i > r0) { const int q = cbase[c] + (i - r0 - 1); wrow[q] = (int)(A - sm) + i * lda; wsrc[q] = i; wgt[q] = th; }
            else { const int q = cbase[c] + (r1 - r0 - 1); wrow[q] = (int)(az - sm) + c * npad; wsrc[q] = -1 - c; wgt[q] = -th; }
        }
    }
    for (int q = KW + tid; q < KW4; q += NTHR) { wrow[q] = (int)(fvec - sm); wsrc[q] = -1 - nq; wgt[q] = 0.0; tvec[q] = 0.0; }      // entry KW = f (transformed like a row, weight 0), then pads to a multiple of four
    for (int j = tid; j < n; j += NTHR) { rx[j] = 0.0; cmap[j] = 0; }
    __syncthreads();
    NS_STAMP(3);
    // ---- 1. reduced row-echelon form of [B | d_B] with column pivoting.  The equality rows are dealt cyclically to the waves and live in REGISTERS (lane = column,
    //      column n = the right-hand side); per pivot the owner of the row finds the pivot column (DPP butterfly on |value| keys), publishes the scaled row, and
    //      after ONE barrier every wave updates its rows (multipliers by v_readlane from the pivot column's lane).  The pivot column keeps the multipliers and is
    //      carried through the later steps like any other column: it ends as column e of B_1^-1 (in-place Gauss-Jordan inversion), which mu needs.
    constexpr int NR1 = 24;          // equality rows one wave holds in registers (single-wave elimination: no barrier, no LDS traffic per pivot)
    if (NSL == 1 && neq <= NR1) {
        // Up to 24 equalities (metric configuration: ~17, i.e. ~9 active bounds + one e_y row per boundary cone): wave 0 keeps ALL rows in registers and runs the
        // whole elimination alone -- per pivot a DPP butterfly, one reciprocal and two v_readlane + one FMA per row; the other waves wait at the barrier behind it.
        // Rows are updated in groups of eight behind ONE uniform test (rows past neq hold zeros and are harmless to update): a scalar branch per row made a pivot
        // cost 2.8 k cycles, mostly branch latency.
        if (wave == 0) {          // (spreading this phase over the SIMDs by hardware wave slot was measured: no change -- it is bound by its own dependent chain, ~1.1 k cycles per pivot)
            double c1[NR1]; float rt1[NR1];
            static_for<NR1>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                const double *ptr = sm + (kk < neq ? erow[kk < neq ? kk : 0] : 0);
                const double v = (kk < neq && lane < n) ? ptr[lane < n ? lane : 0] : 0.0;
                c1[kk] = (kk < neq && lane == n) ? dB[kk < neq ? kk : 0] : v;
                rt1[kk] = (float)fabs(v);
            });
            static_for<NR1 / 8>([&](auto gc) {          // rank tolerance of a row: CE_RANK_TOL x its largest entry as loaded (kept in single precision: a threshold)
                constexpr int g = decltype(gc)::value;
                if (8 * g < neq) {
                    static_for<8>([&](auto kc) {
                        constexpr int kk = 8 * g + decltype(kc)::value;
                        const double rmax = wave_reduce_dpp<true>((double)rt1[kk]);
                        rt1[kk] = (float)(CE_RANK_TOL * (rmax > 0 ? rmax : 1.0));
                    });
                }
            });
            bool cfr = true;
#ifdef CE_TIMING
            long long pacc[4] = {0, 0, 0, 0}, pt0 = __builtin_readcyclecounter();
#define NS_PACC(k) do { const long long t1_ = __builtin_readcyclecounter(); pacc[k] += t1_ - pt0; pt0 = t1_; } while (0)
#else
#define NS_PACC(k) do { } while (0)
#endif
            static_for<NR1>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                if (e < neq) {          // (uniform)
                    NS_PACC(3);
                    const double rv = c1[e];
                    const double key = __hiloint2double(__double2hiint(rv) & 0x7fffffff, (__double2loint(rv) & ~0xFF) | (255 - lane));
                    const double best = wave_reduce_dpp<true>((lane < n && cfr) ? key : 0.0);
                    const int jb = 255 - (__double2loint(best) & 0xFF);
                    const bool ok = best >= (double)rt1[e] && jb < n;
                    NS_PACC(0);
                    if (ok) {
                        const double pv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rv), jb), __builtin_amdgcn_readlane(__double2loint(rv), jb));
                        double inv = __builtin_amdgcn_rcp(pv);
                        inv = fma(fma(-pv, inv, 1.0), inv, inv);
                        inv = fma(fma(-pv, inv, 1.0), inv, inv);
                        const bool isj = lane == jb;
                        // The scaled pivot row carries 1 + 1/p (instead of 1) in the pivot column: the ONE fused update c <- c - m re of another row then leaves
                        // m - m (1 + 1/p) = -m / p there -- the multiplier the in-place inverse keeps in that column -- with no blend per row (two v_cndmask and a
                        // multiply per row and pivot were half of the elimination's instructions).  Error of that entry: |m| (1 + |1/p|) ulp, i.e. relative
                        // (1 + |p|) ulp -- p is the largest entry of its row.
                        const double re = isj ? 1.0 + inv : rv * inv;
                        NS_PACC(1);
                        static_for<NR1 / 8>([&](auto gc) {
                            constexpr int g = decltype(gc)::value;
                            if (8 * g < neq) {          // (uniform; one test per eight rows)
                                double ml[8];
                                static_for<8>([&](auto kc) {          // the eight multipliers first (v_readlane pairs back to back), then the eight updates
                                    constexpr int i = 8 * g + decltype(kc)::value;
                                    ml[decltype(kc)::value] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c1[i]), jb), __builtin_amdgcn_readlane(__double2loint(c1[i]), jb));
                                });
                                static_for<8>([&](auto kc) {
                                    constexpr int i = 8 * g + decltype(kc)::value;
                                    if constexpr (i != e) c1[i] = fma(-ml[decltype(kc)::value], re, c1[i]);          // (the pivot column keeps the multiplier -> column e of B_1^-1)
                                });
                            }
                        });
                        c1[e] = isj ? inv : re;          // (the pivot row itself: 1/p in the pivot column)
                        cfr = cfr && !isj;
                        if (lane == 0) pcol[e] = jb;
                        NS_PACC(2);
                    } else if (lane == 0) { pcol[e] = -1; misc[2] |= 4; }          // redundant equality row: dropped (mu_e = 0), the instance is flagged
                }
            });
            static_for<NR1>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                if (kk < neq) { double *ptr = sm + erow[kk]; if (lane < n) ptr[lane] = c1[kk]; else if (lane == n) dB[kk] = c1[kk]; }
            });
            const bool fr = lane < n && cfr;
            const unsigned long long bal = __ballot(fr);
            if (lane < n) { if (fr) { const int f = __popcll(bal & ((1ull << lane) - 1ull)); cmap[lane] = f; fcol[f] = lane; } else cmap[lane] = -1; }
            if (lane == 0) misc[1] = __popcll(bal);
#ifdef CE_TIMING
            if (lane == 0) for (int k = 0; k < 4; k++) tsub[8 + k] = pacc[k];
#endif
        }
    } else {
        double col[NLOC][NSL];
        double rtol[NLOC];          // rank tolerance of a row: CE_RANK_TOL x its largest entry as loaded (scale-invariant per row; a redundant row ends at rounding level of that)
        bool cfree[NSL];
        static_for<NLOC>([&](auto kc) {          // (a compile-time loop: a plain unroll of this body stayed rolled and put the rows in scratch)
            constexpr int kk = decltype(kc)::value;
            const int e = wave + NWB * kk;
            const double *ptr = sm + (e < neq ? erow[e] : 0);
            const double de = e < neq ? dB[e] : 0.0;
            double rmax = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < NSL; s2++) {
                const int j = lane + 64 * s2;
                const double v = (e < neq && j < n) ? ptr[j < n ? j : 0] : 0.0;
                rmax = fmax(rmax, fabs(v));
                col[kk][s2] = (e < neq && j == n) ? de : v;
            }
            rmax = wave_reduce_dpp<true>(rmax);
            rtol[kk] = CE_RANK_TOL * (rmax > 0 ? rmax : 1.0);
        });
#pragma unroll
        for (int s2 = 0; s2 < NSL; s2++) cfree[s2] = true;
        for (int e = 0; e < neq; e++) {
            const int wo = e % NWB, k = e / NWB;
            double *pb = pub + (e & 1) * PUBP;
            if (wave == wo) {
                double re[NSL], ptolB = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < NSL; s2++) re[s2] = 0.0;
                static_for<NLOC>([&](auto kc) {
                    constexpr int kk = decltype(kc)::value;
                    if (k == kk) {
#pragma unroll
                        for (int s2 = 0; s2 < NSL; s2++) re[s2] = col[kk][s2];
                        ptolB = rtol[kk];
                    }
                });
                double best = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < NSL; s2++) {
                    const int j = lane + 64 * s2;
                    const double key = __hiloint2double(__double2hiint(re[s2]) & 0x7fffffff, (__double2loint(re[s2]) & ~0xFF) | (255 - j));      // |value|, 255 - column in the low mantissa bits
                    best = fmax(best, (j < n && cfree[s2]) ? key : 0.0);
                }
                best = wave_reduce_dpp<true>(best);
                const int jb = 255 - (__double2loint(best) & 0xFF);
                const bool ok = best >= ptolB && jb < n;
                const int jl = ok ? (jb & 63) : 0;
                double pv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(re[0]), jl), __builtin_amdgcn_readlane(__double2loint(re[0]), jl));
                if constexpr (NSL > 1) { if (jb >= 64) pv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(re[NSL - 1]), jl), __builtin_amdgcn_readlane(__double2loint(re[NSL - 1]), jl)); }
                const double pvs = ok ? pv : 1.0;
                double inv = __builtin_amdgcn_rcp(pvs);          // hardware seed + two Newton steps (the IEEE divide expansion sits on the path every wave waits for)
                inv = fma(fma(-pvs, inv, 1.0), inv, inv);
                inv = fma(fma(-pvs, inv, 1.0), inv, inv);
                inv = ok ? inv : 0.0;
#pragma unroll
                for (int s2 = 0; s2 < NSL; s2++) pb[lane + 64 * s2] = (lane + 64 * s2 == jb) ? 1.0 + inv : re[s2] * inv;      // (1 + 1/p in the pivot column: see the single-wave path)
                if (lane == 0) {
                    pb[NCOLP] = inv; reinterpret_cast<int *>(pb + NCOLP + 1)[0] = ok ? jb : -1;
                    pcol[e] = ok ? jb : -1;
                    if (!ok) misc[2] |= 4;          // redundant equality row: dropped (mu_e = 0), the instance is flagged
                }
            }
            __syncthreads();
            // header, inverse pivot and this lane's entries of the scaled row in ONE round trip (requested together, before the header is looked at)
            const int jbv = reinterpret_cast<const int *>(pb + NCOLP + 1)[0];
            const double inv = pb[NCOLP];
            double r[NSL];
#pragma unroll
            for (int s2 = 0; s2 < NSL; s2++) r[s2] = pb[lane + 64 * s2];
            const int jb = __builtin_amdgcn_readfirstlane(jbv);
            if (jb < 0) continue;
            const int jl = jb & 63;
            static_for<(NLOC + 3) / 4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (NWB * 4 * g < neq) {          // (uniform; one test per four local rows: rows past neq hold zeros and are harmless to update)
                    static_for<4>([&](auto kc) {
                        constexpr int kk = 4 * g + decltype(kc)::value;
                        if constexpr (kk < NLOC) {
                            const int ei = wave + NWB * kk;
                            double mlt = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(col[kk][0]), jl), __builtin_amdgcn_readlane(__double2loint(col[kk][0]), jl));
                            if constexpr (NSL > 1) { if (jb >= 64) mlt = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(col[kk][NSL - 1]), jl), __builtin_amdgcn_readlane(__double2loint(col[kk][NSL - 1]), jl)); }
                            const bool isp = ei == e;          // (uniform: the pivot row itself becomes the scaled row, 1/p in the pivot column)
#pragma unroll
                            for (int s2 = 0; s2 < NSL; s2++) {
                                const int j = lane + 64 * s2;
                                col[kk][s2] = isp ? ((j == jb) ? inv : r[s2]) : fma(-mlt, r[s2], col[kk][s2]);          // the pivot column keeps the multiplier (-> column e of B_1^-1)
                            }
                        }
                    });
                }
            });
#pragma unroll
            for (int s2 = 0; s2 < NSL; s2++) if (lane + 64 * s2 == jb) cfree[s2] = false;
        }
        // rows back to LDS (in place: R in the free columns, B_1^-1 in the pivot columns), d~ -> dB
        static_for<NLOC>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            const int e = wave + NWB * kk;
            if (e < neq) {
                double *ptr = sm + erow[e];
#pragma unroll
                for (int s2 = 0; s2 < NSL; s2++) { const int j = lane + 64 * s2; if (j < n) ptr[j] = col[kk][s2]; else if (j == n) dB[e] = col[kk][s2]; }
            }
        });
        if (wave == 0) {          // free columns numbered in increasing order (cfree is the same in every wave)
            int basef = 0;
#pragma unroll
            for (int s2 = 0; s2 < NSL; s2++) {
                const int j = lane + 64 * s2;
                const bool fr = j < n && cfree[s2];
                const unsigned long long bal = __ballot(fr);
                if (j < n) {
                    if (fr) { const int f = basef + __popcll(bal & ((1ull << lane) - 1ull)); cmap[j] = f; fcol[f] = j; }
                    else cmap[j] = -1;
                }
                basef += __popcll(bal);
            }
            if (lane == 0) misc[1] = basef;
        }
    }
    __syncthreads();
    NS_STAMP(4);
    const int nf = misc[1];
    const int NB = (nf + 3) >> 2, cr = 4 * NB;          // blocks of four that hold a reduced variable; column of the right-hand side
    // ---- 2. null-space transform of the weighted rows and of f (list entry KW), in place, on the matrix cores:
    //      row[j] <- row[j] - sum_e row[p_e] R[e][j]  (free j),   t = sum_e row[p_e] d~_e        i.e.  D = A_w[:, piv] [R | d~]   (KW+1 x neq) (neq x n+1)
    //      wave J owns the column tile 16 J .. 16 J + 15 (column n = t); 64 list rows per pass; K = the equalities, four per instruction.
    {
        const int NCT = (n + 16) >> 4;                   // column tiles that hold a column <= n
        if (wave < NCT) {
            const int colj = 16 * wave + lc;
            for (int g0 = 0; g0 <= KW; g0 += 64) {
                const int ng = min(4, (KW - g0 + 16) >> 4);          // row groups of this pass that hold a list row <= KW (uniform)
                v4d tacc[4];
                int wr[4];
#pragma unroll
                for (int g = 0; g < 4; g++) { tacc[g] = v4d{0.0, 0.0, 0.0, 0.0}; wr[g] = wrow[min(g0 + 16 * g + lc, KW4 - 1)]; }
                for (int k0 = 0; k0 < neq; k0 += 4) {
                    const int kq = min(k0 + lg, neq - 1);
                    const int pc = (k0 + lg < neq) ? pcol[kq] : -1, er = erow[kq];
                    const double bop = (pc >= 0 && colj <= n) ? (colj < n ? sm[er + colj] : dB[kq]) : 0.0;
                    const int pcs = pc >= 0 ? pc : 0;
                    double aop[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) aop[g] = sm[wr[g] + pcs];
#pragma unroll
                    for (int g = 0; g < 4; g++) if (g < ng) tacc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(pc >= 0 ? aop[g] : 0.0, bop, tacc[g], 0, 0, 0);
                }
                // write-back in three fenced stages (list offsets, old values, stores): sixteen dependent read-modify-writes in series otherwise
                const bool wcol = colj < n && cmap[colj < n ? colj : 0] >= 0, tcol = colj == n;
                int wq[4][4];
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int r = 0; r < 4; r++) wq[g][r] = wrow[min(g0 + 16 * g + lg + 4 * r, KW4 - 1)];
                __builtin_amdgcn_sched_barrier(0);
                double ov[4][4];
                const int cj = colj < n ? colj : 0;
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int r = 0; r < 4; r++) ov[g][r] = sm[wq[g][r] + cj];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int q = g0 + 16 * g + lg + 4 * r;
                        if (q <= KW) {
                            if (wcol) sm[wq[g][r] + cj] = ov[g][r] - tacc[g][r];
                            else if (tcol && q < KW) tvec[q] = tacc[g][r];
                        }
                    }
            }
        }
    }
    __syncthreads();
    NS_STAMP(5);
    // ---- 3. reduced Hessian (+ right-hand side as column cr) on the matrix cores: wave w accumulates the strip of rows 16 w .. 16 w + 15.
    //      Operands two steps ahead of the instruction that consumes them (list entry, then the row's values): the loop is a chain of LDS round trips otherwise.
    v4d acc[NTILE];
#pragma unroll
    for (int J = 0; J < NTILE; J++) acc[J] = v4d{0.0, 0.0, 0.0, 0.0};
    const int jmax = (cr >> 4) + 1;                     // tiles / strips that hold a row or column <= cr
    if (wave < jmax) {
        // Operand addresses, not operand arithmetic: a lane whose reduced column does not exist reads a ZERO (a pad entry of tvec), the lane of the right-hand-side
        // column reads t_q itself (the column is negated behind the loop), every other lane reads the row's entry of its original column -- so the B operands go
        // from LDS straight into the matrix cores and the A operand takes ONE multiply (the weight).  With fp64 multiplies and blends between the MFMAs the loop
        // ran at ~200 cycles per MFMA (the fp64 vector operations queue behind the matrix instruction in flight); requested in batches of four steps.
        const int tvoff = (int)(tvec - sm), zoff = tvoff + KW;          // (tvec[KW] is a pad entry: 0.0)
        int oc[NTILE], kd[NTILE];          // per tile: original column of this lane's reduced column; kind 0 row entry, 1 the step's t_q, 2 zero
#pragma unroll
        for (int J = 0; J < NTILE; J++) {
            const int colr = 16 * J + lc;
            oc[J] = colr < nf ? fcol[colr < nf ? colr : 0] : 0; kd[J] = colr < nf ? 0 : (colr == cr ? 1 : 2);
        }
        int ocw = 0, kdw = 2;
        static_for<NTILE>([&](auto Jc) { constexpr int J = decltype(Jc)::value; if (wave == J) { ocw = oc[J]; kdw = kd[J]; } });
        double frhs[4];          // (Z^T f)[row] of this lane's four strip rows: requested now, used behind the loop
#pragma unroll
        for (int r = 0; r < 4; r++) { const int rowi = 16 * wave + lg + 4 * r; frhs[r] = fvec[fcol[rowi < nf ? rowi : 0]]; }
        NS_SUB(0);
        constexpr int UB = 4;
        int wr[UB]; double ww[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) { const int q = min(4 * u + lg, KW4 - 1); wr[u] = wrow[q]; ww[u] = wgt[q]; }
        for (int k0 = 0; k0 < KW4; k0 += 4 * UB) {
            double aB[UB], bB[UB][NTILE];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int tq = tvoff + min(k0 + 4 * u + lg, KW4 - 1);
                aB[u] = sm[kdw == 0 ? wr[u] + ocw : (kdw == 1 ? tq : zoff)];
#pragma unroll
                for (int J = 0; J < NTILE; J++) bB[u][J] = sm[kd[J] == 0 ? wr[u] + oc[J] : (kd[J] == 1 ? tq : zoff)];
            }
            double wc[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) wc[u] = ww[u];
#pragma unroll
            for (int u = 0; u < UB; u++) { const int q = min(k0 + 4 * UB + 4 * u + lg, KW4 - 1); wr[u] = wrow[q]; ww[u] = wgt[q]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UB; u++) aB[u] *= wc[u];          // the four weighted A operands first, then nothing but matrix instructions
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (k0 + 4 * u < KW4) {          // (uniform: the last batch may be short)
#pragma unroll
                    for (int J = 0; J < NTILE; J++) if (J < jmax) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB[u], bB[u][J], acc[J], 0, 0, 0);
                }
            }
        }
        NS_SUB(1);
        // + Z^T f on the right-hand-side column; identity on the padding rows of the last block
        static_for<NTILE>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            static_for<4>([&](auto rc) {          // (compile-time indices into the accumulators: a rolled loop here puts them in scratch)
                constexpr int r = decltype(rc)::value;
                const int rowi = 16 * wave + lg + 4 * r;
                const bool rhs = 16 * J + lc == cr;          // the right-hand-side column: Z^T f - sum_k w_k t_k a~_k (the loop accumulated + sum w t a~ there)
                const bool pad = wave == J && lc == lg + 4 * r && rowi >= nf && rowi < cr;
                acc[J][r] = pad ? 1.0 : (rhs ? (rowi < nf ? frhs[r] : 0.0) - acc[J][r] : acc[J][r]);
            });
        });
    }
    NS_SUB(2);
    __syncthreads();          // the Gram's reads of a_z are complete: the union region becomes the sweep's row buffers
    // the ORIGINAL diagonal of the reduced Hessian -> the gaps behind the buffer rows (entry k at row k >> 4 of the buffers, 16 NTILE + (k & 15)); rows 0 .. 3 -> buffer 0
    if (wave < NTILE) {
        static_for<NTILE>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            if (wave == J) {
                double dsel = 0.0;
                static_for<4>([&](auto rc) { constexpr int r = decltype(rc)::value; dsel = (lc == lg + 4 * r) ? acc[J][r] : dsel; });
                if ((lc & 3) == lg) Rbuf[J * LDP + 16 * NTILE + lc] = dsel;          // (the diagonal lanes: lc = lg + 4 r)
            }
        });
    }
    NS_STAMP(6);
    // ---- 4. blocked sweep on the matrix cores (ce_forward_v2.h, S inversion): after block b the rows K = {4b .. 4b+3} hold P R, the others S - C P R;
    //      the right-hand-side column ends as (Z^T H Z)^-1 rhs.  A pivot that is not positive against the tolerance drops its variable (pivot inverse 0) and flags.
    if (wave == 0) {
#pragma unroll
        for (int J = 0; J < NTILE; J++) Rbuf[lg * LDP + 16 * J + lc] = acc[J][0];
    }
    __syncthreads();
    bool tiny = false;
    static_for<4 * NTILE>([&](auto bc) {
        constexpr int b = decltype(bc)::value, k0 = 4 * b, wo = k0 / 16, r0 = (k0 % 16) / 4, c0 = k0 % 16;
        constexpr int k1 = k0 + 4, wn = (k1 / 16) % NTILE, r1 = (k1 % 16) / 4;
        if (b < NB) {
            const double *Rb = Rbuf + (b & 1) * (4 * LDP);
            double *Rn = Rbuf + ((b + 1) & 1) * (4 * LDP);
            if (wave < NTILE) {
                double Bop[NTILE], Cop[4], a[4][4];
#pragma unroll
                for (int J = 0; J < NTILE; J++) Bop[J] = Rb[lg * LDP + 16 * J + lc];
                {
                    int coff = 16 * wave + lc;
                    asm volatile("" : "+v"(coff));
#pragma unroll
                    for (int pq = 0; pq < 4; pq++) Cop[pq] = Rb[pq * LDP + coff];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const double2 v0 = *reinterpret_cast<const double2 *>(Rb + q * LDP + k0), v1 = *reinterpret_cast<const double2 *>(Rb + q * LDP + k0 + 2);
                    a[q][0] = v0.x; a[q][1] = v0.y; a[q][2] = v1.x; a[q][3] = v1.y;
                }
                double dg[4];          // the block's original diagonal entries (rank tolerance relative to each)
                {
                    const double2 d0 = *reinterpret_cast<const double2 *>(Rbuf + wo * LDP + 16 * NTILE + c0), d1 = *reinterpret_cast<const double2 *>(Rbuf + wo * LDP + 16 * NTILE + c0 + 2);
                    dg[0] = d0.x; dg[1] = d0.y; dg[2] = d1.x; dg[3] = d1.y;
                }
                __builtin_amdgcn_sched_barrier(0);
                double e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) e[k] = (lg == k) ? 1.0 : 0.0;
                double pinv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double pv = a[k][k];
                    const bool ok = pv > CE_RANK_TOL * dg[k];
                    tiny |= !ok;
                    const double pvs = ok ? pv : 1.0;
                    double pi_ = __builtin_amdgcn_rcp(pvs);
                    pi_ = fma(fma(-pvs, pi_, 1.0), pi_, pi_);
                    pi_ = fma(fma(-pvs, pi_, 1.0), pi_, pi_);
                    pi_ = ok ? pi_ : 0.0;
                    pinv[k] = pi_;
#pragma unroll
                    for (int i = k + 1; i < 4; i++) {
                        const double lm = a[i][k] * pi_;
#pragma unroll
                        for (int j = k + 1; j < 4; j++) a[i][j] = fma(-lm, a[k][j], a[i][j]);
                        e[i] = fma(-lm, e[k], e[i]);
                    }
                }
                double xs[4];
                xs[3] = e[3] * pinv[3];
                xs[2] = fma(-a[2][3], xs[3], e[2]) * pinv[2];
                xs[1] = fma(-a[1][3], xs[3], fma(-a[1][2], xs[2], e[1])) * pinv[1];
                xs[0] = fma(-a[0][3], xs[3], fma(-a[0][2], xs[2], fma(-a[0][1], xs[1], e[0]))) * pinv[0];
                double wsel = -(Cop[0] * xs[0] + Cop[1] * xs[1] + Cop[2] * xs[2] + Cop[3] * xs[3]);
                const int mr = lc - c0;
                const bool kcol = mr >= 0 && mr < 4;
                const double nk = kcol ? 0.0 : 1.0;
                if (wave == wo) {
                    double psel = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; k++) psel = fma(xs[k], (mr == k) ? 1.0 : 0.0, psel);
                    wsel = fma(wsel, nk, psel);
#pragma unroll
                    for (int J = 0; J < NTILE; J++) acc[J][r0] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) acc[wo][r] *= nk;
                Bop[wo] = fma(Bop[wo], nk, (kcol && mr == lg) ? -1.0 : 0.0);
#pragma unroll
                for (int J = 0; J < NTILE; J++) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, Bop[J], acc[J], 0, 0, 0);
                if (b + 1 < NB && wave == wn) {
#pragma unroll
                    for (int J = 0; J < NTILE; J++) Rn[lg * LDP + 16 * J + lc] = acc[J][r1];
                }
            }
            __syncthreads();
        }
    });
    if (tiny && lane == 0) misc[2] |= 4;          // (every lane of every strip solved the same 4 x 4 blocks: the flag is uniform)
    NS_STAMP(7);
    // ---- 5. solution: x_free from the right-hand-side column, x_piv = d~ - R x_free
    if (wave < NTILE) {
        static_for<NTILE>([&](auto Jc) {
            constexpr int J = decltype(Jc)::value;
            static_for<4>([&](auto rc) { constexpr int r = decltype(rc)::value; const int rowi = 16 * wave + lg + 4 * r; if (16 * J + lc == cr && rowi < nf) rx[fcol[rowi]] = acc[J][r]; });
        });
    }
    __syncthreads();
    // q_k = a_k . r_x = t_k + a~_k . x_free for the z-rows of boundary cones (pivot entries of r_x are still 0; the a_z entries of the list are skipped: their
    // storage was the sweep's), x_piv per equality: 4 lanes per row
    for (int q0 = 0; q0 < KW + neq; q0 += NTHR / 4) {
        const int q = q0 + (tid >> 2), part = tid & 3;
        double a0 = 0.0, a1 = 0.0;
        const bool isw = q < KW;
        const bool live = q < KW + neq && (!isw || wsrc[q] >= 0);
        if (live) {
            const double *row = sm + (isw ? wrow[q] : erow[q - KW]);
            for (int j = part; j < n; j += 8) {
                const int j2 = min(j + 4, n - 1);
                const double r0v = row[j], r1v = row[j2], x0 = rx[j], x1 = rx[j2];
                a0 = fma(r0v, x0, a0); a1 = fma(r1v, (j + 4 < n) ? x1 : 0.0, a1);
            }
        }
        const double a = group_reduce<4, false>(a0 + a1);
        if (live && part == 0) {
            if (isw) tvec[q] += a;
            else { const int e = q - KW; dB[e] = (pcol[e] >= 0) ? dB[e] - a : 0.0; }      // x_piv (the entries of the pivot columns -- the in-place inverse -- met r_x = 0)
        }
    }
    __syncthreads();
    // per boundary cone: a_z . r_x = z-hat . (A_z r_x); q of its rows -> qv2 (the union region: the sweep's buffers are dead); the list keeps
    // q'_i = q_i - z-hat_i (a_z . r_x), so that  H r_x = sum over z-rows of theta a_i q'_i  needs no a_z
    for (int c = tid; c < nq; c += NTHR) {
        if (ckind[c] != 2) continue;
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1], qb = cbase[c];
        const double inz = 1.0 / cinfo[5 * c + 1];
        double zq = 0.0;
        for (int i = r0 + 1; i < r1; i++) zq = fma(vv[i], tvec[qb + i - r0 - 1], zq);
        zq *= inz;
        qaz[c] = zq;
        for (int i = r0 + 1; i < r1; i++) { const double qi = tvec[qb + i - r0 - 1]; qv2[i] = qi; tvec[qb + i - r0 - 1] = qi - vv[i] * inz * zq; }
    }
    for (int e = tid; e < neq; e += NTHR) { const int pc = pcol[e]; if (pc >= 0) rx[pc] = dB[e]; }      // r_x[p_e] = x_piv
    __syncthreads();
    // g_e = (H r_x - f)[p_e] = sum over z-rows of theta a_i[p_e] q'_i - f[p_e]   (16 lanes per equality) -> mu[] (as g)
    for (int e0 = 0; e0 < neq; e0 += NTHR / 16) {
        const int e = e0 + (tid >> 4), part = tid & 15;
        double a = 0.0;
        const int pc = e < neq ? pcol[e] : -1;
        if (pc >= 0) for (int q = part; q < KW; q += 16) { if (wsrc[q] >= 0) a = fma(wgt[q] * sm[wrow[q] + pc], tvec[q], a); }
        a = group_reduce<16, false>(a);
        if (e < neq && part == 0) mu[e] = pc >= 0 ? a - fvec[pc] : 0.0;
    }
    __syncthreads();
    // mu = B_1^-T g.  The pivot columns of the eliminated rows hold B_1^-1 (in-place Gauss-Jordan inversion: column p_e carried column e of the accumulated row
    // operations through the later steps): mu_e = sum_i T[i][e] g_i -> dB
    for (int e0 = 0; e0 < neq; e0 += NTHR / 16) {
        const int e = e0 + (tid >> 4), part = tid & 15;
        double a = 0.0;
        const int pc = e < neq ? pcol[e] : -1;
        if (pc >= 0) for (int i = part; i < neq; i += 16) a = fma(sm[erow[i] + pc], mu[i], a);          // (g of a dropped row is 0)
        a = group_reduce<16, false>(a);
        if (e < neq && part == 0) dB[e] = pc >= 0 ? a : 0.0;
    }
    __syncthreads();
    NS_STAMP(8);
    // ---- r_y (as k_backward_rt; the t-row of a boundary cone holds a_y: a_0 . r_x = sqrt 2 (a_y . r_x) - a_z . r_x)
    for (int i = tid; i < z + T.l; i += NTHR) vv[i] = (eqrow[i] >= 0) ? dB[eqrow[i]] : dv[i];
    for (int c = tid; c < nq; c += NTHR) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        if (ckind[c] == 0) { for (int i = r0; i < r1; i++) vv[i] = dB[eqrow[i]]; }
        else if (ckind[c] == 1) { for (int i = r0; i < r1; i++) vv[i] = dv[i]; }
        else {
            const double lam = cinfo[5 * c], inz = 1.0 / cinfo[5 * c + 1], eyd = cinfo[5 * c + 2], esd = cinfo[5 * c + 3];
            const double rhoy = dB[ceq[c]];
            const double zq = qaz[c];                                             // z-hat . (A_z r_x) = a_z . r_x
            const double eyq = pcol[ceq[c]] >= 0 ? eyd : 0.0;                     // a_y . r_x = e_y . d (the equality; a dropped row is flagged anyway)
            const double q0 = M_SQRT2 * eyq - zq;                                 // a_0 . r_x
            const double esq = (q0 - zq) * M_SQRT1_2;
            const double il = 1.0 / (1 - lam);
            const double cy = rhoy - il * (eyd - lam * eyq), cs = esd - il * (esd - lam * esq);
            const double k0 = (cy + cs) * M_SQRT1_2, kz = (cy - cs) * M_SQRT1_2;
            for (int i = r0 + 1; i < r1; i++) { const double zh = vv[i] * inz; vv[i] = il * (dv[i] - lam * qv2[i]) + kz * zh; }
            vv[r0] = il * (dv[r0] - lam * q0) + k0;
        }
    }
    __syncthreads();
    NS_STAMP(9);
    // ---- outputs in the boundary convention: dA_eval = [-dA.data, db[b_idx]], dq_eval = [dc, 0]   (diffcp_if.py:91-92)
    double *const xs = fvec, *const ys = dv;
    for (int j = tid; j < n; j += NTHR) xs[j] = xg[(size_t)inst * n + j];
    for (int i = tid; i < m; i += NTHR) ys[i] = yg[(size_t)inst * m + i];
    __syncthreads();
    {
        constexpr int OU = 8;
        double *const dArow = dAo + (size_t)inst * T.nnz_aug;
        const int nnz = T.nnz_aug;
        for (int k0 = tid; k0 < nnz; k0 += OU * NTHR) {
            int ii[OU], jj[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) { const int kk = min(k0 + u * NTHR, nnz - 1); ii[u] = T.rowidx[kk]; jj[u] = T.colidx[kk]; }
            __builtin_amdgcn_sched_barrier(0);
            double xv[OU], rv[OU], vi[OU], yi[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) { const int jc = jj[u] < n ? jj[u] : 0; xv[u] = xs[jc]; rv[u] = rx[jc]; vi[u] = vv[ii[u]]; yi[u] = ys[ii[u]]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < OU; u++) {
                const double va = -(xv[u] * vi[u] - yi[u] * rv[u]);
                const double val = (jj[u] < n) ? va : -vi[u];
                if (k0 + u * NTHR < nnz) dArow[k0 + u * NTHR] = val;
            }
        }
    }
    for (int j = tid; j <= n; j += NTHR) dqo[j * sdqk + inst * sdqb] = (j < n) ? -rx[j] : 0.0;
    if (tid == 0) {
        const int fl = misc[2];
        if (adj_status) adj_status[inst] = fl;
        if (fix && (fl & 4)) fix[1 + atomicAdd(fix, 1)] = inst;      // rank-deficient system: diffcp's LSQR element replaces this answer (ce_vjp_qp)
    }
#ifdef CE_TIMING
    NS_STAMP(10);
    if (tid < 10) dAo[(size_t)inst * T.nnz_aug + tid] = (double)(tstamp[tid + 1] - tstamp[tid]);
    if (tid == 10) dAo[(size_t)inst * T.nnz_aug + 10] = (double)(n + neq);
    if (tid == 11) dAo[(size_t)inst * T.nnz_aug + 11] = (double)nf;
    if (tid == 12) { dAo[(size_t)inst * T.nnz_aug + 12] = (double)(tsub[0] - tstamp[5]); dAo[(size_t)inst * T.nnz_aug + 13] = (double)(tsub[1] - tsub[0]); dAo[(size_t)inst * T.nnz_aug + 14] = (double)(tsub[2] - tsub[1]); dAo[(size_t)inst * T.nnz_aug + 15] = (double)KW; }
    if (tid >= 16 && tid < 20) dAo[(size_t)inst * T.nnz_aug + tid] = (double)tsub[8 + tid - 16];
#endif
}
