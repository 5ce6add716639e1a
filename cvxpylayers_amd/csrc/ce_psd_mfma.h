// ce_psd_mfma.h -- projection of a PSD block onto the cone with the dense contractions on the matrix cores
// (v_mfma_f64_16x16x4_f64) and a WARM-STARTED Jacobi eigensolver.  Included inside an anonymous namespace after ce_forward_v2.h.
//
// The ADMM iterates change little from one iteration to the next, so the eigenvectors V of the previous projection almost
// diagonalise the new matrix S:   S' = V^T S V   (two k x k x k products, MFMA)   is nearly diagonal and the cyclic Jacobi sweeps
// of ce_forward_v2.h (psd_jacobi) converge on it in 1-2 sweeps instead of 6-8; they keep accumulating their rotations into V, so
// that S = V diag(w) V^T again, and the projection is   X = V diag(max(w, 0)) V^T   (one more MFMA product).  Callers restart
// from V = I every check interval, which bounds the loss of orthogonality of the accumulated V.
//
// Layout: S, V, T are KP x KP (KP = 16 ceil(k / 16), zero padded) row-major in LDS with pitch P = KP + 1.  Operand reads follow the
// f64 MFMA maps (A: lane l -> [l & 15][l >> 4], B: lane l -> [l >> 4][l & 15]); the accumulator of lane l holds rows (l >> 4) + 4 r
// of column l & 15.  The matrices are tiny (k = 20: 2 x 2 tiles, 8 k-steps per tile), one tile per wave; bank conflicts of the
// strided operand reads do not matter at this size.
#pragma once

typedef double psd_v4d __attribute__((ext_vector_type(4)));

// idx / d for 0 <= idx < 65536, 1 <= d <= 64 without the ~40-instruction integer division expansion (rcp = 1.0f / d, computed once):
// (idx + 0.5) / d stays at least 0.5 / d away from every integer, far more than the rounding error of the float product.
__device__ __forceinline__ int psd_fdiv(int idx, float rcp) { return (int)(((float)idx + 0.5f) * rcp); }

// Jacobi rotation (c, s) annihilating the (p, q) entry:  theta = (a_qq - a_pp) / (2 a_pq),  t = sign(theta) / (|theta| + sqrt(theta^2 + 1)),
// c = 1 / sqrt(t^2 + 1), s = t c  -- with hardware reciprocal / reciprocal-square-root seeds and Newton / Goldschmidt refinement instead of
// the IEEE divide and square-root expansions (four of them, ~200 dependent instructions, sat on the critical path of EVERY round of
// every sweep: one lane per pair computes this while the rest of the workgroup waits).  c^2 + s^2 = 1 to rounding, which is what keeps V
// orthogonal; the angle itself only needs to be approximately optimal for the sweeps to converge.
__device__ __forceinline__ double psd_rcp(double v) { double r = __builtin_amdgcn_rcp(v); r = fma(fma(-v, r, 1.0), r, r); return fma(fma(-v, r, 1.0), r, r); }
__device__ __forceinline__ void psd_rotation(double app, double aqq, double apq, double &c, double &sn) {
    c = 1.0; sn = 0.0;
    if (apq == 0.0) return;
    const double theta = (aqq - app) * psd_rcp(2 * apq), at = fabs(theta);
    double t;
    if (at > 1e100) t = 0.5 * psd_rcp(theta);
    else { double sq, ri; sqrt_rsqrt(fma(theta, theta, 1.0), sq, ri); t = (theta >= 0 ? 1.0 : -1.0) * psd_rcp(at + sq); }
    double sq2, ri2;
    sqrt_rsqrt(fma(t, t, 1.0), sq2, ri2);
    c = ri2; sn = t * ri2;
}

// D = A B on KT x KT tiles of 16 x 16; fa(M, K), fb(K, N): operand elements, out(M, N, v): result sink.  All waves of the workgroup call it.
template <int NTH, class FA, class FB, class FO>
__device__ __forceinline__ void psd_mfma_gemm(int KT, FA &&fa, FB &&fb, FO &&out, int ks = 0) {
    // ks: contraction steps that carry data ((k + 3) / 4 for a k x k block in zero-padded storage; 0: all 4 KT).  The operands of five steps are requested together, then
    // the five MFMAs run back to back (a plain loop waits one LDS round trip per MFMA, eight per tile at KT = 2); a step past ks multiplies by a zero A operand.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lg = lane >> 4, lc = lane & 15;
    const int nst = ks > 0 ? ks : 4 * KT, smax = 4 * KT - 1;
    for (int t = wave; t < KT * KT; t += NTH / 64) {
        const int ti = t / KT, tj = t - ti * KT;
        psd_v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int s0 = 0; s0 < nst; s0 += 5) {
            double a[5], b[5];
#pragma unroll
            for (int u = 0; u < 5; u++) { const int sc = min(s0 + u, smax); a[u] = fa(16 * ti + lc, 4 * sc + lg); b[u] = fb(4 * sc + lg, 16 * tj + lc); if (s0 + u >= nst) a[u] = 0.0; }
            __builtin_amdgcn_sched_barrier(0);      // (keeps the five operand pairs in flight together: see psd_gemm_kk)
#pragma unroll
            for (int u = 0; u < 5; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) out(16 * ti + lg + 4 * r, 16 * tj + lc, acc[r]);
    }
}

// Cyclic Jacobi sweeps on S (k x k, pitch P), rotations accumulated into the columns of V (pitch P); same tournament order and rotation
// formulas as psd_jacobi (ce_forward_v2.h).  On return diag(S) holds the eigenvalues and the columns of V the eigenvectors.  A round is two
// barrier-separated phases: (A) the K / 2 rotation parameters, one lane per pair; (B) the two-sided update S <- J^T S J on the (K / 2)^2
// disjoint 2 x 2 blocks {p_a, q_a} x {p_b, q_b} together with the column rotations V <- V J.  Stop rule: off-diagonal mass below 1e-15 of the
// total, or -- the warm-started matrix S' = V^T S V carries rounding noise of a few 1e-16 |S| per entry, so that target can sit below the
// floor -- already tiny (1e-12) and no longer decreasing.  (Variants that were measured and dropped: one wave doing the whole sweep without
// workgroup barriers; one barrier per round with recomputed or pipelined rotations -- ROUND_NOTES.md.)
template <int NTH>
__device__ __forceinline__ void psd_sweeps_wg(double *Sm, double *Vm, int k, int P, double *cs, double *red, double stop_rel = 0.0) {      // stop_rel > 0: stop as soon as off^2 <= stop_rel * total (a refinement polishes)
    constexpr int NT = NTH, NW = NTH / 64;
    const int tid = threadIdx.x;
    const int K = (k + 1) & ~1, H = K / 2;
    const float rk = 1.0f / (float)k, rH = 1.0f / (float)H;
    double prev_off = 0;
    for (int sweep = 0; sweep < 40; sweep++) {
        double r[2] = {0, 0};
        for (int idx = tid; idx < k * k; idx += NT) { const int i = psd_fdiv(idx, rk), j = idx - i * k; const double v = Sm[i * P + j]; if (i == j) r[1] = fma(v, v, r[1]); else r[0] = fma(v, v, r[0]); }
        block_reduce_n<2, NW>(r, 0u, red);
        if (r[0] <= 1e-30 * (r[0] + r[1]) || r[0] == 0.0 || (r[0] <= 1e-24 * (r[0] + r[1]) && r[0] > 0.25 * prev_off) || r[0] <= stop_rel * (r[0] + r[1])) break;          // uniform
        prev_off = r[0];
        for (int rd = 0; rd < K - 1; rd++) {
            if (tid < H) {
                const int t = tid;
                int p = rd + t; if (p >= K - 1) p -= K - 1;
                if (t == 0) p = K - 1;
                int q = rd + K - 1 - t; if (q >= K - 1) q -= K - 1;
                if (p > q) { const int t_ = p; p = q; q = t_; }
                double c = 1.0, sn = 0.0;
                if (q < k) psd_rotation(Sm[p * P + p], Sm[q * P + q], Sm[p * P + q], c, sn); else q = -1;
                cs[4 * t] = c; cs[4 * t + 1] = sn; cs[4 * t + 2] = (double)p; cs[4 * t + 3] = (double)q;
            }
            __syncthreads();
            const int kh = (k + 1) >> 1;                               // V <- V J: two rows per work item (k = 20: 100 + 100 items, one pass of 256 threads)
            for (int w = tid; w < H * H + kh * H; w += NT) {
                if (w < H * H) {                                       // S <- J^T S J on the 2 x 2 block (pair a) x (pair b)
                    const int a = psd_fdiv(w, rH), b = w - a * H;
                    const double ca = cs[4 * a], sa = cs[4 * a + 1], cb = cs[4 * b], sb = cs[4 * b + 1];
                    const int pa = (int)cs[4 * a + 2], qa = (int)cs[4 * a + 3], pb = (int)cs[4 * b + 2], qb = (int)cs[4 * b + 3];
                    const double s00 = Sm[pa * P + pb], s01 = qb >= 0 ? Sm[pa * P + qb] : 0.0, s10 = qa >= 0 ? Sm[qa * P + pb] : 0.0, s11 = (qa >= 0 && qb >= 0) ? Sm[qa * P + qb] : 0.0;
                    const double t00 = cb * s00 - sb * s01, t01 = sb * s00 + cb * s01, t10 = cb * s10 - sb * s11, t11 = sb * s10 + cb * s11;
                    Sm[pa * P + pb] = ca * t00 - sa * t10;
                    if (qb >= 0) Sm[pa * P + qb] = ca * t01 - sa * t11;
                    if (qa >= 0) Sm[qa * P + pb] = sa * t00 + ca * t10;
                    if (qa >= 0 && qb >= 0) Sm[qa * P + qb] = sa * t01 + ca * t11;
                } else {                                               // V <- V J
                    const int it = w - H * H, rp = psd_fdiv(it, rH), b = it - rp * H;
                    const int pb = (int)cs[4 * b + 2], qb = (int)cs[4 * b + 3];
                    if (qb >= 0) {
                        const double cb = cs[4 * b], sb = cs[4 * b + 1];
                        const int row = 2 * rp;
                        const double x = Vm[row * P + pb], y = Vm[row * P + qb];
                        Vm[row * P + pb] = cb * x - sb * y; Vm[row * P + qb] = sb * x + cb * y;
                        if (row + 1 < k) {
                            const double x1 = Vm[(row + 1) * P + pb], y1 = Vm[(row + 1) * P + qb];
                            Vm[(row + 1) * P + pb] = cb * x1 - sb * y1; Vm[(row + 1) * P + qb] = sb * x1 + cb * y1;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Warm-started eigen-REFINEMENT on the matrix cores (round 3).  The Jacobi sweeps above cost ~1.4 k cycles per round (two barrier-separated
// phases of dependent work), 19 rounds per sweep, 2-3 sweeps per warm-started projection: ~90 us per ADMM iteration for one 20 x 20 block,
// 92 % of k_sa_fwd at BASELINE config 4 with the matrix cores idle 98.6 % of the time.  Once consecutive ADMM iterates differ by less than
// about 1 % (after ~20 of ~100 iterations at config 4) the previous eigenvectors V are an excellent approximate eigenbasis of the new S,
// and the decomposition can be REFINED instead of recomputed (Ogita & Aishima, "Iterative refinement for symmetric eigenvalue decomposition",
// 2018): with  R = I - V^T V,  D = V^T S V,  lam_i = D_ii / (1 - R_ii),
//        E_ij = (D_ij + lam_j R_ij) / (lam_j - lam_i)   (i != j),      E_ii = R_ii / 2,            V <- V + V E
// converges quadratically (off-diagonal mass 1e-3 -> 1e-6 -> 1e-12) and corrects the loss of orthogonality of V as it goes (no periodic restart
// from the identity any more).  Everything except the k^2 divisions is k x k x k products: four MFMA products per step, one workgroup barrier
// each.  A step is only taken when every |E_ij| <= 1/2 (first-order perturbation theory is valid for all pairs); otherwise -- early iterations,
// eigenvalue pairs closer than the perturbation -- the routine falls back to the warm-started Jacobi sweeps on D (or to a cold start when V
// has drifted from orthogonality or a refinement step was already taken).
// Storage: COMPACT k x k row-major matrices with pitch P >= k (no zero padding; the MFMA operand reads are guarded), the contraction runs
// over ceil(k / 4) steps of 4 instead of the padded KP / 4 (k = 20: 5 steps, not 8).
// out(M, N, sum_K A(M, K) B(K, N)) for M, N < k, with A(M, K) = pA[M * sAm + K * sAk] (* max(wK[K], 0) when wK != null) and B(K, N) = pB[K * sBk + N * sBn].
// Address arithmetic is what this costs (the products themselves are five MFMA instructions per tile at k = 20): row / column indices beyond k
// are CLAMPED instead of guarded (their results are never stored), each lane keeps one base pointer per operand and steps it by a uniform
// stride, the operands of up to five contraction steps are loaded before their products are issued; only a contraction index beyond k
// (last step of a k that is not a multiple of 4) is masked to zero.
template <int NTH, class FO>
__device__ __forceinline__ void psd_gemm_kk(int k, const double *pA, int sAm, int sAk, const double *pB, int sBk, int sBn, const double *wK, FO &&out) {
    const int KT = (k + 15) >> 4, KS = (k + 3) >> 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lg = lane >> 4, lc = lane & 15;
    for (int t = wave; t < KT * KT; t += NTH / 64) {
        const int ti = t / KT, tj = t - ti * KT;
        const int am = 16 * ti + lc, bn = 16 * tj + lc;
        const double *qa = pA + min(am, k - 1) * sAm + lg * sAk, *qb = pB + lg * sBk + min(bn, k - 1) * sBn;
        const double *qw = wK ? wK + lg : nullptr;
        psd_v4d acc = {0.0, 0.0, 0.0, 0.0};
        constexpr int CH = 5;
        for (int s0 = 0; s0 < KS; s0 += CH) {
            double a[CH], b[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) {       // branch-free: every operand of the chunk is requested before the first MFMA (a guarded step is a scalar branch around
                const int st = s0 + u, kk = 4 * st + lg;                   // its loads AND its MFMA: one LDS round trip per MFMA); a step past k multiplies by a zero A operand
                const int kc = (kk < k) ? 4 * st : (k - 1 - lg);           // clamped contraction index (relative to the lane's lg)
                a[u] = qa[kc * sAk]; b[u] = qb[kc * sBk];
                if (qw) a[u] *= fmax(qw[kc], 0.0);                         // (wK: eigenvalues; the product wants their positive parts)
                if (kk >= k) a[u] = 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);      // (without the fence the scheduler sinks every operand pair next to its MFMA again: one LDS round trip per product step)
#pragma unroll
            for (int u = 0; u < CH; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) { const int M = 16 * ti + lg + 4 * r; if (M < k && bn < k) out(M, bn, acc[r]); }
    }
}

__host__ __device__ inline int psd_refine_pitch(int k) { return k | 1; }                 // odd pitch: the strided operand reads spread over the banks
// LDS doubles of the scratch shared by all blocks of an instance (S, T / E, D, R + rotation parameters + eigenvalues + lam); every block keeps k P more (V)
__host__ __device__ inline int psd_refine_scratch_doubles(int kmax) { return 4 * kmax * psd_refine_pitch(kmax) + 4 * kmax + 16; }

// phase timing: debug builds only (-DCE_PSD_TIMING; ticks are accumulated in registers and flushed once per projection -- a global atomic per phase
// would stall the instrumented wave for longer than the phase itself).  In the product build the macro is empty: the tick registers and clock reads
// pushed k_sa_fwd further into scratch.
#ifdef CE_PSD_TIMING
#define PSD_TICK(slot) do { if (stats) { const long long t_ = clock64(); tk[(slot) - 8] += t_ - tph; tph = t_; } } while (0)
#else
#define PSD_TICK(slot) do { } while (0)
#endif

// One refinement loop.  Returns 0: converged (Va: eigenvectors, ev: eigenvalues);  1: the very first step would leave the basin of the first-order
// correction and Va is still the caller's orthonormal V: D = Va^T S Va was written to Dm (not symmetrised) for Jacobi sweeps;  2: start over (cold).
// FUSED version (k <= 32, one 16 x 16 tile of D and R per wave, kept in the MFMA accumulators): THREE barriers per step instead of six, and neither
// T = S V nor D, R ever go through LDS --
//   A. every wave forms the column block T(:, tj) = S V(:, tj) of its own tile in registers (the accumulator layout of a tile IS the B-operand
//      layout of the next product: row lg + 4 q of lane l = contraction index 4 s + lg for q = s), then D(ti, tj) = V(:, ti)^T T(:, tj) and
//      (V^T V)(ti, tj) with the same A operand; the diagonal lanes publish lam_i = D_ii (1 + R_ii)                                          | barrier
//   B. E for the lane's own four entries straight from the accumulators (D_ij is used as it is: the two triangles differ by rounding only),
//      E -> LDS, the step's statistics as one DPP reduction per wave                                                                        | barrier
//   C. every thread combines the four partial statistics, decides, V' = V + V E                                                            | barrier
template <int NTH>
__device__ __forceinline__ int psd_refine_loop_fused(int k, int P, const double *Sm, double *&Va, double *&Vb, double *Tm, double *Dm, double *lam, double *ev,
                                                      double *red, unsigned long long *stats, int refine, int maxl, long long (&tk)[6]) {
    (void)tk;
    constexpr int NW = NTH / 64, KSM = 5;                      // k <= 20: five contraction steps (the caller routes larger blocks to the LDS loop)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lc = lane & 15;
    const int KT = (k + 15) >> 4;
    const bool has_tile = wave < KT * KT;
    const int ti = (KT == 2) ? (wave >> 1) : 0, tj = (KT == 2) ? (wave & 1) : 0;
    const int am = 16 * ti + lc, bn = 16 * tj + lc, amc = min(am, k - 1), bnc = min(bn, k - 1);
#ifdef CE_PSD_TIMING
    long long tph = stats ? clock64() : 0;
#endif
    double prev_off = 0;
    for (int it = 0; it < maxl; it++) {
        psd_v4d aD = {0.0, 0.0, 0.0, 0.0}, aG = {0.0, 0.0, 0.0, 0.0};         // D tile, (V^T V) tile
        if (has_tile) {
            // contraction index of step u for this lane: 4 u + lg, clamped to k - 1; steps beyond k contribute zero through the A operand.  ALL operands of the phase are
            // requested up front and every step runs unconditionally (a step past k multiplies by a zero A operand): guarded steps (`if (u < KS)`, a scalar branch around
            // each MFMA) made every MFMA wait for its own LDS round trip -- ten in a row for T, five more for D / R
            double bv[KSM], a0[KSM], a1[KSM], av[KSM];
            {
                const double *srow = Sm + min(lc, k - 1) * P, *srow1 = Sm + min(16 + lc, k - 1) * P;
#pragma unroll
                for (int u = 0; u < KSM; u++) {
                    const int kc = min(4 * u + lg, k - 1);
                    bv[u] = Va[kc * P + bnc]; a0[u] = srow[kc]; a1[u] = (KT == 2) ? srow1[kc] : 0.0; av[u] = Va[kc * P + amc];
                }
#pragma unroll
                for (int u = 0; u < KSM; u++) if (4 * u + lg >= k) { a0[u] = 0.0; a1[u] = 0.0; av[u] = 0.0; }
            }
            psd_v4d aT0 = {0.0, 0.0, 0.0, 0.0}, aT1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < KSM; u++) aT0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], bv[u], aT0, 0, 0, 0);
            if (KT == 2) {
#pragma unroll
                for (int u = 0; u < KSM; u++) aT1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], bv[u], aT1, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < KSM; u++) {
                const double tb = (u < 4) ? aT0[u & 3] : aT1[u & 3];
                aD = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], tb, aD, 0, 0, 0);
                aG = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], aG, 0, 0, 0);
            }
            if (ti == tj) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int M = 16 * ti + lg + 4 * r;
                    if (M == bn && M < k) { const double rii = 1.0 - aG[r]; lam[M] = fma(aD[r], rii, aD[r]); ev[M] = aD[r] * psd_rcp(1.0 - rii); }
                }
            }
        }
        __syncthreads();
        PSD_TICK(8);
        double r[4] = {0, 0, 0, 0};                                   // off^2, diag^2 (sums) ; max |R|, max |E| (1e300: a pair outside the basin) (max)
        if (has_tile && bn < k) {
            const double lj = lam[bn];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int M = 16 * ti + lg + 4 * q;
                if (M >= k) continue;
                const double d = aD[q], rr = (M == bn ? 1.0 : 0.0) - aG[q];
                double e;
                if (M == bn) { e = 0.5 * rr; r[1] = fma(d, d, r[1]); r[3] = fmax(r[3], fabs(e)); }
                else {
                    const double den = lj - lam[M], num = fma(lj, rr, d);
                    if (!(fabs(num) <= 0.8 * fabs(den)) || den == 0.0) { e = 0.5 * rr; if (num != 0.0) r[3] = 1e300; }
                    else { e = num * psd_rcp(den); r[3] = fmax(r[3], fabs(e)); }
                    r[0] = fma(d, d, r[0]);
                }
                r[2] = fmax(r[2], fabs(rr));
                Tm[M * P + bn] = e;
            }
        }
        r[0] = wave_reduce_dpp<false>(r[0]); r[1] = wave_reduce_dpp<false>(r[1]); r[2] = wave_reduce_dpp<true>(r[2]); r[3] = wave_reduce_dpp<true>(r[3]);
        if (lane == 0) { red[4 * wave] = r[0]; red[4 * wave + 1] = r[1]; red[4 * wave + 2] = r[2]; red[4 * wave + 3] = r[3]; }
        __syncthreads();
        PSD_TICK(9);
        double off2 = 0, dg2 = 0, rmax = 0, emax = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) { off2 += red[4 * w]; dg2 += red[4 * w + 1]; rmax = fmax(rmax, red[4 * w + 2]); emax = fmax(emax, red[4 * w + 3]); }
        const double tot = off2 + dg2;
        if (it == 0 && rmax > 1e-6) return 2;                         // the stored V drifted from orthogonality (never after a converged call)
        if (off2 <= 1e-28 * tot && rmax <= 1e-14) return 0;           // converged, nothing left to correct (both tests: D_ij = (lam_i + lam_j) delta_ij does not
                                                                      // see a loss of orthogonality between two columns with lam_i = -lam_j, R does)
        if (emax >= 1e300 || !refine || (it > 0 && !(off2 <= 0.25 * prev_off))) {
            if (it > 0 || rmax > 1e-10) return 2;
            if (has_tile && bn < k) {
#pragma unroll
                for (int q = 0; q < 4; q++) { const int M = 16 * ti + lg + 4 * q; if (M < k) Dm[M * P + bn] = aD[q]; }
            }
            __syncthreads();
            return 1;
        }
        prev_off = off2;
        psd_gemm_kk<NTH>(k, Va, P, 1, Tm, P, 1, nullptr, [&](int M, int N, double v) { Vb[M * P + N] = Va[M * P + N] + v; });      // V' = Va + Va E
        __syncthreads();
        { double *t_ = Va; Va = Vb; Vb = t_; }
        PSD_TICK(10);
        if (stats && tid == 0) atomicAdd(&stats[1], 1ull);
        if (emax <= 1e-7) return 0;                                   // quadratic convergence: this step leaves corrections (and a loss of orthogonality) of order |E|^2 <= 1e-14.
                                                                      // E_ii = R_ii / 2 counts (the norm inflation |E|^2 / 2 of the previous step is removed by this one); |E|, not the
                                                                      // off-diagonal mass, decides (a pair with a tiny gap rotates by off / gap)
    }
    return 2;
}

// The same loop with T, D, R materialised in LDS (any k; six barriers per step): used for k > 32 only.
template <int NTH>
__device__ __forceinline__ int psd_refine_loop_lds(int k, int P, const double *Sm, double *&Va, double *&Vb, double *Tm, double *Dm, double *ev,
                                                    double *red, unsigned long long *stats, int refine, int maxl) {
    constexpr int NT = NTH, NW = NTH / 64;
    const int tid = threadIdx.x;
    const float rk = 1.0f / (float)k;
    double prev_off = 0;
    for (int it = 0; it < maxl; it++) {
        psd_gemm_kk<NTH>(k, Sm, P, 1, Va, P, 1, nullptr, [&](int M, int N, double v) { Tm[M * P + N] = v; });                                  // T = S Va
        psd_gemm_kk<NTH>(k, Va, 1, P, Va, P, 1, nullptr, [&](int M, int N, double v) { Vb[M * P + N] = (M == N ? 1.0 : 0.0) - v; });           // R = I - Va^T Va
        __syncthreads();
        psd_gemm_kk<NTH>(k, Va, 1, P, Tm, P, 1, nullptr, [&](int M, int N, double v) { Dm[M * P + N] = v; });                                  // D = Va^T T
        __syncthreads();
        double r[5] = {0, 0, 0, 0, 0};                          // off^2, diag^2 (sums) ; clipped, max |R|, max |E| (max)
        for (int i = tid; i < k; i += NT) {
            const double dii = Dm[i * P + i], rii = Vb[i * P + i], e = 0.5 * rii;
            Tm[i * P + i] = e; ev[i] = dii * psd_rcp(1.0 - rii);
            r[1] = fma(dii, dii, r[1]); r[3] = fmax(r[3], fabs(rii)); r[4] = fmax(r[4], fabs(e));
        }
        const int hr = (k + 1) >> 1;                            // pairs i < j, one work item each: rows i and k - 1 - i folded into one row of k items
        for (int idx = tid; idx < hr * k; idx += NT) {
            const int i0 = psd_fdiv(idx, rk), j0 = idx - i0 * k;
            int i, j;
            if (j0 > i0) { i = i0; j = j0; } else if (j0 < i0 && 2 * i0 != k - 1) { i = k - 1 - i0; j = k - 1 - j0; } else continue;
            const double sv = 0.5 * (Dm[i * P + j] + Dm[j * P + i]), rv = 0.5 * (Vb[i * P + j] + Vb[j * P + i]);
            const double dii = Dm[i * P + i], djj = Dm[j * P + j], rii = Vb[i * P + i], rjj = Vb[j * P + j];
            const double li = fma(dii, rii, dii), lj = fma(djj, rjj, djj);
            const double den = lj - li, nij = fma(lj, rv, sv), nji = fma(li, rv, sv), lim = 0.8 * fabs(den);
            double eij, eji;
            if (!(fabs(nij) <= lim) || !(fabs(nji) <= lim) || den == 0.0) { eij = eji = 0.5 * rv; if (nij != 0.0 || nji != 0.0) r[2] = 1.0; }
            else { const double inv = psd_rcp(den); eij = nij * inv; eji = -nji * inv; r[4] = fmax(r[4], fmax(fabs(eij), fabs(eji))); }
            Tm[i * P + j] = eij; Tm[j * P + i] = eji;
            r[0] = fma(2.0 * sv, sv, r[0]); r[3] = fmax(r[3], fabs(rv));
        }
        block_reduce_n<5, NW>(r, 0x1Cu, red);
        const double off2 = r[0], tot = r[0] + r[1];
        if (it == 0 && r[3] > 1e-6) return 2;
        if (off2 <= 1e-28 * tot && r[3] <= 1e-14) return 0;
        if (r[2] != 0.0 || !refine || (it > 0 && !(off2 <= 0.25 * prev_off))) return (it == 0 && r[3] <= 1e-10) ? 1 : 2;
        prev_off = off2;
        psd_gemm_kk<NTH>(k, Va, P, 1, Tm, P, 1, nullptr, [&](int M, int N, double v) { Vb[M * P + N] = Va[M * P + N] + v; });
        __syncthreads();
        { double *t_ = Va; Va = Vb; Vb = t_; }
        if (stats && tid == 0) atomicAdd(&stats[1], 1ull);
        if (r[4] <= 1e-7) return 0;
    }
    return 2;
}

// zsvec (svec of S) is replaced by svec(Pi_PSD(S)).  Vst: k x k (pitch P) eigenvectors of the previous call of THIS block, kept by the caller between
// calls (LDS); warm == 0: no previous call.  Sm, Tm, Dm, Rm: k x P scratch each; cs: 4 k + 16 doubles; red: block_reduce scratch.
// Strategy: refine the previous decomposition; when that is not possible (first call, a step outside the basin) run Jacobi sweeps only until the
// off-diagonal mass is 1e-3 of the total and let the refinement polish (quadratic, ~1 k cycles per step against ~27 k per sweep); if even that
// leaves the basin (clustered eigenvalues), sweep to full accuracy.
// stats (debug, may be null): [0] projections, [1] refinement steps, [2] warm Jacobi fall-backs, [3] cold starts, [6] ticks in the sweeps, [8..] ticks per phase.
template <int NTH>
__device__ __forceinline__ void psd_project_refine(double *zsvec, int k, double *Vst, double *Sm, double *Tm, double *Dm, double *Rm, double *cs,
                                                   double *red, int warm, unsigned long long *stats = nullptr, int refine = 1) {
    constexpr int NT = NTH, NW = NTH / 64, MAXL = 8;
    const int tid = threadIdx.x, P = psd_refine_pitch(k);
    const float rk = 1.0f / (float)k;
    double *ev = cs + 2 * (k + 2), *lam = ev + k;                   // eigenvalues, lam of the fused loop (cs[0 ..]: rotation parameters of the Jacobi sweeps)
    const bool fused = k <= 20 && NW >= ((k + 15) >> 4) * ((k + 15) >> 4);
    for (int idx = tid; idx < k * k; idx += NT) {
        const int i = psd_fdiv(idx, rk), j = idx - i * k;
        const int a = i >= j ? i : j, b = i >= j ? j : i;           // lower-triangle entry (a, b), column-major packed
        const double v = zsvec[b * k - (b * (b - 1)) / 2 + (a - b)];
        Sm[i * P + j] = (a == b) ? v : v * M_SQRT1_2;
    }
    __syncthreads();
    double *Va = Vst, *Vb = Rm;
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    int res = warm ? -1 : 2;                                        // -1: try to refine the caller's V
    for (int attempt = 0;; attempt++) {
        if (res < 0 || (res == 3 && refine)) {
            Va = Vst; Vb = Rm;
            res = fused ? psd_refine_loop_fused<NTH>(k, P, Sm, Va, Vb, Tm, Dm, lam, ev, red, stats, refine, MAXL, tk)
                        : psd_refine_loop_lds<NTH>(k, P, Sm, Va, Vb, Tm, Dm, ev, red, stats, refine, MAXL);
            if (res == 0) break;
        }
        if (res == 2) {                                             // cold start: V = I, D = S
            for (int idx = tid; idx < k * k; idx += NT) { const int i = psd_fdiv(idx, rk), j = idx - i * k; Vst[i * P + j] = (i == j) ? 1.0 : 0.0; Dm[i * P + j] = Sm[i * P + j]; }
        } else {                                                    // warm Jacobi: D = V^T S V, exactly symmetric for the rotations
            for (int idx = tid; idx < k * k; idx += NT) {
                const int i = psd_fdiv(idx, rk), j = idx - i * k;
                if (i > j) { const double a = 0.5 * (Dm[i * P + j] + Dm[j * P + i]); Dm[i * P + j] = a; Dm[j * P + i] = a; }
            }
        }
        if (stats && tid == 0) atomicAdd(&stats[res == 2 ? 3 : 2], 1ull);
        __syncthreads();
        Va = Vst;
        const bool coarse = refine && attempt == 0;                 // first fall-back: sweep into the basin of the refinement only
#ifdef CE_PSD_TIMING
        const long long tj0 = stats ? clock64() : 0;
#endif
        psd_sweeps_wg<NTH>(Dm, Vst, k, P, cs, red, coarse ? 1e-6 : 0.0);
#ifdef CE_PSD_TIMING
        if (stats && tid == 0) atomicAdd(&stats[6], (unsigned long long)(clock64() - tj0));
#endif
        if (coarse) { res = 3; continue; }                          // (the sweeps end with the barriers of their last reduction)
        for (int i = tid; i < k; i += NT) ev[i] = Dm[i * P + i];
        __syncthreads();
        break;
    }
    if (stats && tid == 0) atomicAdd(&stats[0], 1ull);
#ifdef CE_PSD_TIMING
    long long tph = stats ? clock64() : 0;
#endif
    // X = (Va diag(w+)) Va^T, written straight into the svec (lower triangle; the two triangles of the product differ by rounding only, and S
    // itself was copied out of zsvec at the start); the refined eigenvectors return to the block's own buffer in the same phase
    psd_gemm_kk<NTH>(k, Va, P, 1, Va, 1, P, ev, [&](int M, int N, double v) { if (M >= N) zsvec[N * k - (N * (N - 1)) / 2 + (M - N)] = (M == N) ? v : v * M_SQRT2; });
    if (Va != Vst) for (int idx = tid; idx < k * k; idx += NT) { const int i = psd_fdiv(idx, rk), j = idx - i * k; Vst[i * P + j] = Va[i * P + j]; }
    __syncthreads();
    PSD_TICK(13);
#ifdef CE_PSD_TIMING
    if (stats && tid == 0) for (int q = 0; q < 6; q++) atomicAdd(&stats[8 + q], (unsigned long long)tk[q]);
#endif
}

// LDS doubles needed: 3 * KP * (KP + 1) + 2 * k + 8  (+ the reduction scratch of block_reduce_n)
__host__ __device__ inline int psd_mfma_kp(int k) { return 16 * ((k + 15) / 16); }

// zsvec (svec of S, lower triangle column-major, sqrt(2) off-diagonals) is replaced by svec(Pi_PSD(S)).
// Vstate: k * k doubles of global memory, the eigenvectors of the previous call (row-major), or NULL; warm != 0: start from them.
// VM_IS_STATE: the eigenvectors of the previous call are still in Vm (a persistent kernel keeps them in LDS): warm != 0 starts from them.
template <int NTH, bool VM_IS_STATE = false>
__device__ __forceinline__ void psd_project_mfma(double *zsvec, int k, double *Sm, double *Vm, double *Tm, double *cs, double *red,
                                                 double *Vstate, int warm) {
    constexpr int NT = NTH;
    const int tid = threadIdx.x;
    const int KP = psd_mfma_kp(k), P = KP + 1, KT = KP / 16;
    const bool use_prev = warm && (VM_IS_STATE || Vstate != nullptr);
    const float rKP = 1.0f / (float)KP, rk = 1.0f / (float)k;
    for (int idx = tid; idx < KP * KP; idx += NT) {
        const int i = psd_fdiv(idx, rKP), j = idx - i * KP;
        double sv = 0.0, vv = 0.0;
        if (i < k && j < k) {
            const int a = i >= j ? i : j, b = i >= j ? j : i;                 // lower-triangle entry (a, b), column-major packed
            const double v = zsvec[b * k - (b * (b - 1)) / 2 + (a - b)];
            sv = (a == b) ? v : v * M_SQRT1_2;
            if constexpr (VM_IS_STATE) vv = use_prev ? Vm[i * P + j] : (i == j ? 1.0 : 0.0);
            else vv = use_prev ? Vstate[i * k + j] : (i == j ? 1.0 : 0.0);
        }
        Sm[i * P + j] = sv; Vm[i * P + j] = vv;
    }
    __syncthreads();
    if (use_prev) {
        // T = S V  (S symmetric: its A operand is read along rows);  S' = V^T T
        psd_mfma_gemm<NTH>(KT, [&](int M, int K) { return Sm[K * P + M]; }, [&](int K, int N) { return Vm[K * P + N]; },
                           [&](int M, int N, double v) { Tm[M * P + N] = v; });
        __syncthreads();
        psd_mfma_gemm<NTH>(KT, [&](int M, int K) { return Vm[K * P + M]; }, [&](int K, int N) { return Tm[K * P + N]; },
                           [&](int M, int N, double v) { Sm[M * P + N] = v; });
        __syncthreads();
        for (int idx = tid; idx < k * k; idx += NT) {      // exact symmetry for the rotations (each pair handled by its lower-triangle thread)
            const int i = psd_fdiv(idx, rk), j = idx - i * k;
            if (i > j) { const double a = 0.5 * (Sm[i * P + j] + Sm[j * P + i]); Sm[i * P + j] = a; Sm[j * P + i] = a; }
        }
        __syncthreads();
    }
    psd_sweeps_wg<NTH>(Sm, Vm, k, P, cs, red);
    for (int i = tid; i < KP; i += NT) cs[i] = i < k ? fmax(Sm[i * P + i], 0.0) : 0.0;
    __syncthreads();
    // X = (V diag(w+)) V^T
    psd_mfma_gemm<NTH>(KT, [&](int M, int K) { return Vm[M * P + K] * cs[K]; }, [&](int K, int N) { return Vm[N * P + K]; },
                       [&](int M, int N, double v) { Tm[M * P + N] = v; });
    __syncthreads();
    for (int idx = tid; idx < k * k; idx += NT) {          // lower triangle (a >= b) -> svec position b k - b (b - 1) / 2 + (a - b)
        const int a = psd_fdiv(idx, rk), b = idx - a * k;
        if (a < b) continue;
        const double v = 0.5 * (Tm[a * P + b] + Tm[b * P + a]);
        zsvec[b * k - (b * (b - 1)) / 2 + (a - b)] = (a == b) ? v : v * M_SQRT2;
    }
    if (Vstate != nullptr)
        for (int idx = tid; idx < k * k; idx += NT) { const int i = psd_fdiv(idx, rk), j = idx - i * k; Vstate[idx] = Vm[i * P + j]; }
    __syncthreads();
}

// the same with the eigenvector state resident in LDS (Vm keeps its content between calls)
template <int NTH>
__device__ __forceinline__ void psd_project_mfma_lds(double *zsvec, int k, double *Sm, double *Vm, double *Tm, double *cs, double *red, int warm) {
    psd_project_mfma<NTH, true>(zsvec, k, Sm, Vm, Tm, cs, red, nullptr, warm);
}
