// ce_shared_a_fwd.h -- SHARED-A forward: the whole SCS-style solve of one instance inside ONE persistent kernel (one workgroup per instance).
//
// Templates whose A does not depend on the parameters (only b, c vary: BASELINE configurations 4 and 5) and whose A consists of a few
// "dense" rows plus rows with a single entry (bounds, identity blocks of PSD / SOC embeddings):   A-hat = [ A_s (singleton rows) ; A_d (r rows) ].
// Then the reduced KKT matrix is diagonal plus rank r,
//        S = rho_x I + A-hat^T Dy A-hat = Dg + A_d^T Dd A_d ,      Dg_j = rho_x + scale * gs_j ,
// and its inverse is applied by the Woodbury identity with ONE r x r matrix per instance,
//        S^-1 t = u - Dg^-1 A_d^T K^-1 A_d u ,   u = Dg^-1 t ,   K = Dd^-1 + A_d Dg^-1 A_d^T ,
// instead of round 1's dense n x n eigenvector products (two rocBLAS GEMMs over the batch per iteration, ~10 launches and a host
// synchronisation per check interval).  K is formed on the matrix cores (v_mfma_f64_16x16x4_f64) from the transposed dense rows
// A_d^T (n x RP, row-major, shared by all instances: coalesced reads that hit L2) and inverted in LDS; all iterates of the instance
// live in LDS; termination, certificates and the adaptive scale run in the kernel (no host round trip); the PSD cone is projected by
// the warm-started MFMA routine of ce_psd_mfma.h with the eigenvectors kept in LDS between iterations.
// Algorithm, constants and order of operations: oracle/cone_oracle.c (SCS 3 restated) / cvxpylayers_amd/interfaces/const_a.py.
// Cones: zero / nonnegative / second-order / PSD / exponential / 3-d power.
#pragma once
#include "ce_shared_a_ops.h"

#ifdef CE_TIMING   // debug build: shader cycles per phase of the iteration (thread 0, accumulated in registers), written over the first entries of the instance's s row
#define SA_T(k) do { const long long t1_ = __builtin_readcyclecounter(); sa_tacc[k] += t1_ - sa_t0; sa_t0 = t1_; } while (0)
#else
#define SA_T(k) do { } while (0)
#endif

struct SaFwd {
    int r, RP;                   // dense rows, padded to a multiple of 16
    const double *AdT;           // [n][RP]  equilibrated dense rows, transposed (solver sign), zero padded
    const int *drow;             // [r]      row index of dense row a
    const int *srow_col;         // [m]      column of a singleton row, -1 otherwise
    const double *srow_val;      // [m]      its (equilibrated, solver-sign) value
    const int *scol_ptr;         // [n + 1]  singleton rows of every column
    const int *scol_row;         // [#singleton entries]
    const double *gs;            // [n]      sum over the singleton rows of column j of d0_i a_i^2
    const double *Dv, *Ev;       // [m], [n] equilibration
    unsigned long long *psd_stats;   // debug (CE_PSD_STATS=1): projections / refinement steps / warm Jacobi fall-backs / cold starts, or null
    double *aa_ws;                   // Anderson acceleration history, [B][4][lp] doubles of global memory (x_prev, f_prev, f_save, [w_prev when it does not fit LDS]; read once
                                     // per acceleration_interval iterations), or null: plain iteration
    int aa_w_lds;                    // the input of the last iteration (w_prev: read by the safeguard, written on two of ten iterations) lives in LDS
    int psd_refine;                  // 1: eigen-refinement on the matrix cores (default); 0 (CE_PSD_REFINE=0): warm-started Jacobi sweeps only, restart at check iterations (round 2)
};

// LDS doubles (see the carve in the kernel)
__host__ __device__ inline size_t sa_fwd_cidx_doubles(int n, int m, int nq, int r, int nsing) {
    return 2 * (size_t)m + ((size_t)(n + 1 + (nsing > 0 ? nsing : 1)) + 1) / 2 + ((size_t)(nq + 1 + (r > 0 ? r : 1)) + 1) / 2 + 2;
}
__host__ __device__ inline size_t sa_fwd_lds_doubles(int n, int m, int nq, int ns, int maxs, int RP, int nth, int ntri = 0) {
    const int l = n + m + 1, lp = l + (l & 1), ne = n + (n & 1), me = m + (m & 1);
    const size_t psd = ns > 0 ? (size_t)ns * maxs * psd_refine_pitch(maxs) + psd_refine_scratch_doubles(maxs) : 0;      // V per block + shared scratch (ce_psd_mfma.h)
    return 6 * (size_t)lp + 2 * (size_t)ne + 2 * (size_t)me + 2 * (size_t)RP * (RP + 1) + 5 * (size_t)RP + 2 * (size_t)(nq > 0 ? nq : 1) +
           psd + (psd & 1) + 2 * nth + (nth / 64) * 8 + 32 + (size_t)(ntri + (ntri & 1));
}

// NTH threads per instance: 256 (two instances per CU when the iterates allow it) or 512 (templates whose iterates fill most of a CU's LDS
// anyway: eight waves keep more loads of the shared matrix in flight and shorten every elementwise pass).
// CIDX: the template's small index arrays (singleton structure, cone offsets) are copied to LDS once -- every iteration reads them, and from global
// memory each dependent lookup is an exposed L2 round trip when a single workgroup owns the CU.
// HTRI false: the template has no exponential / power triples (their projection code is compiled out of the instantiation BASELINE config 5 runs)
template <int RP, int NTH, bool CIDX = false, bool HTRI = true>
__global__ void __launch_bounds__(NTH, NTH == 256 ? 2 : 1)
k_sa_fwd(DevT T, SaFwd F, ce_settings S, const double *__restrict__ BHg, const double *__restrict__ CHg, const double *__restrict__ sigma_g,
         const double *__restrict__ nb0_g, const double *__restrict__ nc0_g, const double *__restrict__ warm_x, const double *__restrict__ warm_y,
         const double *__restrict__ warm_s, double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ so,
         int *__restrict__ iters_o, int *__restrict__ status_o, double *__restrict__ resid_o) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, inst = blockIdx.x;
    const int n = T.n, m = T.m, l = n + m + 1, lp = l + (l & 1), z = T.z, nl = T.l, nq = T.nq, ns = T.ns;
    const int r = F.r, LK = RP + 1;
    constexpr int NT = NTH, NW = NTH / 64;                  // (shadow the engine-wide 256-thread constants)
    const int ne = n + (n & 1), me = m + (m & 1);           // even strides: every LDS vector below starts 16-byte aligned
    double *p = sm;
    double *W = p; p += lp; double *UT = p; p += lp; double *U = p; p += lp; double *G = p; p += lp; double *PHI = p; p += lp; double *zb = p; p += lp;
    double *tv = zb;                                        // refresh / check scratch: the cone-input vector is free there
    double *px = p; p += ne; double *dgi = p; p += ne;
    double *qy = p; p += me; double *bh = p; p += me;
    double *Kinv = p; p += (size_t)RP * LK;
    double *K0 = p; p += (size_t)RP * LK;                   // A_d Dg^-1 A_d^T (kept next to its regularised inverse: see the iteration)
    double *vd = p; p += RP; double *zd = p; p += RP; double *wyd = p; p += RP; double *dyd = p; p += 2 * RP;      // dyd[RP ..]: w_d + z of the iteration
    double *socc = p; p += 2 * (nq > 0 ? nq : 1);
    const int PM = ns > 0 ? T.maxs * psd_refine_pitch(T.maxs) : 0;     // compact k x k storage (ce_psd_mfma.h, psd_project_refine)
    double *Vst = p; p += (size_t)ns * PM;                 // eigenvectors of every PSD block, kept between iterations
    double *Sm = p; p += PM; double *Tm = p; p += PM; double *Dm = p; p += PM; double *Rm = p; p += PM;      // PSD scratch: S, T / E, V^T S V, I - V^T V
    double *cs = p; p += (ns > 0 ? 4 * T.maxs + 16 : 0);
    { const size_t used = ns > 0 ? (size_t)ns * PM + psd_refine_scratch_doubles(T.maxs) : 0; p += used & 1; }       // keep the vectors below 16-byte aligned
    double *part = p; p += 2 * NT;                              // partial sums of the dense-row products
    double *red = p; p += NW * 8;
    double *sc = p; p += 32;
    const int ntri = HTRI ? T.nep + T.np : 0;
    double *troot = p; p += ntri + (ntri & 1);              // exponential / power triples: root of the previous projection (warm start of the Newton iteration)
    double *aaW_lds = p; if (F.aa_w_lds) p += lp;           // Anderson acceleration: input of the last iteration
    const int *c_srow_col = F.srow_col, *c_rowcone = T.rowcone, *c_scol_ptr = F.scol_ptr, *c_scol_row = F.scol_row, *c_qoff = T.qoff, *c_drow = F.drow;
    const double *c_srow_val = F.srow_val;
    if constexpr (CIDX) {
        const int nsing = F.scol_ptr[n];
        double *sv = p; p += m;
        int *ip = reinterpret_cast<int *>(p);
        int *i_col = ip; ip += m; int *i_cone = ip; ip += m; int *i_ptr = ip; ip += n + 1; int *i_row = ip; ip += (nsing > 0 ? nsing : 1); int *i_qoff = ip; ip += nq + 1; int *i_drow = ip;
        for (int i = tid; i < m; i += NT) { sv[i] = F.srow_val[i]; i_col[i] = F.srow_col[i]; i_cone[i] = T.rowcone[i]; }
        for (int j = tid; j <= n; j += NT) i_ptr[j] = F.scol_ptr[j];
        for (int k = tid; k < nsing; k += NT) i_row[k] = F.scol_row[k];
        for (int c = tid; c <= nq; c += NT) i_qoff[c] = T.qoff[c];
        for (int a = tid; a < r; a += NT) i_drow[a] = F.drow[a];
        c_srow_val = sv; c_srow_col = i_col; c_rowcone = i_cone; c_scol_ptr = i_ptr; c_scol_row = i_row; c_qoff = i_qoff; c_drow = i_drow;
    }
    const double *ch = CHg + (size_t)inst * n;              // c-hat stays in global memory (read in refresh / checks only)
    const double rho_x = S.rho_x, rtau = TAU_FACTOR, alpha = S.alpha;
    // loop-carried values that are equal in every lane live in scalar registers (uniform_d = readfirstlane, ce_forward_v2.h): the 256-thread
    // instantiation sits at the 256-VGPR ceiling and every VGPR held across the iteration loop is one more scratch reload per iteration
    const double sigma = uniform_d(sigma_g[inst]), isg = uniform_d(1.0 / sigma);
    double scale = S.scale;
    auto dyv = [&](int i) -> double { return (i < z) ? ZERO_CONE_FACTOR * scale : scale; };

    for (int i = tid; i < m; i += NT) bh[i] = BHg[(size_t)inst * m + i];
    for (int e = tid; e < lp; e += NT) { W[e] = 0.0; UT[e] = 0.0; U[e] = 0.0; }
    for (int c = tid; c < ntri; c += NT) troot[c] = 0.0;
    ce_math_table_init(sc + 12, tid);                       // sc[12 ..]: ce_math.h coefficient table (CE_MATH_TAB = 18 <= 20)
    __syncthreads();
    const double *mtab = sc + 12;

    // ---------------- products with the shared matrix (ce_shared_a_ops.h)
    auto dense_times = [&](const double *xin, double *out) { sa_dense_times<NT, RP>(F.AdT, n, xin, part, out); };
    auto A_times = [&](const double *xin, auto &&out) { sa_A_times<NT, RP>(F, n, m, xin, part, vd, out); };
    auto AT_times = [&](const double *yin, auto &&out) { sa_AT_times<NT, RP>(F, n, yin, wyd, out); };
    // pout <- S^-1 t  for pout = u = Dg^-1 t on entry (Woodbury);  zd = K^-1 A_d u is left behind
    auto wood_u = [&](double *pout) {
        dense_times(pout, vd);
        for (int a = tid >> 3; a < RP; a += NT / 8) {       // zd = K^-1 vd : eight lanes per row
            const double *kr = Kinv + a * LK;
            double s_ = 0;
            {      // this lane's RP / 8 products with their operands requested together (the run-time bound b < r made it a chain of r / 8 LDS round trips; the padding of K and of the vector is zero)
                double kv[RP / 8], xv[RP / 8];
            #pragma unroll
                for (int bb = 0; bb < RP / 8; bb++) { const int b = (tid & 7) + 8 * bb; kv[bb] = kr[b]; xv[bb] = vd[b]; }
            #pragma unroll
                for (int bb = 0; bb < RP / 8; bb++) s_ = fma(kv[bb], xv[bb], s_);
            }
            s_ = group_reduce<8, false>(s_);
            if ((tid & 7) == 0) zd[a] = a < r ? s_ : 0.0;
        }
        __syncthreads();
        sa_rows_dot<NT, RP>(F.AdT, n, zd, [&](int, int) { return 0.0; }, [&](int j, double a) { pout[j] -= dgi[j] * a; });
        __syncthreads();
    };
    // pout = S^-1 tin.  tin / pout may alias.
    auto wood = [&](const double *tin, double *pout) {
        for (int j = tid; j < n; j += NT) pout[j] = tin[j] * dgi[j];
        __syncthreads();
        wood_u(pout);
    };

    // ---------------- (re)factor for the current scale: Dg^-1, K^-1 (MFMA + Gauss-Jordan in LDS), then g, h.g, phi
    double hg = 0, inv_den = 0;
    auto refresh = [&]() {
        for (int j = tid; j < n; j += NT) dgi[j] = 1.0 / (rho_x + scale * F.gs[j]);
        for (int a = tid; a < RP; a += NT) dyd[a] = a < r ? dyv(F.drow[a]) : 1.0;
        __syncthreads();
        {   // K = Dd^-1 + A_d Dg^-1 A_d^T : tiles of 16 x 16 over the waves, operands straight from the shared A_d^T (coalesced, L2)
            typedef double v4d __attribute__((ext_vector_type(4)));
            const int KT = RP / 16, wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lc = lane & 15;
            for (int t = wave; t < KT * KT; t += NW) {
                const int ti = t / KT, tj = t - ti * KT;
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                for (int j0 = 0; j0 < n; j0 += 4) {
                    const int j = j0 + lg;
                    double a = 0.0, b = 0.0;
                    if (j < n) { const double *row = F.AdT + (size_t)j * RP; a = row[16 * ti + lc] * dgi[j]; b = row[16 * tj + lc]; }
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = 16 * ti + lg + 4 * q, col = 16 * tj + lc;
                    K0[row * LK + col] = acc[q];
                    Kinv[row * LK + col] = acc[q] + (row == col ? (row < r ? 1.0 / dyd[row] : 1.0) : 0.0);
                }
            }
        }
        __syncthreads();
        for (int k = 0; k < r; k++) {       // in-place Gauss-Jordan inverse (K is symmetric positive definite: no pivoting)
            const double pinv = 1.0 / Kinv[k * LK + k];
            __syncthreads();
            for (int idx = tid; idx < r * r; idx += NT) {
                const int i = idx / r, j = idx - i * r;
                if (i != k && j != k) Kinv[i * LK + j] = fma(-Kinv[i * LK + k] * pinv, Kinv[k * LK + j], Kinv[i * LK + j]);
            }
            __syncthreads();
            for (int i = tid; i < r; i += NT) {
                if (i != k) { const double ck = Kinv[i * LK + k], rk = Kinv[k * LK + i]; Kinv[i * LK + k] = -ck * pinv; Kinv[k * LK + i] = rk * pinv; }
                else Kinv[k * LK + k] = pinv;
            }
            __syncthreads();
        }
        // a = A^T (Dy b) ;  gx = S^-1 (c - a) ;  pk = S^-1 (c + a) ;  gy = Dy (A gx + b) ;  hg = c.gx + b.gy ;  phi = (rho pk ; b - A pk)
        for (int i = tid; i < m; i += NT) qy[i] = dyv(i) * bh[i];
        __syncthreads();
        AT_times(qy, [&](int j, double a) { const double cj = ch[j]; tv[j] = cj - a; px[j] = cj + a; });
        wood(tv, tv);                   // gx
        wood(px, px);                   // pk
        A_times(tv, [&](int i, double a) { G[n + i] = dyv(i) * (a + bh[i]); });
        A_times(px, [&](int i, double a) { PHI[n + i] = bh[i] - a; });
        double rr[1] = {0};
        for (int j = tid; j < n; j += NT) { G[j] = tv[j]; PHI[j] = rho_x * px[j]; rr[0] = fma(ch[j], tv[j], rr[0]); }
        for (int i = tid; i < m; i += NT) rr[0] = fma(bh[i], G[n + i], rr[0]);
        block_reduce_n<1, NW>(rr, 0u, red);
        hg = uniform_d(rr[0]); inv_den = uniform_d(1.0 / (rtau + hg));
        if (tid == 0) { G[l - 1] = 0.0; PHI[l - 1] = 0.0; }
        __syncthreads();
    };

    if (tid == 0) W[l - 1] = 1.0;       // cold start w = (0, 0, 1)
    __syncthreads();
    if (S.warm_start && warm_x) {       // SCS warm start u = (x^, y^, 1), v = (0, s^, 0): w = u + R^-1 v in the equilibrated space
        double bad[1] = {0};
        for (int j = tid; j < n; j += NT) { const double v = sigma * warm_x[(size_t)inst * n + j] / F.Ev[j]; tv[j] = v; if (!(fabs(v) < 1e300)) bad[0] = 1.0; }
        for (int i = tid; i < m; i += NT) {
            const double dvi = F.Dv[i];
            const double v = sigma * warm_y[(size_t)inst * m + i] / dvi + sigma * dvi * warm_s[(size_t)inst * m + i] * dyv(i);
            qy[i] = v; if (!(fabs(v) < 1e300)) bad[0] = 1.0;
        }
        block_reduce_n<1, NW>(bad, 1u, red);
        if (bad[0] == 0.0) { for (int j = tid; j < n; j += NT) W[j] = tv[j]; for (int i = tid; i < m; i += NT) W[n + i] = qy[i]; }
        __syncthreads();
    }

    int status = 0, iter = 0, last_scale_iter = 0, n_log = 0;
    double sum_log = 0;
    bool done = false, resume = false;
    // Anderson acceleration of the iteration map w -> F(w): type I, one secant pair, residual safeguard, switched off after AA_MAX_REJECT rejections --
    // the algorithm of k_fwd2 (ce_forward_v2.h) and of the oracle with aa_mem = 1.  Every acceleration_interval iterations, with x = input and
    // f = output of the last iteration, g = x - f, s = x - x_prev, y = g - g_prev, d = f - f_prev:  w <- f - (s.g / (s.y + 1e-8 |s||y|)) d;  the
    // next iteration's residual is the safeguard.  The five history vectors live in GLOBAL memory (L2): they are touched on two of every ten
    // iterations only, and LDS is what limits the residency of this kernel.
    bool aa_on = S.acceleration_lookback > 0 && F.aa_ws != nullptr, aa_pending = false, aa_stale = false;
    const int aa_int = S.acceleration_interval > 0 ? S.acceleration_interval : 10;
    int aa_iter = 0, aa_rej = 0;
    double aa_normg = 0, aa_hs = 1.0;     // |g| before the step ; factor by which the stored history has to be scaled (the renormalisations of w since it was stored)
    double *const aaXP = F.aa_ws ? F.aa_ws + (size_t)inst * 4 * lp : nullptr, *const aaFP = aaXP + lp, *const aaFS = aaFP + lp;
    double *const aaWP = F.aa_w_lds ? aaW_lds : aaFS + lp;      // (generic pointer: LDS when it fits, else the fourth global vector)
    double res3[3] = {NAN, NAN, NAN};
#ifdef CE_TIMING
    long long sa_tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sa_t0 = __builtin_readcyclecounter();
#endif
    while (!done) {
        refresh();
        SA_T(0);
        if (resume) {      // relaxed update owed by the iteration a rescale interrupted
            for (int e = tid; e < l; e += NT) W[e] += alpha * (U[e] - UT[e]);
            __syncthreads();
            resume = false; iter++;
        }
        for (;;) {
            if (iter >= S.max_iters) { done = true; break; }
            const bool check = (iter % CONVERGED_INTERVAL) == 0, last = iter + 1 >= S.max_iters;
#ifdef CE_PSD_TIMING
            const long long t_iter0 = F.psd_stats ? clock64() : 0;
#endif
            if (aa_on) {
                if (aa_pending) {      // safeguard: residual of the map at the accelerated point against the residual before the step
                    double rs[1] = {0};
                    for (int e = tid; e < l; e += NT) { const double dd = aaWP[e] - W[e]; rs[0] = fma(dd, dd, rs[0]); }
                    block_reduce_n<1, NW>(rs, 0u, red);
                    if (!(sqrt(rs[0]) <= aa_normg)) {
                        for (int e = tid; e < l; e += NT) W[e] = aaFS[e] * aa_hs;
                        aa_iter = 0;
                        if (++aa_rej >= AA_MAX_REJECT) aa_on = false;
                        __syncthreads();
                    }
                    aa_pending = false;
                }
                if (aa_on && iter > 0 && iter % aa_int == 0 && !aa_stale) {      // (aa_stale: see k_fwd2)
                    if (aa_iter > 0) {
                        double rr[5] = {0, 0, 0, 0, 0};
                        for (int e = tid; e < l; e += NT) {
                            const double xv = aaWP[e], fv = W[e], gv = xv - fv, xp = aaXP[e] * aa_hs, fp = aaFP[e] * aa_hs;
                            const double sv = xv - xp, yv = gv - (xp - fp);
                            rr[0] = fma(sv, sv, rr[0]); rr[1] = fma(yv, yv, rr[1]); rr[2] = fma(sv, yv, rr[2]); rr[3] = fma(sv, gv, rr[3]); rr[4] = fma(gv, gv, rr[4]);
                        }
                        block_reduce_n<5, NW>(rr, 0u, red);
                        const double mm = rr[2] + 1e-8 * sqrt(rr[0]) * sqrt(rr[1]), gam = uniform_d(rr[3] / mm);
                        const bool ok = fabs(mm) > 1e-300 && fabs(gam) < 1e10;
                        for (int e = tid; e < l; e += NT) {                      // (second read of the history: L2-warm)
                            const double xv = aaWP[e], fv = W[e], fp = aaFP[e] * aa_hs;
                            aaXP[e] = xv; aaFP[e] = fv;
                            if (ok) { aaFS[e] = fv; W[e] = fv - gam * (fv - fp); }
                        }
                        aa_hs = 1.0;
                        if (ok) { aa_normg = uniform_d(sqrt(rr[4])); aa_pending = true; } else aa_iter = 0;
                    } else {
                        for (int e = tid; e < l; e += NT) { aaXP[e] = aaWP[e]; aaFP[e] = W[e]; }
                        aa_hs = 1.0;
                    }
                    aa_iter++;
                    __syncthreads();
                }
            }
            if (check && iter > 0) {       // keep the homogeneous iterate in range
                double rn[1] = {0};
                for (int e = tid; e < l; e += NT) rn[0] = fma(W[e], W[e], rn[0]);
                block_reduce_n<1, NW>(rn, 0u, red);
                const double nw = sqrt(rn[0]);
                if (nw > 0) {
                    const double f = sqrt((double)l) / nw;
                    for (int e = tid; e < l; e += NT) W[e] *= f;
                    if (aa_on) { aa_hs = uniform_d(aa_hs * f); aa_normg = uniform_d(aa_normg * f); }      // the map is positively homogeneous: the stored history scales with w (lazily)
                }
                __syncthreads();
            }
            if (aa_on && (aa_pending || (iter + 1) % aa_int == 0)) { aa_stale = false; for (int e = tid; e < l; e += NT) aaWP[e] = W[e]; }      // input of this iteration, where the next one needs it
            // tau-tilde needs phi.w: the partial sums ride on the barriers of the products below
            {
                double rt = 0;
                for (int e = tid; e < l - 1; e += NT) rt = fma(PHI[e], W[e], rt);
                rt = wave_reduce_dpp<false>(rt);
                if ((tid & 63) == 0) red[tid >> 6] = rt;
            }
            // p_x = S^-1 t,  t = rho w_x - A^T w_y,  with TWO passes over the shared dense rows instead of three.  Split t by the two row classes:
            //   t' = Dg^-1 (rho w_x - A_s^T w_y)  (singleton rows: a gather),   u = Dg^-1 t = t' - Dg^-1 A_d^T w_d,   w_d = the dense rows of w_y.
            //   A_d u = A_d t' - K0 w_d,  K0 = A_d Dg^-1 A_d^T (r x r, in LDS since the factorisation)        -> one pass  (A_d t')
            //   z = K^-1 A_d u ;   p_x = u - Dg^-1 A_d^T z = t' - Dg^-1 A_d^T (w_d + z)                        -> one pass  (A_d^T (w_d + z))
            for (int a = tid; a < RP; a += NT) wyd[a] = a < r ? W[n + c_drow[a]] : 0.0;
            for (int j = tid; j < n; j += NT) {
                double acc = rho_x * W[j];
                for (int k = c_scol_ptr[j]; k < c_scol_ptr[j + 1]; k++) { const int i = c_scol_row[k]; acc = fma(-c_srow_val[i], W[n + i], acc); }
                px[j] = acc * dgi[j];
            }
            __syncthreads();
            SA_T(1);
            sa_dense_partials<NT, RP>(F.AdT, n, px, part);
            for (int a = tid >> 3; a < RP; a += NT / 8) {       // K0 w_d : eight lanes per row
                const double *kr = K0 + a * LK;
                double s_ = 0;
                {      // this lane's RP / 8 products with their operands requested together (the run-time bound b < r made it a chain of r / 8 LDS round trips; the padding of K and of the vector is zero)
                    double kv[RP / 8], xv[RP / 8];
                #pragma unroll
                    for (int bb = 0; bb < RP / 8; bb++) { const int b = (tid & 7) + 8 * bb; kv[bb] = kr[b]; xv[bb] = wyd[b]; }
                #pragma unroll
                    for (int bb = 0; bb < RP / 8; bb++) s_ = fma(kv[bb], xv[bb], s_);
                }
                s_ = group_reduce<8, false>(s_);
                if ((tid & 7) == 0) zd[a] = s_;
            }
            __syncthreads();
            SA_T(2);
            if (tid < RP) {      // A_d u: the partial sums are requested together (the plain loop waited for every pair of them: eight LDS round trips in series with everybody else at the barrier)
                constexpr int ng = 2 * NT / RP, CHK = ng < 16 ? ng : 16;       // (in batches of at most 16: ng is 64 for the narrowest variant)
                double s_ = -zd[tid];
#pragma unroll
                for (int g0 = 0; g0 < ng; g0 += CHK) {
                    double pv[CHK];
#pragma unroll
                    for (int gg = 0; gg < CHK; gg++) pv[gg] = part[(g0 + gg) * RP + tid];
#pragma unroll
                    for (int gg = 0; gg < CHK; gg++) s_ += pv[gg];
                }
                vd[tid] = s_;
            }
            __syncthreads();
            for (int a = tid >> 3; a < RP; a += NT / 8) {       // z = K^-1 (A_d u)
                const double *kr = Kinv + a * LK;
                double s_ = 0;
                {      // this lane's RP / 8 products with their operands requested together (the run-time bound b < r made it a chain of r / 8 LDS round trips; the padding of K and of the vector is zero)
                    double kv[RP / 8], xv[RP / 8];
                #pragma unroll
                    for (int bb = 0; bb < RP / 8; bb++) { const int b = (tid & 7) + 8 * bb; kv[bb] = kr[b]; xv[bb] = vd[b]; }
                #pragma unroll
                    for (int bb = 0; bb < RP / 8; bb++) s_ = fma(kv[bb], xv[bb], s_);
                }
                s_ = group_reduce<8, false>(s_);
                if ((tid & 7) == 0) { const double zv = a < r ? s_ : 0.0; zd[a] = zv; dyd[RP + a] = wyd[a] + zv; }
            }
            __syncthreads();
            SA_T(3);
            sa_rows_dot<NT, RP>(F.AdT, n, dyd + RP, [&](int, int) { return 0.0; }, [&](int j, double a) { px[j] -= dgi[j] * a; });
            __syncthreads();
            SA_T(4);
            double rts = 0;
            {
                double rw[NW];          // (requested together: the plain sum waited for each pair)
#pragma unroll
                for (int w = 0; w < NW; w++) rw[w] = red[w];
#pragma unroll
                for (int w = 0; w < NW; w++) rts += rw[w];
            }
            const double tau_t = (rtau * W[l - 1] + rts) * inv_den;
            for (int e = tid; e < l; e += NT) {
                double ute, ze;
                const double we = W[e];
                if (e < n) { ute = px[e] - tau_t * G[e]; ze = 2 * ute - we; }
                else if (e < l - 1) {
                    const int i = e - n, c = c_srow_col[i];           // >= 0: column of a singleton row, -2 - a: dense row in slot a, -1: empty row
                    const double qi = c >= 0 ? c_srow_val[i] * px[c] : (c <= -2 ? zd[-2 - c] / dyd[-2 - c] : 0.0);      // dense rows: Dd^-1 z
                    ute = we + dyv(i) * qi - tau_t * G[e]; ze = 2 * ute - we;
                    if (i >= z && i < z + nl && ze < 0) ze = 0;
                } else { ute = tau_t; ze = fmax(0.0, 2 * tau_t - we); }
                UT[e] = ute; zb[e] = ze;
            }
            __syncthreads();
            SA_T(5);
            if (nq > 0) {
                for (int c = tid >> 6; c < nq; c += NW) {         // one wave per cone
                    const int r0 = n + c_qoff[c], r1 = n + c_qoff[c + 1];
                    const double t0 = zb[r0]; double nz = 0;
                    for (int k = r0 + 1 + (tid & 63); k < r1; k += 64) nz = fma(zb[k], zb[k], nz);
                    nz = sqrt(wave_reduce_dpp<false>(nz));
                    double c0, f;
                    if (r1 - r0 == 1) { c0 = fmax(t0, 0.0); f = 0.0; }
                    else if (nz <= t0) { c0 = t0; f = 1.0; }
                    else if (nz <= -t0) { c0 = 0.0; f = 0.0; }
                    else { c0 = 0.5 * (t0 + nz); f = c0 / nz; }
                    if ((tid & 63) == 0) { socc[2 * c] = c0; socc[2 * c + 1] = f; }
                }
                __syncthreads();
                if (ns > 0) {      // the PSD projection below works on zb in place: finish the second-order cones first
                    for (int i = tid + z + nl; i < m; i += NT) { const int c = c_rowcone[i]; if (c >= 0) zb[n + i] = (i == c_qoff[c]) ? socc[2 * c] : socc[2 * c + 1] * zb[n + i]; }
                    __syncthreads();
                }
            }
            if constexpr (HTRI) if (ntri > 0) {    // exponential / power cone triples (after the PSD blocks): one thread per cone, in place (ce_expcone.h)
                for (int c = tid; c < ntri; c += NT) {
                    double *zc = zb + n + T.eoff + 3 * c;
                    if (c < T.nep) exp_project_dual(zc, troot + c); else pow_project_dual_of_entry(zc, T.pw[c - T.nep], troot + c);
                }
                __syncthreads();
            }
            // the projected value of element e (second-order cone rows are scaled here when there is no PSD block, saving a pass and a barrier)
            auto proj_e = [&](int e) -> double {
                double ue = zb[e];
                if (nq > 0 && ns == 0 && e >= n + z + nl && e < l - 1) { const int i = e - n, c = c_rowcone[i]; if (c >= 0) ue = (i == c_qoff[c]) ? socc[2 * c] : socc[2 * c + 1] * ue; }
                return ue;
            };
#ifndef SA_SKIP_PSD        // (debug builds time the kernel without the projection)
            if constexpr (NTH == 256) {        // (templates with PSD blocks always run the 256-thread instantiation)
#ifdef CE_PSD_TIMING
                const long long tp0 = (F.psd_stats && ns > 0) ? clock64() : 0;
#endif
                for (int c = 0; c < ns; c++)   // PSD blocks: warm-started eigen-refinement on the matrix cores (Jacobi sweeps as the fall-back), eigenvectors stay in LDS
                    psd_project_refine<NT>(zb + n + T.soff[c], T.sord[c], Vst + (size_t)c * PM, Sm, Tm, Dm, Rm, cs, red, (iter > 0 && (F.psd_refine || !check)) ? 1 : 0, F.psd_stats, F.psd_refine);
#ifdef CE_PSD_TIMING
                if (F.psd_stats && ns > 0 && tid == 0) { const long long t1 = clock64(); atomicAdd(&F.psd_stats[4], (unsigned long long)(t1 - tp0)); atomicAdd(&F.psd_stats[5], (unsigned long long)(t1 - t_iter0)); }
#endif
            }
#endif
            SA_T(6);
            if (!check && !last) {
                for (int e = tid; e < l; e += NT) { const double ue = proj_e(e); U[e] = ue; W[e] += alpha * (ue - UT[e]); }
                __syncthreads();
                SA_T(7);
                iter++;
                continue;
            }
            for (int e = tid; e < l; e += NT) U[e] = proj_e(e);
            __syncthreads();
            // ---- check iteration: residuals, termination, certificates, adaptive scale
            bool stop = false, rescale = false;
            if (check) {
                A_times(U, [&](int i, double a) { qy[i] = a; });                      // A x-hat
                AT_times(U + n, [&](int j, double a) { tv[j] = a; });                 // A^T y-hat
                const double tau = fabs(U[l - 1]);
                double rr[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rp, nax, ns, naxs, rd, naty (max) ; ctx, bty (sum)
                for (int i = tid; i < m; i += NT) {
                    const double sc_ = isg / F.Dv[i];
                    const double ax = qy[i] * sc_, uy = U[n + i];
                    const double sh = (uy + W[n + i] - 2 * UT[n + i]) / dyv(i) * sc_;
                    const double bt = bh[i] * tau * sc_;
                    rr[0] = fmax(rr[0], fabs(ax + sh - bt)); rr[1] = fmax(rr[1], fabs(ax)); rr[2] = fmax(rr[2], fabs(sh)); rr[3] = fmax(rr[3], fabs(ax + sh));
                    rr[7] += bh[i] * uy * isg * isg;
                }
                for (int j = tid; j < n; j += NT) {
                    const double sc_ = isg / F.Ev[j];
                    const double aty = tv[j] * sc_, cj = ch[j];
                    rr[4] = fmax(rr[4], fabs(aty + cj * tau * sc_)); rr[5] = fmax(rr[5], fabs(aty));
                    rr[6] += cj * U[j] * isg * isg;
                }
                block_reduce_n<8, NW>(rr, 0x3Fu, red);
                const double rp = rr[0], nax = rr[1], nsn = rr[2], naxs = rr[3], rd = rr[4], naty = rr[5], ctx = rr[6], bty = rr[7];
                const double nrm_b0 = nb0_g[inst], nrm_c0 = nc0_g[inst];
                if (tau > 0) {
                    const double res_pri = rp / tau, res_dual = rd / tau, gap = fabs(ctx + bty) / tau;
                    res3[0] = uniform_d(res_pri); res3[1] = uniform_d(res_dual); res3[2] = uniform_d(gap);
                    const double prl = fmax(fmax(nrm_b0 * tau, nsn), nax) / tau, drl = fmax(nrm_c0 * tau, naty) / tau;
                    const double grl = fmax(fabs(ctx), fabs(bty)) / tau;
                    if (res_pri <= S.eps_abs + S.eps_rel * prl && res_dual <= S.eps_abs + S.eps_rel * drl && gap <= S.eps_abs + S.eps_rel * grl) { status = 1; stop = true; }
                }
                if (!stop && bty < 0 && naty / (-bty) <= S.eps_infeas) { status = -2; stop = true; }
                if (!stop && ctx < 0 && naxs / (-ctx) <= S.eps_infeas) { status = -1; stop = true; }
                if (!stop && S.adaptive_scale && iter > 0) {
                    const double dp = fmax(fmax(nax, nsn), nrm_b0 * tau), dd = fmax(naty, nrm_c0 * tau);
                    const double rel_p = rp / (dp > 0 ? dp : 1), rel_d = rd / (dd > 0 ? dd : 1);
                    if (rel_p > 0 && rel_d > 0 && isfinite(rel_p) && isfinite(rel_d)) {
                        sum_log = uniform_d(sum_log + ce_log(rel_p, mtab) - ce_log(rel_d, mtab)); n_log++;
                        const double factor = ce_exp(0.5 * sum_log / n_log, mtab);
                        if (iter - last_scale_iter >= RESCALING_MIN_ITERS) {
                            const double ns2 = fmin(fmax(scale * factor, MIN_SCALE_VALUE), MAX_SCALE_VALUE);
                            if (ns2 != scale && (factor > sqrt(10.0) || factor < 1.0 / sqrt(10.0))) {
                                const double dy_ratio = ns2 / scale;       // keep (s, kappa):  w_y+ = rsk_y / r_y+ + 2 ut_y - u_y
                                for (int e = tid + n; e < l - 1; e += NT) { const double ue = U[e], ute = UT[e]; W[e] = (ue + W[e] - 2 * ute) * dy_ratio + 2 * ute - ue; }
                                n_log = 0; sum_log = 0; last_scale_iter = iter; scale = uniform_d(ns2); rescale = true; aa_iter = 0; aa_pending = false; aa_stale = true;
                                __syncthreads();
                            }
                        }
                    }
                }
            }
            if (stop) { done = true; break; }
            if (last) { iter++; done = true; break; }
            if (rescale) { resume = true; break; }
            for (int e = tid; e < l; e += NT) W[e] += alpha * (U[e] - UT[e]);
            __syncthreads();
            SA_T(8);
            iter++;
        }
    }
    __syncthreads();
    const double tau = fabs(U[l - 1]);
    if (status == 0) {   // ran out of iterations (SCS set_unfinished)
        const double kap = fabs(rtau * (U[l - 1] + W[l - 1] - 2 * UT[l - 1]));
        double rr[2] = {0, 0};
        for (int j = tid; j < n; j += NT) rr[0] += ch[j] * U[j] * isg * isg;
        for (int i = tid; i < m; i += NT) rr[1] += bh[i] * U[n + i] * isg * isg;
        block_reduce_n<2, NW>(rr, 0u, red);
        if (tau > kap) status = 2; else if (rr[1] < rr[0]) status = -7; else status = -6;
    }
    const bool solved = (status == 1 || status == 2), infeas = (status == -2 || status == -7);
    const double it = solved ? 1.0 / (sigma * tau) : 1.0 / sigma;
    for (int j = tid; j < n; j += NT) xo[(size_t)inst * n + j] = infeas ? NAN : F.Ev[j] * U[j] * it;
    for (int i = tid; i < m; i += NT) {
        const double uy = U[n + i], di = F.Dv[i];
        const double sh = (uy + W[n + i] - 2 * UT[n + i]) / dyv(i);
        yo[(size_t)inst * m + i] = (solved || infeas) ? di * uy * it : NAN;
        so[(size_t)inst * m + i] = infeas ? NAN : sh / di * it;
    }
#ifdef CE_TIMING
    __syncthreads();
    if (tid == 0) for (int k = 0; k < 12; k++) so[(size_t)inst * m + k] = (double)sa_tacc[k];
#endif
    if (tid == 0) {
        iters_o[inst] = iter; status_o[inst] = status;
        if (resid_o) { resid_o[3 * inst] = res3[0]; resid_o[3 * inst + 1] = res3[1]; resid_o[3 * inst + 2] = res3[2]; }
    }
}
