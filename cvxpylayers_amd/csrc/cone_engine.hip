// cone_engine.hip -- MI355X (gfx950 / CDNA4) batched cone-program solve + differentiate engine.
//
// One problem instance per 256-thread workgroup (4 wave64).  The instance's data (dense A in solver
// form, the explicit inverse G of the reduced KKT matrix, all iterates) live in LDS for the whole
// solve: HBM is touched once per instance on the way in (coalesced batch-major value rows) and once
// on the way out.  fp64 throughout (the reference returns float64, diffcp_if.py:374-375).
//
// Forward  : homogeneous self-dual embedding + Douglas-Rachford splitting (SCS 3 algorithm, restated
//            in oracle/cone_oracle.c which this file must agree with), with the per-iteration KKT
//            solve done as  t = rho_x w_x - A^T w_y ;  p_x = G t ;  p_y = w_y + Dy (A p_x)  where
//            G = (rho_x I + A^T Dy A)^{-1} is formed once per (re)scaling by in-LDS Gauss-Jordan.
//            Triangular solves are latency-bound on a GPU; an explicit inverse turns them into matvecs.
// Backward : diffcp's adjoint  M^T r = dz  solved DIRECTLY: the homogeneous embedding makes M singular
//            along z, dA/db/dc are invariant to that null component, so r_tau is pinned to 0 and the
//            remaining (n+m) system is reduced, cone block by cone block, with the spectral structure
//            of DPi (eigenvalue 1 -> equality row, 0 -> eliminated, lambda in (0,1) -> Schur term) to a
//            symmetric saddle system of size n + #equality rows, solved by Gauss-Jordan with partial
//            pivoting in LDS.  See DESIGN.md section "Backward".
//
// Reference boundary mirrored: cvxpylayers/interfaces/diffcp_if.py:46-96,329-403.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cone_engine.h"

namespace {

constexpr int NT = 256;            // threads per workgroup
constexpr int NW = NT / 64;        // waves per workgroup
constexpr int CONVERGED_INTERVAL = 25;
constexpr int RESCALING_MIN_ITERS = 100;
constexpr int NUM_RUIZ_PASSES = 25;
constexpr int NUM_L2_PASSES = 1;
constexpr double MIN_SCALE = 1e-4, MAX_SCALE = 1e4;
constexpr double MIN_SCALE_VALUE = 1e-6, MAX_SCALE_VALUE = 1e6;
constexpr double TAU_FACTOR = 10.0, ZERO_CONE_FACTOR = 1000.0;

struct DevT {
    int n, m, nnz_aug, nnzA, z, l, nq, lda, ldg;
    const int *rowidx;    // [nnz_aug] row of structural entry k
    const int *colidx;    // [nnz_aug] column (n == the b column)
    const int *rowcone;   // [m] -1 for zero / nonneg rows, else SOC index
    const int *qoff;      // [nq+1] first row of SOC c
};

thread_local std::string g_err;

// ------------------------------------------------------------------------------------------------
// workgroup reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// reduces K values over the workgroup; bit k of maxmask selects max instead of sum.  red: NW*K doubles of LDS.
template <int K>
__device__ __forceinline__ void block_reduce(double (&v)[K], unsigned maxmask, double *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = ((maxmask >> k) & 1u) ? wave_max(v[k]) : wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[wid * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double a = red[k];
#pragma unroll
        for (int w = 1; w < NW; w++) a = ((maxmask >> k) & 1u) ? fmax(a, red[w * K + k]) : a + red[w * K + k];
        v[k] = a;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// matvec building blocks on an LDS (or L2-resident) row-major matrix  Mat[rows][ld]
// P1:  part[ch][j] = sum_{i in chunk ch} Mat[i][j] * v[i]       (out indexed by COLUMN; lanes walk j -> conflict-free)
__device__ __forceinline__ int chunks_for(int outs) { int c = NT / outs; return c < 1 ? 1 : c; }

__device__ __forceinline__ void mv_cols_partial(const double *Mat, int ld, int rows, int cols, const double *v, double *part) {
    const int CH = chunks_for(cols);
    const int len = (rows + CH - 1) / CH;
    for (int idx = threadIdx.x; idx < cols * CH; idx += NT) {
        const int j = idx % cols, ch = idx / cols;
        const int i0 = ch * len, i1 = min(rows, i0 + len);
        double a0 = 0, a1 = 0;
        int i = i0;
        for (; i + 1 < i1; i += 2) {
            a0 = fma(Mat[i * ld + j], v[i], a0);
            a1 = fma(Mat[(i + 1) * ld + j], v[i + 1], a1);
        }
        if (i < i1) a0 = fma(Mat[i * ld + j], v[i], a0);
        part[ch * cols + j] = a0 + a1;
    }
}
__device__ __forceinline__ double sum_parts(const double *part, int outs, int idx) {
    const int CH = chunks_for(outs);
    double a = part[idx];
    for (int c = 1; c < CH; c++) a += part[c * outs + idx];
    return a;
}
// P2:  part[ch][i] = sum_{j in chunk ch} Mat[i][j] * v[j]       (out indexed by ROW; ld odd -> conflict-free ds_read_b64)
__device__ __forceinline__ void mv_rows_partial(const double *Mat, int ld, int rows, int cols, const double *v, double *part) {
    const int CH = chunks_for(rows);
    const int len = (cols + CH - 1) / CH;
    for (int idx = threadIdx.x; idx < rows * CH; idx += NT) {
        const int i = idx % rows, ch = idx / rows;
        const int j0 = ch * len, j1 = min(cols, j0 + len);
        const double *r = Mat + i * ld;
        double a0 = 0, a1 = 0;
        int j = j0;
        for (; j + 1 < j1; j += 2) {
            a0 = fma(r[j], v[j], a0);
            a1 = fma(r[j + 1], v[j + 1], a1);
        }
        if (j < j1) a0 = fma(r[j], v[j], a0);
        part[ch * rows + i] = a0 + a1;
    }
}

__device__ __forceinline__ double clamp_scale(double v) { return v < MIN_SCALE ? 1.0 : (v > MAX_SCALE ? MAX_SCALE : v); }

// scatter one instance's boundary values (batch-major row of [A_cvx | b_cvx] values) into dense solver form
__device__ __forceinline__ void load_instance(const DevT &T, const double *vals, double *A, double *bv) {
    const int n = T.n, m = T.m, lda = T.lda;
    for (int i = threadIdx.x; i < m * lda; i += NT) A[i] = 0.0;
    for (int i = threadIdx.x; i < m; i += NT) bv[i] = 0.0;
    __syncthreads();
    for (int k = threadIdx.x; k < T.nnz_aug; k += NT) {
        const double val = vals[k];
        const int r = T.rowidx[k], c = T.colidx[k];
        if (c < n) A[r * lda + c] = -val;      // solver sees A = -A_cvx (diffcp_if.py:65)
        else bv[r] = val;                      // b = b_cvx          (diffcp_if.py:66)
    }
    __syncthreads();
}

// ================================================================================================
// FORWARD
// ================================================================================================
template <bool A_LDS, bool G_LDS>
__global__ void __launch_bounds__(NT)
k_forward(DevT T, ce_settings S, const double *__restrict__ Avals, const double *__restrict__ qv, long sqk, long sqb,
          double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ so, int *__restrict__ iters_o,
          int *__restrict__ status_o, double *__restrict__ resid_o, double *gwsA, double *gwsG) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, inst = blockIdx.x;
    const int n = T.n, m = T.m, l = n + m + 1, lda = T.lda, ldg = T.ldg, nq = T.nq, z = T.z;
    const int PB = max(NT, max(n, m));      // partial-sum buffer length

    double *p = sm;
    double *A, *G;
    if constexpr (A_LDS) { A = p; p += m * lda; } else { A = gwsA + (size_t)inst * m * lda; }
    if constexpr (G_LDS) { G = p; p += n * ldg; } else { G = gwsG + (size_t)inst * n * ldg; }
    double *bv = p; p += m;      // b-hat
    double *cv = p; p += n;      // c-hat
    double *Dv = p; p += m;      // row equilibration
    double *Ev = p; p += n;      // column equilibration
    double *g = p; p += l;       // (R_z + M_zz)^{-1} h
    double *w = p; p += l;       // DR iterate
    double *ut = p; p += l;      // u-tilde
    double *u = p; p += l;       // cone iterate
    double *phi = p; p += l;     // functional giving the tau-tilde numerator
    double *tv = p; p += max(n, m); // scratch vector
    double *part = p; p += PB;
    double *part2 = p; p += PB;
    double *red = p; p += NW * 8;
    double *socc = p; p += 2 * (nq > 0 ? nq : 1);
    double *wpart = p; p += NW;  // per-wave partials of phi . w
    double *sc = p; p += 2 * n;  // refactor() right-hand sides (keeps u / ut intact across a rescale)

    // ---------------------------------------------------------------- load
    load_instance(T, Avals + (size_t)inst * T.nnz_aug, A, bv);
    for (int j = tid; j < n; j += NT) { cv[j] = qv[j * sqk + inst * sqb]; Ev[j] = 1.0; }
    for (int i = tid; i < m; i += NT) Dv[i] = 1.0;
    __syncthreads();
    double nrm_b0, nrm_c0;
    {
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT) r[0] = fmax(r[0], fabs(bv[i]));
        for (int j = tid; j < n; j += NT) r[1] = fmax(r[1], fabs(cv[j]));
        block_reduce<2>(r, 3u, red);
        nrm_b0 = r[0]; nrm_c0 = r[1];
    }
    // ---------------------------------------------------------------- equilibration (SCS normalize)
    double sigma = 1.0;
    if (S.normalize) {
        for (int pass = 0; pass < NUM_RUIZ_PASSES + NUM_L2_PASSES; pass++) {
            const bool l2 = pass >= NUM_RUIZ_PASSES;
            {   // row norms -> part (indexed by row), column norms -> part2 (indexed by column)
                const int CH = chunks_for(m), len = (n + CH - 1) / CH;
                for (int idx = tid; idx < m * CH; idx += NT) {
                    const int i = idx % m, ch = idx / m, j0 = ch * len, j1 = min(n, j0 + len);
                    const double *r = A + i * lda; double a = 0;
                    for (int j = j0; j < j1; j++) { const double v = r[j]; a = l2 ? fma(v, v, a) : fmax(a, fabs(v)); }
                    part[ch * m + i] = a;
                }
                const int CH2 = chunks_for(n), len2 = (m + CH2 - 1) / CH2;
                for (int idx = tid; idx < n * CH2; idx += NT) {
                    const int j = idx % n, ch = idx / n, i0 = ch * len2, i1 = min(m, i0 + len2);
                    double a = 0;
                    for (int i = i0; i < i1; i++) { const double v = A[i * lda + j]; a = l2 ? fma(v, v, a) : fmax(a, fabs(v)); }
                    part2[ch * n + j] = a;
                }
            }
            __syncthreads();
            for (int i = tid; i < m; i += NT) {
                const int CH = chunks_for(m); double a = part[i];
                for (int c = 1; c < CH; c++) a = l2 ? a + part[c * m + i] : fmax(a, part[c * m + i]);
                if (l2) a = sqrt(a);
                tv[i] = (T.rowcone[i] < 0) ? 1.0 / sqrt(clamp_scale(a)) : a;   // SOC rows: raw norm, averaged below
            }
            for (int j = tid; j < n; j += NT) {
                const int CH = chunks_for(n); double a = part2[j];
                for (int c = 1; c < CH; c++) a = l2 ? a + part2[c * n + j] : fmax(a, part2[c * n + j]);
                if (l2) a = sqrt(a);
                u[j] = 1.0 / sqrt(clamp_scale(a));                                // Et (u is free scratch here)
            }
            __syncthreads();
            if (nq > 0) {   // block-average the row scaling inside each SOC so the scaled cone is still the cone
                for (int c = tid; c < nq; c += NT) {
                    const int r0 = T.qoff[c], r1 = T.qoff[c + 1]; double a = 0;
                    for (int i = r0; i < r1; i++) a += tv[i];
                    if (r1 > r0) { a = 1.0 / sqrt(clamp_scale(a / (r1 - r0))); for (int i = r0; i < r1; i++) tv[i] = a; }
                }
                __syncthreads();
            }
            for (int idx = tid; idx < m * n; idx += NT) { const int i = idx / n, j = idx % n; A[i * lda + j] *= tv[i] * u[j]; }
            for (int i = tid; i < m; i += NT) Dv[i] *= tv[i];
            for (int j = tid; j < n; j += NT) Ev[j] *= u[j];
            __syncthreads();
        }
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT) { bv[i] *= Dv[i]; r[0] = fmax(r[0], fabs(bv[i])); }
        for (int j = tid; j < n; j += NT) { cv[j] *= Ev[j]; r[1] = fmax(r[1], fabs(cv[j])); }
        block_reduce<2>(r, 3u, red);
        sigma = 1.0 / clamp_scale(fmax(r[0], r[1]));
        for (int i = tid; i < m; i += NT) bv[i] *= sigma;
        for (int j = tid; j < n; j += NT) cv[j] *= sigma;
        __syncthreads();
    }

    double scale = S.scale;
    const double rho_x = S.rho_x, rtau = TAU_FACTOR, alpha = S.alpha;
    double hg = 0;
    auto dyv = [&](int i) -> double { return (i < z) ? ZERO_CONE_FACTOR * scale : scale; };   // 1 / r_y

    // ---- (re)factor: G = (rho_x I + A^T Dy A)^{-1}, g, h.g, phi    (uniform control flow; ends synchronised)
    auto refactor = [&]() {
        for (int idx = tid; idx < n * n; idx += NT) {
            const int a = idx / n, b2 = idx % n;
            double acc0 = 0, acc1 = 0; int i = 0;
            for (; i + 1 < m; i += 2) {
                acc0 = fma(A[i * lda + a] * dyv(i), A[i * lda + b2], acc0);
                acc1 = fma(A[(i + 1) * lda + a] * dyv(i + 1), A[(i + 1) * lda + b2], acc1);
            }
            if (i < m) acc0 = fma(A[i * lda + a] * dyv(i), A[i * lda + b2], acc0);
            G[a * ldg + b2] = acc0 + acc1 + (a == b2 ? rho_x : 0.0);
        }
        __syncthreads();
        // in-place Gauss-Jordan inversion (SPD: no pivoting needed)
        double *colk = part, *rowk = part2;
        for (int k = 0; k < n; k++) {
            for (int i = tid; i < n; i += NT) { colk[i] = G[i * ldg + k]; rowk[i] = G[k * ldg + i]; }
            __syncthreads();
            const double pinv = 1.0 / rowk[k];
            for (int idx = tid; idx < n * n; idx += NT) {
                const int i = idx / n, j = idx % n;
                double v;
                if (i == k) v = (j == k) ? pinv : rowk[j] * pinv;
                else if (j == k) v = -colk[i] * pinv;
                else v = fma(-colk[i] * pinv, rowk[j], G[i * ldg + j]);
                G[i * ldg + j] = v;
            }
            __syncthreads();
        }
        // tv = Dy*b  (m) ;  part = A^T tv partials
        for (int i = tid; i < m; i += NT) tv[i] = dyv(i) * bv[i];
        __syncthreads();
        mv_cols_partial(A, lda, m, n, tv, part);
        __syncthreads();
        // sc[0:n] = c - A^T Dy b   (rhs for g_x) ;  sc[n:2n] = c + A^T Dy b  (k, for phi)
        for (int j = tid; j < n; j += NT) { const double a = sum_parts(part, n, j); sc[j] = cv[j] - a; sc[n + j] = cv[j] + a; }
        __syncthreads();
        mv_cols_partial(G, ldg, n, n, sc, part);      // G symmetric: column form == row form
        mv_cols_partial(G, ldg, n, n, sc + n, part2);
        __syncthreads();
        for (int j = tid; j < n; j += NT) { g[j] = sum_parts(part, n, j); tv[j] = sum_parts(part2, n, j); }   // tv[0:n] = G k
        __syncthreads();
        mv_rows_partial(A, lda, m, n, g, part);      // A g_x
        mv_rows_partial(A, lda, m, n, tv, part2);    // A G k
        __syncthreads();
        double r[1] = {0};
        for (int i = tid; i < m; i += NT) {
            const double gy = dyv(i) * (sum_parts(part, m, i) + bv[i]);
            g[n + i] = gy; r[0] += bv[i] * gy;
            phi[n + i] = bv[i] - sum_parts(part2, m, i);
        }
        for (int j = tid; j < n; j += NT) { r[0] += cv[j] * g[j]; phi[j] = rho_x * tv[j]; }
        block_reduce<1>(r, 0u, red);
        hg = r[0];
    };
    // phi . w  (z part), as per-wave partials consumed one iteration later
    auto phiw_partials = [&]() {
        double a = 0;
        for (int e = tid; e < l - 1; e += NT) a += phi[e] * w[e];
        a = wave_sum(a);
        if ((tid & 63) == 0) wpart[tid >> 6] = a;
    };

    refactor();
    for (int e = tid; e < l; e += NT) w[e] = (e == l - 1) ? 1.0 : 0.0;    // cold start
    if (tid < NW) wpart[tid] = 0.0;
    __syncthreads();

    int status = 0, iter = 0, last_scale_iter = 0, n_log = 0;
    double sum_log = 0, res_pri = NAN, res_dual = NAN, gap = NAN;
    double tau = 0, kap = 0, ctx = 0, bty = 0;

    for (iter = 0; iter < S.max_iters; iter++) {
        const bool check = (iter % CONVERGED_INTERVAL) == 0;
        if (check && iter > 0) {   // keep the homogeneous iterate in range
            double r[1] = {0};
            for (int e = tid; e < l; e += NT) r[0] += w[e] * w[e];
            block_reduce<1>(r, 0u, red);
            const double nw = sqrt(r[0]);
            if (nw > 0) { const double f = sqrt((double)l) / nw; for (int e = tid; e < l; e += NT) w[e] *= f; }
            __syncthreads();
            phiw_partials();
            __syncthreads();
        }
        // S1: A^T w_y
        mv_cols_partial(A, lda, m, n, w + n, part);
        __syncthreads();
        // S2: t = rho_x w_x - A^T w_y
        for (int j = tid; j < n; j += NT) tv[j] = rho_x * w[j] - sum_parts(part, n, j);
        __syncthreads();
        // S3: G t
        mv_cols_partial(G, ldg, n, n, tv, part);
        __syncthreads();
        // S4: p_x
        for (int j = tid; j < n; j += NT) ut[j] = sum_parts(part, n, j);
        __syncthreads();
        // S5: A p_x
        mv_rows_partial(A, lda, m, n, ut, part);
        __syncthreads();
        // S6/S7: tau-tilde, u-tilde, cone input
        double numer = rtau * w[l - 1];
#pragma unroll
        for (int k = 0; k < NW; k++) numer += wpart[k];
        const double tau_t = numer / (rtau + hg);
        for (int e = tid; e < l; e += NT) {
            double ute, ue;
            if (e < n) { ute = ut[e] - tau_t * g[e]; ue = 2 * ute - w[e]; }
            else if (e < l - 1) {
                const int i = e - n;
                const double py = w[e] + dyv(i) * sum_parts(part, m, i);
                ute = py - tau_t * g[e]; ue = 2 * ute - w[e];
                if (i >= z && T.rowcone[i] < 0 && ue < 0) ue = 0;      // nonneg rows; zero-cone dual is free
            } else { ute = tau_t; ue = fmax(0.0, 2 * tau_t - w[e]); }
            ut[e] = ute; u[e] = ue;
        }
        __syncthreads();
        // S8: SOC projection coefficients  u_c = (c0, f * zbar)
        if (nq > 0) {
            for (int c = tid; c < nq; c += NT) {
                const int r0 = n + T.qoff[c], r1 = n + T.qoff[c + 1];
                if (r1 - r0 == 1) { socc[2 * c] = fmax(u[r0], 0.0); socc[2 * c + 1] = 0.0; continue; }
                const double t0 = u[r0]; double nz = 0;
                for (int e = r0 + 1; e < r1; e++) nz = fma(u[e], u[e], nz);
                nz = sqrt(nz);
                double c0, f;
                if (nz <= t0) { c0 = t0; f = 1.0; }
                else if (nz <= -t0) { c0 = 0.0; f = 0.0; }
                else { c0 = 0.5 * (t0 + nz); f = c0 / nz; }
                socc[2 * c] = c0; socc[2 * c + 1] = f;
            }
            __syncthreads();
            for (int i = tid + (z + T.l); i < m; i += NT) {
                const int c = T.rowcone[i];
                u[n + i] = (i == T.qoff[c]) ? socc[2 * c] : socc[2 * c + 1] * u[n + i];
            }
            __syncthreads();
        }
        // ---- termination test / adaptive scale (uniform branch)
        bool stop = false;
        if (check) {
            mv_rows_partial(A, lda, m, n, u, part);          // A-hat x-hat
            mv_cols_partial(A, lda, m, n, u + n, part2);     // A-hat^T y-hat
            __syncthreads();
            tau = fabs(u[l - 1]);
            kap = fabs(rtau * (u[l - 1] + w[l - 1] - 2 * ut[l - 1]));
            const double isg = 1.0 / sigma;
            double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rp, nax, ns, naxs, rd, naty (max) ; ctx, bty (sum)
            for (int i = tid; i < m; i += NT) {
                const double sc = isg / Dv[i];
                const double ax = sum_parts(part, m, i) * sc;
                const double sh = (u[n + i] + w[n + i] - 2 * ut[n + i]) / dyv(i) * sc;
                const double bt = bv[i] * tau * sc;
                r[0] = fmax(r[0], fabs(ax + sh - bt)); r[1] = fmax(r[1], fabs(ax)); r[2] = fmax(r[2], fabs(sh));
                r[3] = fmax(r[3], fabs(ax + sh));
                r[7] += bv[i] * u[n + i] * isg * isg;
            }
            for (int j = tid; j < n; j += NT) {
                const double sc = isg / Ev[j];
                const double aty = sum_parts(part2, n, j) * sc;
                r[4] = fmax(r[4], fabs(aty + cv[j] * tau * sc)); r[5] = fmax(r[5], fabs(aty));
                r[6] += cv[j] * u[j] * isg * isg;
            }
            block_reduce<8>(r, 0x3Fu, red);
            const double rp = r[0], nax = r[1], ns = r[2], naxs = r[3], rd = r[4], naty = r[5];
            ctx = r[6]; bty = r[7];
            if (tau > 0) {
                res_pri = rp / tau; res_dual = rd / tau; gap = fabs(ctx + bty) / tau;
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
                    sum_log += log(rel_p) - log(rel_d); n_log++;
                    const double factor = sqrt(exp(sum_log / n_log));
                    if (iter - last_scale_iter >= RESCALING_MIN_ITERS) {
                        const double ns2 = fmin(fmax(scale * factor, MIN_SCALE_VALUE), MAX_SCALE_VALUE);
                        if (ns2 != scale && (factor > sqrt(10.0) || factor < 1.0 / sqrt(10.0))) {
                            // keep (s, kappa):  R+ (w+ + u - 2 ut) = rsk  ->  w_y+ = rsk_y / r_y+ + 2 ut_y - u_y
                            const double dy_ratio = ns2 / scale;       // Dy+ / Dy, same for zero and cone rows
                            for (int i = tid; i < m; i += NT) {
                                const double d0 = u[n + i] + w[n + i] - 2 * ut[n + i];   // = rsk_y * Dy
                                w[n + i] = d0 * dy_ratio + 2 * ut[n + i] - u[n + i];
                            }
                            sum_log = 0; n_log = 0; last_scale_iter = iter; scale = ns2;
                            __syncthreads();
                            refactor();
                            phiw_partials();
                            __syncthreads();
                        }
                    }
                }
            }
        }
        if (stop) break;
        if (iter + 1 >= S.max_iters) { iter++; break; }   // keep w pre-update so (s, kappa) match the last cone step
        // S9: relaxed update of w, and phi.w for the next iteration
        {
            double a = 0;
            for (int e = tid; e < l; e += NT) {
                const double we = w[e] + alpha * (u[e] - ut[e]);
                w[e] = we;
                if (e < l - 1) a += phi[e] * we;
            }
            a = wave_sum(a);
            if ((tid & 63) == 0) wpart[tid >> 6] = a;
        }
        __syncthreads();
    }

    if (status == 0) {   // ran out of iterations (SCS set_unfinished)
        tau = fabs(u[l - 1]);
        kap = fabs(rtau * (u[l - 1] + w[l - 1] - 2 * ut[l - 1]));
        double r[2] = {0, 0};
        const double isg = 1.0 / sigma;
        for (int j = tid; j < n; j += NT) r[0] += cv[j] * u[j] * isg * isg;
        for (int i = tid; i < m; i += NT) r[1] += bv[i] * u[n + i] * isg * isg;
        block_reduce<2>(r, 0u, red);
        if (tau > kap) status = 2; else if (r[1] < r[0]) status = -7; else status = -6;
    }
    // ---------------------------------------------------------------- write back (un-normalise)
    {
        const bool solved = (status == 1 || status == 2);
        const bool infeas = (status == -2 || status == -7);
        const double it = solved ? 1.0 / (sigma * tau) : 1.0 / sigma;
        for (int j = tid; j < n; j += NT) xo[(size_t)inst * n + j] = infeas ? NAN : Ev[j] * u[j] * it;
        for (int i = tid; i < m; i += NT) {
            const double sh = (u[n + i] + w[n + i] - 2 * ut[n + i]) / dyv(i);
            yo[(size_t)inst * m + i] = (solved || infeas) ? Dv[i] * u[n + i] * it : NAN;
            so[(size_t)inst * m + i] = infeas ? NAN : sh / Dv[i] * it;
        }
        if (tid == 0) {
            iters_o[inst] = iter; status_o[inst] = status;
            if (resid_o) { resid_o[3 * inst] = res_pri; resid_o[3 * inst + 1] = res_dual; resid_o[3 * inst + 2] = gap; }
        }
    }
}

// ================================================================================================
// BACKWARD
// ================================================================================================
// row kinds after classifying DPi_{K*}(v), v = y - s
enum { RK_EQ = 0, RK_FREE = 1, RK_SOCB = 2 };

template <bool A_LDS, bool K_LDS>
__global__ void __launch_bounds__(NT)
k_backward(DevT T, int nkcap, int ldk, const double *__restrict__ Avals, const double *__restrict__ xg,
           const double *__restrict__ yg, const double *__restrict__ sg, const double *__restrict__ dxg,
           const double *__restrict__ dyg, double *__restrict__ dAo, double *__restrict__ dqo, long sdqk, long sdqb,
           int *__restrict__ adj_status, double *gwsA, double *gwsK) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, inst = blockIdx.x;
    const int n = T.n, m = T.m, lda = T.lda, nq = T.nq, z = T.z;
    const int PB = max(NT, max(n, m));
    const int nqs = nq > 0 ? nq : 1;

    double *p = sm;
    double *A, *K;
    if constexpr (A_LDS) { A = p; p += m * lda; } else { A = gwsA + (size_t)inst * m * lda; }
    if constexpr (K_LDS) { K = p; p += nkcap * ldk; } else { K = gwsK + (size_t)inst * nkcap * ldk; }
    double *bv = p; p += m;          // (unused values; load_instance fills b)
    double *xv = p; p += n;
    double *yv = p; p += m;
    double *vv = p; p += m;          // v = y - s ; later r_y
    double *dv = p; p += m;          // d = DPi dy
    double *qv2 = p; p += m;         // A r_x
    double *rx = p; p += n;
    double *ay = p; p += nqs * n;    // A_c^T e_y
    double *as = p; p += nqs * n;    // A_c^T e_s
    double *cinfo = p; p += 6 * nqs; // per cone: lambda, nz, e_y.d, e_s.d, (spare)
    double *part = p; p += PB;
    double *red = p; p += NW * 8;
    int *ip = (int *)p;
    int *rkind = ip; ip += m;        // row kind
    int *eqrow = ip; ip += m;        // equality index of row (RK_EQ) or -1
    int *ckind = ip; ip += nqs;      // cone kind: 0 = interior of K* (all EQ), 1 = in -K (all FREE), 2 = boundary
    int *ceq = ip; ip += nqs;        // equality index of the e_y row of a boundary cone
    int *perm = ip; ip += nkcap;
    int *misc = ip; ip += 4;         // [0] n_eq, [1] pivot row, [2] flags

    load_instance(T, Avals + (size_t)inst * T.nnz_aug, A, bv);
    for (int j = tid; j < n; j += NT) xv[j] = xg[(size_t)inst * n + j];
    for (int i = tid; i < m; i += NT) {
        const double yi = yg[(size_t)inst * m + i];
        yv[i] = yi; vv[i] = yi - sg[(size_t)inst * m + i];
    }
    __syncthreads();
    // ---- classify
    for (int i = tid; i < z + T.l; i += NT) rkind[i] = (i < z || vv[i] > 0) ? RK_EQ : RK_FREE;
    for (int c = tid; c < nq; c += NT) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1], d = r1 - r0;
        int kind; double lam = 0, nz = 0;
        if (d == 1) kind = vv[r0] >= 0 ? 0 : 1;
        else {
            for (int i = r0 + 1; i < r1; i++) nz = fma(vv[i], vv[i], nz);
            nz = sqrt(nz);
            const double t0 = vv[r0];
            if (nz <= t0) kind = 0; else if (nz <= -t0) kind = 1; else { kind = 2; lam = (t0 + nz) / (2 * nz); }
        }
        ckind[c] = kind; cinfo[6 * c] = lam; cinfo[6 * c + 1] = nz;
        for (int i = r0; i < r1; i++) rkind[i] = kind == 0 ? RK_EQ : (kind == 1 ? RK_FREE : RK_SOCB);
    }
    __syncthreads();
    if (tid == 0) {   // equality numbering (serial scan; m is small)
        int ne = 0;
        for (int i = 0; i < m; i++) eqrow[i] = (rkind[i] == RK_EQ) ? ne++ : -1;
        for (int c = 0; c < nq; c++) ceq[c] = (ckind[c] == 2) ? ne++ : -1;
        misc[0] = ne; misc[2] = 0;
    }
    __syncthreads();
    const int neq = misc[0];
    const int NK = n + neq;
    if (NK > nkcap) {   // more active rows than the direct solve holds: degenerate instance (flagged, zero gradient)
        for (int k = tid; k < T.nnz_aug; k += NT) dAo[(size_t)inst * T.nnz_aug + k] = 0.0;
        for (int j = tid; j <= n; j += NT) dqo[j * sdqk + inst * sdqb] = 0.0;
        if (tid == 0 && adj_status) adj_status[inst] = 2;
        return;
    }
    // ---- d = DPi(v) dy   (symmetric), per-cone scalars e_y.d, e_s.d
    for (int i = tid; i < z + T.l; i += NT) dv[i] = rkind[i] == RK_EQ ? dyg[(size_t)inst * m + i] : 0.0;
    for (int c = tid; c < nq; c += NT) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        const double *h = dyg + (size_t)inst * m;
        if (ckind[c] == 0) { for (int i = r0; i < r1; i++) dv[i] = h[i]; }
        else if (ckind[c] == 1) { for (int i = r0; i < r1; i++) dv[i] = 0.0; }
        else {
            const double t0 = vv[r0], nz = cinfo[6 * c + 1];
            double zh = 0; for (int i = r0 + 1; i < r1; i++) zh = fma(vv[i], h[i], zh);
            dv[r0] = (nz * h[r0] + zh) / (2 * nz);
            for (int i = r0 + 1; i < r1; i++) dv[i] = (vv[i] * h[r0] + (t0 + nz) * h[i] - t0 * vv[i] * zh / (nz * nz)) / (2 * nz);
            // e_y = (1, zhat)/sqrt2, e_s = (1, -zhat)/sqrt2
            double zd = 0; for (int i = r0 + 1; i < r1; i++) zd = fma(vv[i], dv[i], zd);
            zd /= nz;
            cinfo[6 * c + 2] = (dv[r0] + zd) * M_SQRT1_2;   // e_y . d
            cinfo[6 * c + 3] = (dv[r0] - zd) * M_SQRT1_2;   // e_s . d
        }
    }
    __syncthreads();
    // ---- a_y, a_s for boundary cones
    for (int idx = tid; idx < nq * n; idx += NT) {
        const int c = idx / n, j = idx % n;
        if (ckind[c] != 2) continue;
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        const double inz = 1.0 / cinfo[6 * c + 1];
        double a = 0; for (int i = r0 + 1; i < r1; i++) a = fma(A[i * lda + j], vv[i], a);
        a *= inz;
        ay[c * n + j] = (A[r0 * lda + j] + a) * M_SQRT1_2;
        as[c * n + j] = (A[r0 * lda + j] - a) * M_SQRT1_2;
    }
    __syncthreads();
    // ---- assemble K = [[H, -B^T],[B, 0]] | rhs
    for (int idx = tid; idx < NK * (NK + 1); idx += NT) {
        const int r = idx / (NK + 1), cidx = idx % (NK + 1);
        double val = 0;
        if (r < n && cidx < n) {            // H[a][b] = sum_c theta_c (A_c^T A_c - a_y a_y^T - a_s a_s^T)
            for (int c = 0; c < nq; c++) {
                if (ckind[c] != 2) continue;
                const double lam = cinfo[6 * c], th = lam / (1 - lam);
                const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
                double a = 0; for (int i = r0; i < r1; i++) a = fma(A[i * lda + r], A[i * lda + cidx], a);
                a -= ay[c * n + r] * ay[c * n + cidx] + as[c * n + r] * as[c * n + cidx];
                val = fma(th, a, val);
            }
        } else if (r < n && cidx == NK) {   // f = dx + sum_FREE a_i d_i + sum_B [ a_s (e_s.d) + A_c^T P d / (1-lam) ]
            val = dxg[(size_t)inst * n + r];
            for (int i = 0; i < z + T.l; i++) if (rkind[i] == RK_FREE) val = fma(A[i * lda + r], dv[i], val);
            for (int c = 0; c < nq; c++) {
                const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
                if (ckind[c] == 1) { for (int i = r0; i < r1; i++) val = fma(A[i * lda + r], dv[i], val); }
                else if (ckind[c] == 2) {
                    const double lam = cinfo[6 * c], eyd = cinfo[6 * c + 2], esd = cinfo[6 * c + 3];
                    double a = 0; for (int i = r0; i < r1; i++) a = fma(A[i * lda + r], dv[i], a);
                    a -= ay[c * n + r] * eyd + as[c * n + r] * esd;      // A_c^T P d
                    val += as[c * n + r] * esd + a / (1 - lam);
                }
            }
        } else if (r >= n && cidx == NK) {  // d_B  (filled below by the owning row / cone)
            val = 0;
        } else val = 0;
        K[r * ldk + cidx] = val;
    }
    __syncthreads();
    for (int idx = tid; idx < m * n; idx += NT) {   // B rows of plain equality rows
        const int i = idx / n, j = idx % n; const int e = eqrow[i];
        if (e >= 0) { const double a = A[i * lda + j]; K[(n + e) * ldk + j] = a; K[j * ldk + (n + e)] = -a; }
    }
    for (int idx = tid; idx < nq * n; idx += NT) {  // B rows of boundary cones (e_y rows)
        const int c = idx / n, j = idx % n; const int e = ceq[c];
        if (e >= 0) { const double a = ay[c * n + j]; K[(n + e) * ldk + j] = a; K[j * ldk + (n + e)] = -a; }
    }
    for (int i = tid; i < m; i += NT) if (eqrow[i] >= 0) K[(n + eqrow[i]) * ldk + NK] = dv[i];
    for (int c = tid; c < nq; c += NT) if (ceq[c] >= 0) K[(n + ceq[c]) * ldk + NK] = cinfo[6 * c + 2];
    for (int i = tid; i < NK; i += NT) perm[i] = i;
    __syncthreads();
    // ---- Gauss-Jordan with partial pivoting on [K | rhs]
    double kmax;
    {
        double r[1] = {0};
        for (int idx = tid; idx < NK * NK; idx += NT) r[0] = fmax(r[0], fabs(K[(idx / NK) * ldk + idx % NK]));
        block_reduce<1>(r, 1u, red);
        kmax = r[0];
    }
    const double ptol = 1e-13 * (kmax > 0 ? kmax : 1.0);
    for (int k = 0; k < NK; k++) {
        if (tid < 64) {   // pivot search by wave 0 over logical rows k..NK-1
            double best = -1; int bi = k;
            for (int i = k + tid; i < NK; i += 64) { const double v = fabs(K[perm[i] * ldk + k]); if (v > best) { best = v; bi = i; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) {
                const int t = perm[k]; perm[k] = perm[bi]; perm[bi] = t;
                if (best < ptol) { misc[2] = 1; }
            }
        }
        __syncthreads();
        const int pk = perm[k];
        double piv = K[pk * ldk + k];
        if (fabs(piv) < ptol) piv = (piv < 0 ? -ptol : ptol);
        const double pinv = 1.0 / piv;
        const int wcols = NK - k;   // columns k+1 .. NK
        for (int idx = tid; idx < NK * wcols; idx += NT) {
            const int i = idx / wcols, j = k + 1 + idx % wcols;
            if (i == k) continue;
            const int pi = perm[i];
            const double f = K[pi * ldk + k] * pinv;
            if (f != 0.0) K[pi * ldk + j] = fma(-f, K[pk * ldk + j], K[pi * ldk + j]);
        }
        __syncthreads();
    }
    // solution: sol_k = rhs[perm[k]] / K[perm[k]][k];  r_x -> rx, multipliers rho -> bv (b is not needed by the adjoint)
    for (int k = tid; k < NK; k += NT) {
        const int pk = perm[k];
        double piv = K[pk * ldk + k];
        if (fabs(piv) < ptol) piv = (piv < 0 ? -ptol : ptol);
        const double sol = K[pk * ldk + NK] / piv;
        if (k < n) rx[k] = sol; else bv[k - n] = sol;
    }
    __syncthreads();
    // ---- q = A r_x ; r_y
    mv_rows_partial(A, lda, m, n, rx, part);
    __syncthreads();
    for (int i = tid; i < m; i += NT) qv2[i] = sum_parts(part, m, i);
    __syncthreads();
    for (int i = tid; i < z + T.l; i += NT) vv[i] = (eqrow[i] >= 0) ? bv[eqrow[i]] : dv[i];
    for (int c = tid; c < nq; c += NT) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        if (ckind[c] == 0) { for (int i = r0; i < r1; i++) vv[i] = bv[eqrow[i]]; }
        else if (ckind[c] == 1) { for (int i = r0; i < r1; i++) vv[i] = dv[i]; }
        else {
            // r_y = rho_y e_y + (e_s.d) e_s + (P d - lam P q) / (1 - lam),   P = I - e_y e_y^T - e_s e_s^T
            const double lam = cinfo[6 * c], inz = 1.0 / cinfo[6 * c + 1], eyd = cinfo[6 * c + 2], esd = cinfo[6 * c + 3];
            const double rhoy = bv[ceq[c]];
            double zq = 0; for (int i = r0 + 1; i < r1; i++) zq = fma(vv[i], qv2[i], zq);
            zq *= inz;
            const double eyq = (qv2[r0] + zq) * M_SQRT1_2, esq = (qv2[r0] - zq) * M_SQRT1_2;
            const double il = 1.0 / (1 - lam);
            // coefficients on e_y and e_s after expanding P
            const double cy = rhoy - il * (eyd - lam * eyq), cs = esd - il * (esd - lam * esq);
            // component form: e_y = (1, zhat)/sqrt2 ; e_s = (1, -zhat)/sqrt2
            const double k0 = (cy + cs) * M_SQRT1_2, kz = (cy - cs) * M_SQRT1_2;
            // careful: vv[] (zbar) is overwritten in place -> do row 0 last, scale zhat on the fly
            for (int i = r0 + 1; i < r1; i++) { const double zh = vv[i] * inz; vv[i] = il * (dv[i] - lam * qv2[i]) + kz * zh; }
            vv[r0] = il * (dv[r0] - lam * qv2[r0]) + k0;
        }
    }
    __syncthreads();
    // ---- outputs in the boundary convention: dA_eval = [-dA.data, db[b_idx]], dq_eval = [dc, 0]
    //      dA_ij = x_j r_y,i - y_i r_x,j ; db = -r_y ; dc = -r_x     (r_tau pinned to 0)
    for (int k = tid; k < T.nnz_aug; k += NT) {
        const int i = T.rowidx[k], j = T.colidx[k];
        const double val = (j < n) ? -(xv[j] * vv[i] - yv[i] * rx[j]) : -vv[i];
        dAo[(size_t)inst * T.nnz_aug + k] = val;
    }
    for (int j = tid; j <= n; j += NT) dqo[j * sdqk + inst * sdqb] = (j < n) ? -rx[j] : 0.0;
    if (tid == 0 && adj_status) adj_status[inst] = misc[2];
}

// ================================================================================================
// layout kernels: (R x C) row-major <-> (C x R) row-major, fp64, 32x32 LDS tiles (+1 pad)
// ================================================================================================
__global__ void __launch_bounds__(256) k_transpose(const double *__restrict__ in, double *__restrict__ out, int R, int C) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) { const int rr = by + r, cc = bx + tx; if (rr < R && cc < C) tile[r][tx] = in[(size_t)rr * C + cc]; }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int cc = bx + r, rr = by + tx; if (rr < R && cc < C) out[(size_t)cc * R + rr] = tile[tx][r]; }
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
struct ce_engine {
    int device = 0;
    DevT T{};
    std::vector<int> q, s;
    int *d_rowidx = nullptr, *d_colidx = nullptr, *d_rowcone = nullptr, *d_qoff = nullptr;
    // workspace
    double *wsA = nullptr; size_t wsA_bytes = 0;          // batch-major copy of A_vals  [B][nnz_aug]
    double *wsdA = nullptr; size_t wsdA_bytes = 0;        // batch-major dA              [B][nnz_aug]
    double *gws = nullptr; size_t gws_bytes = 0;          // global residency fallback
    const double *retained_A = nullptr; int retained_B = 0;
    // launch plan
    int fwd_mode = 0, bwd_mode = 0; size_t fwd_lds = 0, bwd_lds = 0; int nkcap = 0, ldk = 0;
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[3];
};

#define HIPCHK(call)                                                                 \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return CE_E_HIP;                                                         \
        }                                                                            \
    } while (0)

static constexpr size_t LDS_LIMIT = 160 * 1024;

static size_t fwd_lds_bytes(const DevT &T, bool a_lds, bool g_lds) {
    const int n = T.n, m = T.m, l = n + m + 1, PB = std::max(NT, std::max(n, m));
    size_t d = 0;
    if (a_lds) d += (size_t)m * T.lda;
    if (g_lds) d += (size_t)n * T.ldg;
    d += 2 * (size_t)m + 2 * (size_t)n + 5 * (size_t)l + std::max(n, m) + 2 * (size_t)PB + NW * 8 + 2 * std::max(T.nq, 1) + NW + 2 * (size_t)n;
    return d * 8 + 16;
}
static size_t bwd_lds_bytes(const DevT &T, bool a_lds, bool k_lds, int nkcap, int ldk) {
    const int n = T.n, m = T.m, PB = std::max(NT, std::max(n, m)), nqs = std::max(T.nq, 1);
    size_t d = 0;
    if (a_lds) d += (size_t)m * T.lda;
    if (k_lds) d += (size_t)nkcap * ldk;
    d += 5 * (size_t)m + 2 * (size_t)n + 2 * (size_t)nqs * n + 6 * nqs + PB + NW * 8;
    size_t ints = 2 * (size_t)m + 2 * nqs + nkcap + 4;
    return d * 8 + ints * 4 + 16;
}

extern "C" {

const char *ce_last_error(void) { return g_err.c_str(); }

void ce_default_settings(ce_settings *s) {
    s->eps_abs = 1e-4; s->eps_rel = 1e-4; s->eps_infeas = 1e-7; s->alpha = 1.5; s->rho_x = 1e-6; s->scale = 0.1;
    s->max_iters = 100000; s->normalize = 1; s->adaptive_scale = 1; s->reserved = 0;
}

int ce_create(const ce_template *tpl, int device, ce_handle *out) {
    if (!tpl || !out || tpl->n <= 0 || tpl->m <= 0 || !tpl->indices || !tpl->indptr) { g_err = "bad template"; return CE_E_BADARG; }
    if (tpl->ns > 0 || tpl->nep > 0 || tpl->np > 0) { g_err = "PSD / exponential / power cones are not implemented on the device path yet"; return CE_E_UNSUPPORTED; }
    int rows = tpl->z + tpl->l;
    for (int i = 0; i < tpl->nq; i++) { if (tpl->q[i] < 1) { g_err = "bad SOC dim"; return CE_E_BADARG; } rows += tpl->q[i]; }
    if (rows != tpl->m) { g_err = "cone dims do not add up to m"; return CE_E_BADARG; }
    if (tpl->indptr[tpl->n + 1] != tpl->nnz_aug) { g_err = "indptr[n+1] != nnz_aug"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(device));
    ce_engine *h = new ce_engine();
    h->device = device;
    DevT &T = h->T;
    T.n = tpl->n; T.m = tpl->m; T.nnz_aug = tpl->nnz_aug; T.nnzA = tpl->indptr[tpl->n]; T.z = tpl->z; T.l = tpl->l; T.nq = tpl->nq;
    T.lda = tpl->n | 1; T.ldg = tpl->n | 1;     // odd leading dimension: conflict-free ds_read_b64 down a column of rows
    std::vector<int> colidx(tpl->nnz_aug), rowcone(tpl->m, -1), qoff(tpl->nq + 1, 0);
    for (int j = 0; j <= tpl->n; j++)
        for (int k = tpl->indptr[j]; k < tpl->indptr[j + 1]; k++) {
            if (tpl->indices[k] < 0 || tpl->indices[k] >= tpl->m) { delete h; g_err = "row index out of range"; return CE_E_BADARG; }
            colidx[k] = j;
        }
    int r = tpl->z + tpl->l;
    for (int c = 0; c < tpl->nq; c++) { qoff[c] = r; for (int i = 0; i < tpl->q[c]; i++) rowcone[r + i] = c; r += tpl->q[c]; }
    qoff[tpl->nq] = r;
    h->q.assign(tpl->q, tpl->q + tpl->nq);
    HIPCHK(hipMalloc(&h->d_rowidx, sizeof(int) * tpl->nnz_aug));
    HIPCHK(hipMalloc(&h->d_colidx, sizeof(int) * tpl->nnz_aug));
    HIPCHK(hipMalloc(&h->d_rowcone, sizeof(int) * tpl->m));
    HIPCHK(hipMalloc(&h->d_qoff, sizeof(int) * (tpl->nq + 1)));
    HIPCHK(hipMemcpy(h->d_rowidx, tpl->indices, sizeof(int) * tpl->nnz_aug, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_colidx, colidx.data(), sizeof(int) * tpl->nnz_aug, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_rowcone, rowcone.data(), sizeof(int) * tpl->m, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_qoff, qoff.data(), sizeof(int) * (tpl->nq + 1), hipMemcpyHostToDevice));
    T.rowidx = h->d_rowidx; T.colidx = h->d_colidx; T.rowcone = h->d_rowcone; T.qoff = h->d_qoff;
    // residency plan: mode 0 = everything in LDS, 1 = A in LDS / big matrix in global, 2 = both in global
    if (fwd_lds_bytes(T, true, true) <= LDS_LIMIT) h->fwd_mode = 0;
    else if (fwd_lds_bytes(T, true, false) <= LDS_LIMIT) h->fwd_mode = 1;
    else if (fwd_lds_bytes(T, false, false) <= LDS_LIMIT) h->fwd_mode = 2;
    else { ce_destroy(h); g_err = "instance vectors do not fit LDS"; return CE_E_TOO_LARGE; }
    h->fwd_lds = fwd_lds_bytes(T, h->fwd_mode <= 1, h->fwd_mode == 0);
    h->nkcap = T.n + std::min(T.m, T.n);
    h->ldk = (h->nkcap + 1) | 1;
    if (bwd_lds_bytes(T, true, true, h->nkcap, h->ldk) <= LDS_LIMIT) h->bwd_mode = 0;
    else if (bwd_lds_bytes(T, true, false, h->nkcap, h->ldk) <= LDS_LIMIT) h->bwd_mode = 1;
    else if (bwd_lds_bytes(T, false, false, h->nkcap, h->ldk) <= LDS_LIMIT) h->bwd_mode = 2;
    else { ce_destroy(h); g_err = "instance vectors do not fit LDS"; return CE_E_TOO_LARGE; }
    h->bwd_lds = bwd_lds_bytes(T, h->bwd_mode <= 1, h->bwd_mode == 0, h->nkcap, h->ldk);
#define SETATTR(kern, bytes) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
    SETATTR((k_forward<true, true>), LDS_LIMIT);  SETATTR((k_forward<true, false>), LDS_LIMIT);  SETATTR((k_forward<false, false>), LDS_LIMIT);
    SETATTR((k_backward<true, true>), LDS_LIMIT); SETATTR((k_backward<true, false>), LDS_LIMIT); SETATTR((k_backward<false, false>), LDS_LIMIT);
#undef SETATTR
    *out = h;
    return CE_OK;
}

int ce_destroy(ce_handle h) {
    if (!h) return CE_OK;
    hipSetDevice(h->device);
    hipFree(h->d_rowidx); hipFree(h->d_colidx); hipFree(h->d_rowcone); hipFree(h->d_qoff);
    hipFree(h->wsA); hipFree(h->wsdA); hipFree(h->gws);
    for (auto &v : h->ev) for (auto &p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    delete h;
    return CE_OK;
}

static int ensure(double **ptr, size_t *have, size_t need) {
    if (*have >= need) return CE_OK;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr; *have = 0;
    HIPCHK(hipMalloc(ptr, need));
    *have = need;
    return CE_OK;
}

struct ProfScope {
    ce_engine *h; int which; hipStream_t st; hipEvent_t a{}, b{}; bool on;
    ProfScope(ce_engine *h_, int w, hipStream_t s) : h(h_), which(w), st(s), on(h_->prof) {
        if (on) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, st); }
    }
    ~ProfScope() { if (on) { hipEventRecord(b, st); h->ev[which].push_back({a, b}); } }
};

// batch-minor (K x B) -> batch-major (B x K) when needed; returns the batch-major pointer
static int to_batch_major(ce_engine *h, int B, const double *vals, long sk, long sb, hipStream_t st, const double **out) {
    const int K = h->T.nnz_aug;
    if (sk == 1 && sb == K) { *out = vals; return CE_OK; }
    if (!(sb == 1 && sk == B)) { g_err = "A_vals must be contiguous batch-minor (sk=B,sb=1) or batch-major (sk=1,sb=nnz_aug)"; return CE_E_BADARG; }
    int rc = ensure(&h->wsA, &h->wsA_bytes, sizeof(double) * (size_t)B * K);
    if (rc) return rc;
    {
        ProfScope ps(h, 2, st);
        dim3 grid((B + 31) / 32, (K + 31) / 32);
        hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, st, vals, h->wsA, K, B);
    }
    *out = h->wsA;
    return CE_OK;
}

int ce_solve(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *q_vals, long sq_k, long sq_b,
             const ce_settings *settings, double *x, double *y, double *s, int *iters, int *status, double *resid, void *stream) {
    if (!h || B <= 0 || !A_vals || !q_vals || !x || !y || !s || !iters || !status) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    ce_settings S; if (settings) S = *settings; else ce_default_settings(&S);
    const double *Abm = nullptr;
    int rc = to_batch_major(h, B, A_vals, sA_k, sA_b, st, &Abm);
    if (rc) return rc;
    h->retained_A = Abm; h->retained_B = B;
    const DevT &T = h->T;
    double *gA = nullptr, *gG = nullptr;
    if (h->fwd_mode > 0) {
        size_t perA = (h->fwd_mode == 2) ? (size_t)T.m * T.lda : 0, perG = (size_t)T.n * T.ldg;
        rc = ensure(&h->gws, &h->gws_bytes, sizeof(double) * (size_t)B * (perA + perG));
        if (rc) return rc;
        gG = h->gws; gA = h->gws + (size_t)B * perG;
    }
    {
        ProfScope ps(h, 0, st);
        dim3 grid(B), block(NT);
#define LAUNCH_F(AL, GL) hipLaunchKernelGGL((k_forward<AL, GL>), grid, block, h->fwd_lds, st, T, S, Abm, q_vals, sq_k, sq_b, x, y, s, iters, status, resid, gA, gG)
        if (h->fwd_mode == 0) LAUNCH_F(true, true); else if (h->fwd_mode == 1) LAUNCH_F(true, false); else LAUNCH_F(false, false);
#undef LAUNCH_F
    }
    HIPCHK(hipGetLastError());
    return CE_OK;
}

int ce_vjp(ce_handle h, int B, const double *A_vals, long sA_k, long sA_b, const double *q_vals, long sq_k, long sq_b,
           const double *x, const double *y, const double *s, const double *dx, const double *dy,
           double *dA_vals, long sdA_k, long sdA_b, double *dq_vals, long sdq_k, long sdq_b, int *adj_status, void *stream) {
    (void)q_vals; (void)sq_k; (void)sq_b;   // b, c do not enter the adjoint system once r_tau is pinned
    if (!h || B <= 0 || !x || !y || !s || !dx || !dy || !dA_vals || !dq_vals) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const DevT &T = h->T;
    const double *Abm = nullptr;
    int rc;
    if (A_vals) { rc = to_batch_major(h, B, A_vals, sA_k, sA_b, st, &Abm); if (rc) return rc; }
    else { if (!h->retained_A || h->retained_B != B) { g_err = "no retained forward inputs"; return CE_E_STATE; } Abm = h->retained_A; }
    const int K = T.nnz_aug;
    double *dAbm = nullptr; bool need_tr = false;
    if (sdA_k == 1 && sdA_b == K) dAbm = dA_vals;
    else if (sdA_b == 1 && sdA_k == B) { rc = ensure(&h->wsdA, &h->wsdA_bytes, sizeof(double) * (size_t)B * K); if (rc) return rc; dAbm = h->wsdA; need_tr = true; }
    else { g_err = "dA_vals must be contiguous batch-minor or batch-major"; return CE_E_BADARG; }
    double *gA = nullptr, *gK = nullptr;
    if (h->bwd_mode > 0) {
        size_t perA = (h->bwd_mode == 2) ? (size_t)T.m * T.lda : 0, perK = (size_t)h->nkcap * h->ldk;
        rc = ensure(&h->gws, &h->gws_bytes, sizeof(double) * (size_t)B * (perA + perK));
        if (rc) return rc;
        gK = h->gws; gA = h->gws + (size_t)B * perK;
    }
    {
        ProfScope ps(h, 1, st);
        dim3 grid(B), block(NT);
#define LAUNCH_B(AL, KL) hipLaunchKernelGGL((k_backward<AL, KL>), grid, block, h->bwd_lds, st, T, h->nkcap, h->ldk, Abm, x, y, s, dx, dy, dAbm, dq_vals, sdq_k, sdq_b, adj_status, gA, gK)
        if (h->bwd_mode == 0) LAUNCH_B(true, true); else if (h->bwd_mode == 1) LAUNCH_B(true, false); else LAUNCH_B(false, false);
#undef LAUNCH_B
    }
    if (need_tr) {
        ProfScope ps(h, 2, st);
        dim3 grid((K + 31) / 32, (B + 31) / 32);
        hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, st, dAbm, dA_vals, B, K);
    }
    HIPCHK(hipGetLastError());
    return CE_OK;
}

int ce_transpose(ce_handle h, int rows, int cols, const double *in, double *out, void *stream) {
    if (!h || !in || !out || rows <= 0 || cols <= 0) { g_err = "null argument"; return CE_E_BADARG; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    {
        ProfScope ps(h, 2, st);
        dim3 grid((cols + 31) / 32, (rows + 31) / 32);
        hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, st, in, out, rows, cols);
    }
    HIPCHK(hipGetLastError());
    return CE_OK;
}

int ce_set_profiling(ce_handle h, int enable) { if (!h) return CE_E_BADARG; h->prof = enable != 0; return CE_OK; }
int ce_reset_profile(ce_handle h) {
    if (!h) return CE_E_BADARG;
    for (auto &v : h->ev) { for (auto &p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); } v.clear(); }
    return CE_OK;
}
int ce_get_profile(ce_handle h, int which, double *mean_ms, int *launches) {
    if (!h || which < 0 || which > 2) return CE_E_BADARG;
    double tot = 0; int nl = 0;
    for (auto &p : h->ev[which]) { HIPCHK(hipEventSynchronize(p.second)); float ms = 0; HIPCHK(hipEventElapsedTime(&ms, p.first, p.second)); tot += ms; nl++; }
    if (mean_ms) *mean_ms = nl ? tot / nl : 0.0;
    if (launches) *launches = nl;
    return CE_OK;
}
int ce_get_launch_info(ce_handle h, int *fl, int *bl, int *fm, int *bm) {
    if (!h) return CE_E_BADARG;
    if (fl) *fl = (int)h->fwd_lds; if (bl) *bl = (int)h->bwd_lds; if (fm) *fm = h->fwd_mode; if (bm) *bm = h->bwd_mode;
    return CE_OK;
}

}  // extern "C"
