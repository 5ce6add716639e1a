// ce_shared_a.h -- SHARED-A adjoint: diffcp's adjoint solved by LSQR entirely inside ONE kernel, one workgroup per instance.
//
// For templates whose A does not depend on the parameters (only b, c vary: BASELINE configurations 4 and 5) the operator of the reduced
// adjoint system
//        N (r_x, r_y) = ( -A^T r_y ,  DPi (A r_x - r_y) + r_y )  =  ( dx , DPi dy )                (r_tau = 0, oracle/cone_oracle.c adjoint_one)
// uses the SAME sparse matrix for every instance; only DPi = D Pi_K*(y - s) is per instance.  Round 1 ran LSQR on it from Python (every
// operator application two rocBLAS GEMMs over the batch with the DENSE A plus ~30 small torch kernels: launch-bound, 250 ms for 1024
// instances of the 20 x 20 SDP).  Here every LSQR vector of an instance lives in LDS, A is applied from its sparse structure (CSR for A v,
// CSC for A^T v; values and indices stream from L2, shared by all workgroups), the convergence test is evaluated in the kernel, and the
// dense contractions of the PSD cone's derivative   DPi(V)[H] = U (B o (U^T H U)) U^T   run on the matrix cores (ce_psd_mfma.h).
// diffcp itself solves M^T r = dz with LSQR (its default mode); stopping rule and recurrences are Paige & Saunders', as in the oracle.
//
// Cones: zero / nonnegative / second-order / PSD (exponential and power cones take the batched torch path of const_a.py).
#pragma once
#include "ce_shared_a_ops.h"

struct SaStruct {            // sparse structure of the template's A part (device arrays, built once per engine)
    const int *csc_ptr;      // [n + 1]   column starts in the value order of the boundary (CSC of [A_cvx | b_cvx], first nnzA entries)
    const int *csc_row;      // [nnzA]
    const int *csr_ptr;      // [m + 1]
    const int *csr_col;      // [nnzA]
    const int *csr_src;      // [nnzA]    position of the entry in the value order
    int nnzA;
};

constexpr int SA_G = 8;      // lanes per row / column of a sparse product (DPP butterfly over 8 lanes)

// out(i, sum_k vals[src[k]] x[idx[k]]) for every row i of a CSR-like structure; all threads call it
template <class FX, class FO>
__device__ __forceinline__ void sa_spmv(const int *__restrict__ ptr, const int *__restrict__ idx, const int *__restrict__ src,
                                        const double *__restrict__ vals, int nrows, FX &&xv, FO &&out) {
    const int g = threadIdx.x / SA_G, c = threadIdx.x % SA_G;
    for (int i0 = 0; i0 < nrows; i0 += NT / SA_G) {           // uniform trip count: the DPP reduction needs whole groups
        const int i = i0 + g;
        double a = 0;
        if (i < nrows) {
            const int k1 = ptr[i + 1];
            for (int k = ptr[i] + c; k < k1; k += SA_G) a = fma(vals[src ? src[k] : k], xv(idx[k]), a);
        }
        a = group_reduce<SA_G, false>(a);
        if (i < nrows && c == 0) out(i, a);
    }
}

// LDS doubles: [split products: 2 RP + 2 NT] + 8 m + 5 n + 4 nq + ns (4 KP (KP + 1) + 2 KP + 8) + NW * 8 + 16
__host__ __device__ inline size_t sa_lsqr_lds_doubles(int n, int m, int nq, int ns, int maxs, int RP) {
    const int kp = ns > 0 ? psd_mfma_kp(maxs) : 0;
    return (size_t)(RP > 0 ? 2 * RP + 2 * NT : 0) + 8 * (size_t)m + 5 * (size_t)n + 4 * (size_t)(nq > 0 ? nq : 1) +
           (size_t)(ns > 0 ? (2 * ns + 2) * kp * (kp + 1) + 2 * kp + 8 : 0) + NW * 8 + 16;
}

// RP > 0: A is applied through its split into singleton rows and r <= RP dense rows (ce_shared_a_ops.h: balanced, wide loads); RP == 0: through
// the CSR / CSC structure (any sparsity pattern, but rows of very different lengths serialise on the longest).
template <int RP>
__global__ void __launch_bounds__(NT)
k_sa_lsqr(DevT T, SaStruct S, SaSplit F, const double *__restrict__ Avals0, const double *__restrict__ xg, const double *__restrict__ yg,
          const double *__restrict__ sg, const double *__restrict__ dxg, const double *__restrict__ dyg, double *__restrict__ dAo,
          double *__restrict__ dqo, long sdqk, long sdqb, int *__restrict__ adj_status, int *__restrict__ iters_o, double atol, double btol, int itn_lim) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, inst = blockIdx.x;
    const int n = T.n, m = T.m, z = T.z, nl = T.l, nq = T.nq, ns = T.ns;
    double *p = sm;
    double *wyd = p, *vd = p, *part = p;                     // split products: 16-byte aligned at the start of the carve
    if constexpr (RP > 0) { wyd = p; p += RP; vd = p; p += RP; part = p; p += 2 * NT; }
    double *vv = p; p += m;            // v = y - s
    double *uy = p; p += m; double *vy = p; p += m; double *wy = p; p += m; double *ry = p; p += m; double *ty = p; p += m; double *qv = p; p += m; double *tmp = p; p += m;
    double *ux = p; p += n; double *vx = p; p += n; double *wx = p; p += n; double *rx = p; p += n; double *tx = p; p += n;
    double *socs = p; p += 4 * (nq > 0 ? nq : 1);          // per cone: t, |z|, case, z.h
    const int KP = ns > 0 ? psd_mfma_kp(T.maxs) : 0, P = KP + 1, PM = KP * P;
    const float rKP = KP > 0 ? 1.0f / (float)KP : 1.0f;
    double *Um = p; p += (size_t)ns * PM;                    // eigenvectors of smat(v_c), per cone
    double *Bm = p; p += (size_t)ns * PM;                    // divided differences, per cone
    double *Hm = p; p += PM; double *Ym = p; p += PM;        // scratch
    double *cs = p; p += (ns > 0 ? 2 * KP + 8 : 0);
    double *red = p; p += NW * 8;
    const double *x = xg + (size_t)inst * n, *y = yg + (size_t)inst * m, *s = sg + (size_t)inst * m;

    for (int i = tid; i < m; i += NT) vv[i] = y[i] - s[i];
    __syncthreads();
    // ---- per-cone data of DPi
    for (int c = tid >> 6; c < nq; c += NW) {            // one wave per cone
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        const double t = vv[r0]; double nz = 0;
        for (int k = r0 + 1 + (tid & 63); k < r1; k += 64) nz = fma(vv[k], vv[k], nz);
        nz = sqrt(wave_reduce_dpp<false>(nz));
        if ((tid & 63) == 0) {
            socs[4 * c] = t; socs[4 * c + 1] = nz;
            socs[4 * c + 2] = (r1 - r0 == 1) ? (t >= 0 ? 0.0 : 1.0) : (nz <= t ? 0.0 : (nz <= -t ? 1.0 : 2.0));     // 0 identity, 1 zero, 2 boundary
        }
    }
    for (int c = 0; c < ns; c++) {      // eigenvectors and divided differences of every PSD block (cold Jacobi, once)
        const int k = T.sord[c];
        double *U = Um + (size_t)c * PM, *Bc = Bm + (size_t)c * PM;
        const double *zs = vv + T.soff[c];
        for (int idx = tid; idx < KP * KP; idx += NT) {
            const int i = psd_fdiv(idx, rKP), j = idx - i * KP;
            double sv = 0.0;
            if (i < k && j < k) { const int a = i >= j ? i : j, b = i >= j ? j : i; const double v0 = zs[b * k - (b * (b - 1)) / 2 + (a - b)]; sv = (a == b) ? v0 : v0 * M_SQRT1_2; }
            Hm[i * P + j] = sv; U[i * P + j] = (i == j && i < k) ? 1.0 : 0.0;
        }
        __syncthreads();
        psd_sweeps_wg<NT>(Hm, U, k, P, cs, red);
        for (int idx = tid; idx < KP * KP; idx += NT) {
            const int i = psd_fdiv(idx, rKP), j = idx - i * KP;
            double bv = 0.0;
            if (i < k && j < k) {
                const double wi = Hm[i * P + i], wj = Hm[j * P + j];
                if (wi > 0 && wj > 0) bv = 1.0;
                else if (wi <= 0 && wj <= 0) bv = 0.0;
                else { const double den = wi - wj; bv = (fmax(wi, 0.0) - fmax(wj, 0.0)) / (den == 0 ? 1.0 : den); }
            }
            Bc[i * P + j] = bv;
        }
        __syncthreads();
    }
    __syncthreads();

    // q <- DPi(h), in place safe (h may alias q).  All threads call it; ends synchronised.
    auto dproj = [&](const double *h, double *q) {
        if (nq > 0) {
            for (int c = tid >> 6; c < nq; c += NW) {       // z.h per cone: one wave per cone
                double a = 0;
                for (int k = T.qoff[c] + 1 + (tid & 63); k < T.qoff[c + 1]; k += 64) a = fma(vv[k], h[k], a);
                a = wave_reduce_dpp<false>(a);
                if ((tid & 63) == 0) socs[4 * c + 3] = a;
            }
            __syncthreads();
        }
        for (int i = tid; i < m; i += NT) {
            double o = h[i];
            if (i >= z && i < z + nl) o = (vv[i] > 0) ? o : 0.0;
            else {
                const int c = (i >= z + nl && nq > 0) ? T.rowcone[i] : -1;
                if (c >= 0) {
                    const double kase = socs[4 * c + 2];
                    if (kase == 1.0) o = 0.0;
                    else if (kase == 2.0) {
                        const int r0 = T.qoff[c];
                        const double t = socs[4 * c], nz = socs[4 * c + 1], zh = socs[4 * c + 3], h0 = h[r0];
                        const double nzs = fmax(nz, 1e-300);
                        if (i == r0) o = (nz * h0 + zh) / (2 * nzs);
                        else o = (vv[i] * h0 + (t + nz) * h[i] - t * vv[i] * zh / (nzs * nzs)) / (2 * nzs);
                    }
                }
            }
            tmp[i] = o;
        }
        __syncthreads();
        for (int c = 0; c < ns; c++) {      // PSD block: Z = U (B o (U^T H U)) U^T on the matrix cores
            const int k = T.sord[c], off = T.soff[c], KT = KP / 16;
            const double *U = Um + (size_t)c * PM, *Bc = Bm + (size_t)c * PM;
            for (int idx = tid; idx < KP * KP; idx += NT) {
                const int i = psd_fdiv(idx, rKP), j = idx - i * KP;
                double sv = 0.0;
                if (i < k && j < k) { const int a = i >= j ? i : j, b = i >= j ? j : i; const double v0 = h[off + b * k - (b * (b - 1)) / 2 + (a - b)]; sv = (a == b) ? v0 : v0 * M_SQRT1_2; }
                Hm[i * P + j] = sv;
            }
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return Hm[K * P + M]; }, [&](int K, int N) { return U[K * P + N]; }, [&](int M, int N, double v) { Ym[M * P + N] = v; });   // H U
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return U[K * P + M]; }, [&](int K, int N) { return Ym[K * P + N]; }, [&](int M, int N, double v) { Hm[M * P + N] = v * Bc[M * P + N]; });   // B o (U^T H U)
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return U[M * P + K]; }, [&](int K, int N) { return Hm[K * P + N]; }, [&](int M, int N, double v) { Ym[M * P + N] = v; });   // U Y
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return Ym[M * P + K]; }, [&](int K, int N) { return U[N * P + K]; }, [&](int M, int N, double v) { Hm[M * P + N] = v; });   // (U Y) U^T
            __syncthreads();
            for (int idx = tid; idx < k * k; idx += NT) {          // lower triangle (a >= b) -> svec position
                const int a = psd_fdiv(idx, 1.0f / (float)k), b = idx - a * k;
                if (a < b) continue;
                const double v0 = 0.5 * (Hm[a * P + b] + Hm[b * P + a]);
                tmp[off + b * k - (b * (b - 1)) / 2 + (a - b)] = (a == b) ? v0 : v0 * M_SQRT2;
            }
            __syncthreads();
        }
        for (int i = tid; i < m; i += NT) q[i] = tmp[i];
        __syncthreads();
    };
    // solver-form A = -A_cvx: the stored values carry the boundary's sign
    auto A_times = [&](const double *xin, auto &&out) {
        if constexpr (RP > 0) sa_A_times<NT, RP>(F, n, m, xin, part, vd, out);
        else sa_spmv(S.csr_ptr, S.csr_col, S.csr_src, Avals0, m, [&](int j) { return xin[j]; }, [&](int i, double a) { out(i, -a); });
    };
    auto AT_times = [&](const double *yin, auto &&out) {
        if constexpr (RP > 0) sa_AT_times<NT, RP>(F, n, yin, wyd, out);
        else sa_spmv(S.csc_ptr, S.csc_row, (const int *)nullptr, Avals0, n, [&](int i) { return yin[i]; }, [&](int j, double a) { out(j, -a); });
    };
    auto nrm2 = [&](const double *a, const double *b) -> double {
        double r[1] = {0};
        for (int j = tid; j < n; j += NT) r[0] = fma(a[j], a[j], r[0]);
        for (int i = tid; i < m; i += NT) r[0] = fma(b[i], b[i], r[0]);
        block_reduce<1>(r, 0u, red);
        return sqrt(r[0]);
    };
    auto safe = [](double t) -> double { return t > 0 ? t : 1.0; };

    // ---- LSQR (Paige & Saunders) on N r = (dx, DPi dy)
    for (int j = tid; j < n; j += NT) { ux[j] = dxg[(size_t)inst * n + j]; rx[j] = 0.0; }
    for (int i = tid; i < m; i += NT) { ty[i] = dyg[(size_t)inst * m + i]; ry[i] = 0.0; }
    __syncthreads();
    dproj(ty, uy);
    const double bnorm = nrm2(ux, uy);
    double beta = bnorm;
    for (int j = tid; j < n; j += NT) ux[j] /= safe(beta);
    for (int i = tid; i < m; i += NT) uy[i] /= safe(beta);
    __syncthreads();
    // (vx, vy) = N^T u :  q = DPi(uy);  vx = A^T q;  vy = -A ux - q + uy
    dproj(uy, qv);
    AT_times(qv, [&](int j, double a) { vx[j] = a; });
    A_times(ux, [&](int i, double a) { vy[i] = -a - qv[i] + uy[i]; });
    __syncthreads();
    double alfa = nrm2(vx, vy);
    for (int j = tid; j < n; j += NT) { vx[j] /= safe(alfa); wx[j] = vx[j]; }
    for (int i = tid; i < m; i += NT) { vy[i] /= safe(alfa); wy[i] = vy[i]; }
    __syncthreads();
    double rhobar = alfa, phibar = beta, anorm = 0, ddnorm = 0, xxnorm = 0, zz = 0, cs2 = -1, sn2 = 0;
    bool live = bnorm > 0 && alfa * beta > 0;
    int itn = 0;
    while (live && itn < itn_lim) {
        itn++;
        // (tx, ty) = N v :  tx = -A^T vy ;  ty = DPi(A vx - vy) + vy
        AT_times(vy, [&](int j, double a) { tx[j] = -a; });
        A_times(vx, [&](int i, double a) { ty[i] = a - vy[i]; });
        __syncthreads();
        dproj(ty, ty);
        for (int j = tid; j < n; j += NT) ux[j] = tx[j] - alfa * ux[j];
        for (int i = tid; i < m; i += NT) uy[i] = ty[i] + vy[i] - alfa * uy[i];
        __syncthreads();
        beta = nrm2(ux, uy);
        for (int j = tid; j < n; j += NT) ux[j] /= safe(beta);
        for (int i = tid; i < m; i += NT) uy[i] /= safe(beta);
        __syncthreads();
        anorm = sqrt(anorm * anorm + alfa * alfa + beta * beta);
        // (tx, ty) = N^T u
        dproj(uy, qv);
        AT_times(qv, [&](int j, double a) { tx[j] = a; });
        A_times(ux, [&](int i, double a) { ty[i] = -a - qv[i] + uy[i]; });
        __syncthreads();
        for (int j = tid; j < n; j += NT) vx[j] = tx[j] - beta * vx[j];
        for (int i = tid; i < m; i += NT) vy[i] = ty[i] - beta * vy[i];
        __syncthreads();
        alfa = nrm2(vx, vy);
        const double rho = sqrt(rhobar * rhobar + beta * beta);
        const double cs_ = rhobar / safe(rho), sn = beta / safe(rho);
        const double theta = sn * alfa; rhobar = -cs_ * alfa; const double phi = cs_ * phibar; phibar = sn * phibar; const double tau = sn * phi;
        const double t1 = phi / safe(rho), t2 = -theta / safe(rho);
        double rw[1] = {0};
        for (int j = tid; j < n; j += NT) { vx[j] /= safe(alfa); const double w = wx[j]; rw[0] = fma(w, w, rw[0]); rx[j] += t1 * w; wx[j] = vx[j] + t2 * w; }
        for (int i = tid; i < m; i += NT) { vy[i] /= safe(alfa); const double w = wy[i]; rw[0] = fma(w, w, rw[0]); ry[i] += t1 * w; wy[i] = vy[i] + t2 * w; }
        block_reduce<1>(rw, 0u, red);
        ddnorm += rw[0] / safe(rho * rho);
        const double delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * zz, zbar = rhs / safe(fabs(gambar)) * (gambar > 0 ? 1.0 : (gambar < 0 ? -1.0 : 0.0));
        const double xnorm = sqrt(xxnorm + zbar * zbar);
        const double gamma = sqrt(gambar * gambar + theta * theta);
        cs2 = gambar / safe(gamma); sn2 = theta / safe(gamma); zz = rhs / safe(gamma); xxnorm += zz * zz;
        const double rnorm = phibar, arnorm = alfa * fabs(tau);
        const double test1 = rnorm / safe(bnorm), test2 = arnorm / (anorm * rnorm + 1e-300);
        const double rtol = btol + atol * anorm * xnorm / safe(bnorm);
        if (test1 <= rtol || test2 <= atol) live = false;
    }
    __syncthreads();
    // ---- outputs in the boundary convention: dA_eval = [-dA.data, db[b_idx]], dq_eval = [dc, 0]  (diffcp_if.py:91-92);
    //      dA_ij = x_j r_y,i - y_i r_x,j ,  db = -r_y ,  dc = -r_x
    double *dA = dAo + (size_t)inst * T.nnz_aug;
    for (int k = tid; k < T.nnz_aug; k += NT) {
        const int r = T.rowidx[k], c = T.colidx[k];
        dA[k] = (c < n) ? -(x[c] * ry[r] - y[r] * rx[c]) : -ry[r];
    }
    for (int j = tid; j <= n; j += NT) dqo[j * sdqk + inst * sdqb] = (j < n) ? -rx[j] : 0.0;
    if (tid == 0) { if (adj_status) adj_status[inst] = live ? 1 : 0; if (iters_o) iters_o[inst] = itn; }
}
