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

// D = A B on KT x KT tiles of 16 x 16; fa(M, K), fb(K, N): operand elements, out(M, N, v): result sink.  All waves of the workgroup call it.
template <int NTH, class FA, class FB, class FO>
__device__ __forceinline__ void psd_mfma_gemm(int KT, FA &&fa, FB &&fb, FO &&out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lg = lane >> 4, lc = lane & 15;
    for (int t = wave; t < KT * KT; t += NTH / 64) {
        const int ti = t / KT, tj = t - ti * KT;
        psd_v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int s = 0; s < 4 * KT; s++)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(16 * ti + lc, 4 * s + lg), fb(4 * s + lg, 16 * tj + lc), acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) out(16 * ti + lg + 4 * r, 16 * tj + lc, acc[r]);
    }
}

// cyclic Jacobi sweeps on S (k x k, pitch P), rotations accumulated into the columns of V (pitch P); same tournament order and
// rotation formulas as psd_jacobi.  On return diag(S) holds the eigenvalues and the columns of V the eigenvectors.
template <int NTH>
__device__ __forceinline__ void psd_sweeps(double *Sm, double *Vm, int k, int P, double *cs, double *red) {
    constexpr int NT = NTH, NW = NTH / 64;
    const int tid = threadIdx.x;
    const int K = (k + 1) & ~1;
    for (int sweep = 0; sweep < 40; sweep++) {
        double r[2] = {0, 0};
        for (int idx = tid; idx < k * k; idx += NT) { const int i = idx / k, j = idx - i * k; const double v = Sm[i * P + j]; if (i == j) r[1] = fma(v, v, r[1]); else r[0] = fma(v, v, r[0]); }
        block_reduce_n<2, NW>(r, 0u, red);
        if (r[0] <= 1e-30 * (r[0] + r[1]) || r[0] == 0.0) break;          // uniform
        for (int rd = 0; rd < K - 1; rd++) {
            if (tid < K / 2) {
                int p = (tid == 0) ? K - 1 : (rd + tid) % (K - 1);
                int q = (rd + K - 1 - tid) % (K - 1);
                if (p > q) { const int t_ = p; p = q; q = t_; }
                double c = 1.0, sn = 0.0;
                if (q < k) {
                    const double apq = Sm[p * P + q];
                    if (apq != 0.0) {
                        const double theta = (Sm[q * P + q] - Sm[p * P + p]) / (2 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                        c = 1 / sqrt(t * t + 1); sn = t * c;
                    }
                } else { p = -1; }
                cs[4 * tid] = c; cs[4 * tid + 1] = sn; cs[4 * tid + 2] = (double)p; cs[4 * tid + 3] = (double)q;
            }
            __syncthreads();
            for (int idx = tid; idx < (K / 2) * k * 2; idx += NT) {          // column pass on S and V
                const int which = idx / ((K / 2) * k), rem = idx - which * (K / 2) * k;
                const int pi = rem / k, row = rem - pi * k;
                const int p = (int)cs[4 * pi + 2], q = (int)cs[4 * pi + 3];
                if (p < 0) continue;
                const double c = cs[4 * pi], sn = cs[4 * pi + 1];
                double *M = which ? Vm : Sm;
                const double a = M[row * P + p], b = M[row * P + q];
                M[row * P + p] = c * a - sn * b; M[row * P + q] = sn * a + c * b;
            }
            __syncthreads();
            for (int idx = tid; idx < (K / 2) * k; idx += NT) {              // row pass on S
                const int pi = idx / k, col = idx - pi * k;
                const int p = (int)cs[4 * pi + 2], q = (int)cs[4 * pi + 3];
                if (p < 0) continue;
                const double c = cs[4 * pi], sn = cs[4 * pi + 1];
                const double a = Sm[p * P + col], b = Sm[q * P + col];
                Sm[p * P + col] = c * a - sn * b; Sm[q * P + col] = sn * a + c * b;
            }
            __syncthreads();
        }
    }
}

// LDS doubles needed: 3 * KP * (KP + 1) + 2 * k + 8  (+ the reduction scratch of block_reduce_n)
__host__ __device__ inline int psd_mfma_kp(int k) { return 16 * ((k + 15) / 16); }

// zsvec (svec of S, lower triangle column-major, sqrt(2) off-diagonals) is replaced by svec(Pi_PSD(S)).
// Vstate: k * k doubles of global memory, the eigenvectors of the previous call (row-major), or NULL; warm != 0: start from them.
template <int NTH>
__device__ __forceinline__ void psd_project_mfma(double *zsvec, int k, double *Sm, double *Vm, double *Tm, double *cs, double *red,
                                                 double *Vstate, int warm) {
    constexpr int NT = NTH;
    const int tid = threadIdx.x;
    const int KP = psd_mfma_kp(k), P = KP + 1, KT = KP / 16;
    const bool use_prev = warm && Vstate != nullptr;
    for (int idx = tid; idx < KP * KP; idx += NT) {
        const int i = idx / KP, j = idx - i * KP;
        double sv = 0.0, vv = 0.0;
        if (i < k && j < k) {
            const int a = i >= j ? i : j, b = i >= j ? j : i;                 // lower-triangle entry (a, b), column-major packed
            const double v = zsvec[b * k - (b * (b - 1)) / 2 + (a - b)];
            sv = (a == b) ? v : v * M_SQRT1_2;
            vv = use_prev ? Vstate[i * k + j] : (i == j ? 1.0 : 0.0);
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
            const int i = idx / k, j = idx - i * k;
            if (i > j) { const double a = 0.5 * (Sm[i * P + j] + Sm[j * P + i]); Sm[i * P + j] = a; Sm[j * P + i] = a; }
        }
        __syncthreads();
    }
    psd_sweeps<NTH>(Sm, Vm, k, P, cs, red);
    for (int i = tid; i < KP; i += NT) cs[i] = i < k ? fmax(Sm[i * P + i], 0.0) : 0.0;
    __syncthreads();
    // X = (V diag(w+)) V^T
    psd_mfma_gemm<NTH>(KT, [&](int M, int K) { return Vm[M * P + K] * cs[K]; }, [&](int K, int N) { return Vm[N * P + K]; },
                       [&](int M, int N, double v) { Tm[M * P + N] = v; });
    __syncthreads();
    for (int pos = tid; pos < k * (k + 1) / 2; pos += NT) {
        int b = 0, rem = pos;
        while (rem >= k - b) { rem -= k - b; b++; }
        const int a = b + rem;
        const double v = 0.5 * (Tm[a * P + b] + Tm[b * P + a]);
        zsvec[pos] = (a == b) ? v : v * M_SQRT2;
    }
    if (Vstate != nullptr)
        for (int idx = tid; idx < k * k; idx += NT) { const int i = idx / k, j = idx - i * k; Vstate[idx] = Vm[i * P + j]; }
    __syncthreads();
}
