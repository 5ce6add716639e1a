// ce_backward_rt.h -- register-tiled structured direct adjoint (hot path for instances with n + min(m,n) + 1 <= 16*TJ).
//
// Same mathematics as ce_backward.h (diffcp adjoint M^T r = dz, r_tau pinned to 0, reduced cone block by cone block to the
// saddle system  K [r_x; mu] = [f; d_B],  K = [[H, -B^T],[B, 0]]  of order NK = n + #equality rows), different machine mapping:
//   * K (with the right-hand side as column NK) lives in REGISTERS: the 256 threads of a workgroup form a 16 x 16 grid,
//     thread (ra, cb) owns K[ra + 16 i][cb + 16 j], i < TI, j < TJ (2-D cyclic: the live part of every thread's tile shrinks
//     evenly as elimination proceeds, and the outer loop over j-slots is unrolled so the shrinking is static).
//   * Gauss-Jordan with partial pivoting: per pivot the 16 lanes that own column k (one DPP row of one wave) find the pivot
//     with a 4-stage butterfly and publish the column through LDS; the owners of the pivot row publish the row; every thread
//     then applies the rank-1 update to its tile.  Two workgroup barriers per pivot, no integer division, no LDS-resident K.
//   * LDS holds only the dense instance matrix (needed for assembly and for q = A r_x) and vectors: ~52 KB per workgroup at
//     the metric configuration -> 3 workgroups (12 waves) per CU instead of 1.
//   * H = sum_c theta_c (A_c^T A_c - a_y a_y^T - a_s a_s^T) is accumulated directly in tile layout with 4x4-style register
//     blocking (rows {ra+16i} x columns {cb+16j}: TI + TJ LDS reads feed TI*TJ FMAs per cone row).
#pragma once

// v_max_f64 without the canonicalisation fmax() puts in front of it (operands are finite by construction)
__device__ __forceinline__ double vmax_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// max(a, |b|) with the bare instruction (b's sign bit may carry anything)
__device__ __forceinline__ double vmax_abs(double a, double b) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }

constexpr int BG = 16;   // default thread grid is BG x BG
constexpr int BGC = 16;  // column residues (always one DPP row wide)

#ifdef CE_TIMING   // debug build: phase durations (shader cycles) of every workgroup overwrite the first entries of its dA row
#define CE_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) tstamp[i] = __builtin_readcyclecounter(); } while (0)
// cycles per phase of one pivot of the elimination, accumulated over the pivots (this wave's clock)
#define CE_BACC(k) do { const long long t1_ = __builtin_readcyclecounter(); bacc[k] += t1_ - bt0; bt0 = t1_; } while (0)
#else
#define CE_STAMP(i) do { } while (0)
#define CE_BACC(k) do { } while (0)
#endif

__host__ __device__ inline int bwd_rt_union_doubles(int n, int m, int nqs, int TI, int TJ, int BGR = 16) {
    int a = 2 * nqs * n, b = 2 * BGR * TI + 2 * BGC * TJ, c = (NT > m ? NT : m) + m;
    int r = a > b ? a : b;
    return r > c ? r : c;
}

template <int TI, int TJ, int TH, bool PSD = false, int BGR = 16>
__global__ void __launch_bounds__(BGR * 16, (BGR == 16 ? 3 : 2))
k_backward_rt(DevT T, const double *__restrict__ Avals, const double *__restrict__ xg, const double *__restrict__ yg,
              const double *__restrict__ sg, const double *__restrict__ dxg, const double *__restrict__ dyg,
              double *__restrict__ dAo, double *__restrict__ dqo, long sdqk, long sdqb, int *__restrict__ adj_status,
              const double *__restrict__ Pvals_g = nullptr, int nnzP = 0, const int *__restrict__ pmap = nullptr,
              const int *__restrict__ prow = nullptr, const int *__restrict__ pcol = nullptr, int p_tri = 0,
              double *__restrict__ dPo = nullptr, int retry = 0, int *__restrict__ nk_max = nullptr, int *__restrict__ fix = nullptr, int nonfinal = 0) {
    // retry: second launch of a two-tile plan (cone_engine.hip ce_vjp_qp): a SMALLER tile variant has already served every instance whose system fits it
    // and flagged the others (adj_status 2, zero gradient); this launch -- the template's worst-case tile -- recomputes the flagged ones only.
    if (retry && adj_status[blockIdx.x] != 2) return;
    // Quadratic objective (Pvals_g != nullptr): the reduced adjoint system is [[H + P, -B^T], [B, 0]] (x-block of M^T r = dz gains
    // P r_x) and dP = -sym(r_x x^T) (oracle/cone_oracle.c adjoint_one with r_tau = 0).  pmap: n x n map to the entries of the P
    // structure (-1: structural zero); a one-triangle structure (p_tri) maps (i,j) and (j,i) to one entry, whose gradient is doubled.
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, inst = blockIdx.x;
    const int n = T.n, m = T.m, lda = T.lda, nq = T.nq, z = T.z;
    const int nqs = nq > 0 ? nq : 1;
    constexpr int NTB = BGR * BGC, NWB = NTB / 64;      // threads / waves of this instantiation
    const int ra = tid & (BGR - 1), cb = tid / BGR;
    constexpr int NKCAP = BGC * TJ - 1;          // column NK (the right-hand side) must exist: NK <= 16*TJ - 1, rows NK <= 16*TI
    constexpr int THI = (TH * BGC + BGR - 1) / BGR;      // row slots of the H block (columns: TH slots of 16; rows: slots of BGR)
    static_assert(THI <= TI && TH <= TJ, "the H block lives inside the tile");

    // ---- LDS carve.  U is a union region: {a_y, a_s} during assembly, {colbuf, rowbuf} during elimination, {part, A r_x} after.
    double *p = sm;
    double *A = p; p += m * lda;
    double *bv = p; p += m;          // b (unused by the adjoint) ; later: multipliers mu[e]
    double *vv = p; p += m;          // v = y - s ; later r_y
    double *dv = p; p += m;          // d = DPi dy
    double *rx = p; p += n;
    double *fvec = p; p += n;        // sum over boundary cones of [ a_s (e_s.d) + A_c^T P d / (1 - lam) ]
    double *cinfo = p; p += 6 * nqs; // per cone: lambda, nz, e_y.d, e_s.d
    double *pivrow = p; p += BGR * TI;   // pivot value of the row that served as pivot
    double *pinfo = p + ((p - sm) & 1); p = pinfo + 4;      // per buffer one 16-byte record {pivot value, pivot row}: ONE ds_read_b128 after the pivot's barrier
    double *red = p; p += NWB * 8;
    double *psdU = p, *psdEv = p, *lamr = p, *psdScr = p, *expW = p;     // PSD: eigenvectors per cone, eigenvalues, DPi eigenvalue per rotated row, scratch
    if constexpr (PSD) { psdU = p; p += T.ns * T.maxs * T.maxs; psdEv = p; p += T.ns * T.maxs; lamr = p; p += m; psdScr = p; p += 2 * NWB * T.maxs * T.maxs + 2 * T.maxs + 8;   /* one (X, W) pair per wave */ expW = p; p += 9 * (T.nep + T.np); }
    double *U = p; p += bwd_rt_union_doubles(n, m, nqs, TI, TJ, BGR);
    double *ay = U, *as = U + nqs * n;                              // A_c^T e_y, A_c^T e_s
    double *colbuf = U, *rowbuf = U + 2 * BGR * TI;
    double *part = U, *qv2 = U + max(NT, m);                        // A r_x
    int *ip = (int *)p;
    int *rkind = ip; ip += m;
    int *eqrow = ip; ip += m;
    int *ckind = ip; ip += nqs;
    int *ceq = ip; ip += nqs;
    int *esrc = ip; ip += BGC * TJ;      // equality e -> source: row index (>= 0) or -1 - cone
    int *colof = ip; ip += BGR * TI;     // pivot row r -> column it eliminated
    int *wcnt = ip; ip += NWB + 1;
    int *misc = ip; ip += 8;            // [0] n_eq, [2] flags, [4],[5] pivot row per buffer

#ifdef CE_TIMING
    __shared__ long long tstamp[12];
    long long bacc[6] = {0, 0, 0, 0, 0, 0}, bt0 = 0;
#endif
    CE_STAMP(0);
    load_instance<20>(T, Avals + (size_t)inst * T.nnz_aug, A, bv);      // (20 entries per lane in flight: the metric configuration's 5100 values in ONE round trip instead of three)
    for (int i = tid; i < m; i += NTB) { vv[i] = yg[(size_t)inst * m + i] - sg[(size_t)inst * m + i]; dv[i] = dyg[(size_t)inst * m + i]; }      // dv: the incoming dy for now (one coalesced
    if (tid < 8) misc[tid] = 0;                                                                                                                    // pass; the per-cone code below used to read it from global memory row by row)
    __syncthreads();
    CE_STAMP(1);
    // ---- classify
    for (int i = tid; i < z + T.l; i += NTB) rkind[i] = (i < z || vv[i] > 0) ? RK_EQ : RK_FREE;
    for (int c = tid; c < nq; c += NTB) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1], d = r1 - r0;
        int kind; double lam = 0, nz = 0;
        if (d == 1) kind = vv[r0] >= 0 ? 0 : 1;
        else {
            double nz1 = 0;
            for (int i = r0 + 1; i < r1; i += 4) {     // four rows per step, reads requested together (clamped index, masked contribution)
                double w[4];
#pragma unroll
                for (int u = 0; u < 4; u++) w[u] = vv[min(i + u, r1 - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) { const double t = (i + u < r1) ? w[u] : 0.0; if (u & 1) nz1 = fma(t, t, nz1); else nz = fma(t, t, nz); }
            }
            nz = sqrt(nz + nz1);
            const double t0 = vv[r0];
            if (nz <= t0) kind = 0; else if (nz <= -t0) kind = 1; else { kind = 2; lam = (t0 + nz) / (2 * nz); }
        }
        ckind[c] = kind; cinfo[6 * c] = lam; cinfo[6 * c + 1] = nz;
        for (int i = r0; i < r1; i++) rkind[i] = kind == 0 ? RK_EQ : (kind == 1 ? RK_FREE : RK_SOCB);
    }
    __syncthreads();
    if constexpr (PSD) {
        // PSD cones: V = smat(v_c) = U Lambda U^T.  DPi(v) is diagonal in the orthonormal basis E_ab = svec(sym(u_a u_b^T)) with
        // eigenvalue B_ab (1: both eigenvalues positive, 0: both non-positive, l+/(l+ - l-): mixed).  The rows of the cone are
        // ROTATED into that basis in place, A_c <- Q^T A_c with Q^T x = svec(U^T smat(x) U): afterwards every rotated row is an
        // ordinary equality (B=1) / free (B=0) / weighted (theta = B/(1-B)) row and the generic machinery below applies.
        const int lane = tid & 63, wid = tid >> 6;
        for (int c = 0; c < T.ns; c++) {
            const int k = T.sord[c], r0 = T.soff[c], d = k * (k + 1) / 2;
            double *Um = psdU + c * T.maxs * T.maxs, *ev = psdEv + c * T.maxs;
            psd_jacobi<NTB>(vv + r0, k, psdScr, Um, psdScr + 2 * T.maxs * T.maxs, red);     // eigenvalues on diag(psdScr), vectors in Um
            for (int i = tid; i < k; i += NTB) ev[i] = psdScr[i * k + i];
            __syncthreads();
            // rotate the n columns of A_c and (as column n) the incoming dy_c; one column per wave at a time
            for (int g0 = 0; g0 <= n; g0 += NWB) {
                const int col = g0 + wid;
                double *X = psdScr + wid * 2 * T.maxs * T.maxs, *W = X + T.maxs * T.maxs;
                if (col <= n) {
                    for (int idx = lane; idx < k * k; idx += 64) {
                        const int i = idx / k, j = idx - i * k, a = i >= j ? i : j, b = i >= j ? j : i;
                        const int pos = b * k - (b * (b - 1)) / 2 + (a - b);
                        const double v = (col < n) ? A[(r0 + pos) * lda + col] : dyg[(size_t)inst * m + r0 + pos];
                        X[idx] = (a == b) ? v : v * M_SQRT1_2;
                    }
                }
                __syncthreads();
                if (col <= n) {
                    for (int idx = lane; idx < k * k; idx += 64) {       // W = X U
                        const int i = idx / k, j = idx - i * k;
                        double acc = 0; for (int a = 0; a < k; a++) acc = fma(X[i * k + a], Um[a * k + j], acc);
                        W[idx] = acc;
                    }
                }
                __syncthreads();
                if (col <= n) {
                    for (int pos = lane; pos < d; pos += 64) {           // T = U^T W, packed back as svec
                        int b = 0, rem = pos; while (rem >= k - b) { rem -= k - b; b++; }
                        const int a = b + rem;
                        double acc = 0; for (int i = 0; i < k; i++) acc = fma(Um[i * k + a], W[i * k + b], acc);
                        const double t = (a == b) ? acc : acc * M_SQRT2;
                        if (col < n) A[(r0 + pos) * lda + col] = t;
                        else {
                            const double la = ev[a], lb = ev[b];
                            const double Bv = (la > 0 && lb > 0) ? 1.0 : ((la <= 0 && lb <= 0) ? 0.0 : fmax(la, lb) / (fmax(la, lb) - fmin(la, lb)));
                            lamr[r0 + pos] = Bv; dv[r0 + pos] = Bv * t;
                            rkind[r0 + pos] = (Bv == 1.0) ? RK_EQ : (Bv == 0.0 ? RK_FREE : RK_MIX);
                        }
                    }
                }
                __syncthreads();
            }
        }
            // Exponential / power cones: S = D Pi_K*(v_c) = W diag(theta) W^T (3x3, one thread per cone).  The triple's rows are rotated,
        // A_c <- W^T A_c, exactly like a PSD block: afterwards each row is an equality (theta = 1), free (0) or weighted row.
        if (T.nep + T.np > 0) {
            for (int c = tid; c < T.nep + T.np; c += NTB) {
                const int r0 = T.eoff + 3 * c;
                double W[9], th[3];
                if (c < T.nep) exp_dual_eig(vv + r0, W, th); else pow_dual_eig(vv + r0, T.pw[c - T.nep], W, th);
                const double *h = dyg + (size_t)inst * m + r0;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const double t = W[a] * h[0] + W[3 + a] * h[1] + W[6 + a] * h[2];     // (W^T dy)_a
                    double Bv = th[a];
                    if (Bv < 1e-9) Bv = 0.0; else if (Bv > 1.0 - 1e-9) Bv = 1.0;
                    lamr[r0 + a] = Bv; dv[r0 + a] = Bv * t;
                    rkind[r0 + a] = (Bv == 1.0) ? RK_EQ : (Bv == 0.0 ? RK_FREE : RK_MIX);
                }
#pragma unroll
                for (int k = 0; k < 9; k++) expW[9 * c + k] = W[k];
            }
            __syncthreads();
            for (int idx = tid; idx < (T.nep + T.np) * n; idx += NTB) {
                const int c = idx / n, j = idx - c * n, r0 = T.eoff + 3 * c;
                const double *W = expW + 9 * c;
                const double a0 = A[r0 * lda + j], a1 = A[(r0 + 1) * lda + j], a2 = A[(r0 + 2) * lda + j];
#pragma unroll
                for (int a = 0; a < 3; a++) A[(r0 + a) * lda + j] = W[a] * a0 + W[3 + a] * a1 + W[6 + a] * a2;
            }
            __syncthreads();
        }
}
    // ---- equality numbering: ballot prefix sums (rows in order, then one e_y row per boundary cone)
    {
        int base = 0;
        for (int i0 = 0; i0 < m; i0 += NTB) {
            const int i = i0 + tid;
            const bool f = (i < m) && (rkind[i] == RK_EQ);
            const unsigned long long bal = __ballot(f);
            const int lane = tid & 63, wid = tid >> 6;
            if (lane == 0) wcnt[wid] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wid; w++) off += wcnt[w];
            int tot = 0;
            for (int w = 0; w < NWB; w++) tot += wcnt[w];
            if (i < m) {
                const int e = f ? off + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
                eqrow[i] = e;
                if (f && e < BGC * TJ) esrc[e] = i;
            }
            base += tot;
            __syncthreads();
        }
        if (tid == 0) {
            int ne = base;
            for (int c = 0; c < nq; c++) { if (ckind[c] == 2) { if (ne < BGC * TJ) esrc[ne] = -1 - c; ceq[c] = ne++; } else ceq[c] = -1; }
            misc[0] = ne;
        }
        __syncthreads();
    }
    const int neq = misc[0];
    const int NK = n + neq;
    if (nk_max && tid == 0) atomicMax(nk_max, NK);          // (the host sizes the NEXT call's first tile by it)
    if (NK > NKCAP || NK > BGR * TI) {   // more active rows than the register tile holds: degenerate instance (flagged, zero gradient)
        for (int k = tid; k < T.nnz_aug; k += NTB) dAo[(size_t)inst * T.nnz_aug + k] = 0.0;
        for (int j = tid; j <= n; j += NTB) dqo[j * sdqk + inst * sdqb] = 0.0;
        if (dPo) for (int k = tid; k < nnzP; k += NTB) dPo[(size_t)inst * nnzP + k] = 0.0;
        if (tid == 0 && adj_status) adj_status[inst] = 2;
        if (tid == 0 && fix && !nonfinal) fix[1 + atomicAdd(fix, 1)] = inst;      // (the LSQR launch behind this kernel serves it)
        return;
    }
    CE_STAMP(2);
    // ---- d = DPi(v) dy, per-cone scalars
    // (dv holds dy: transformed in place; PSD / exponential rows were already replaced by the rotation above)
    for (int i = tid; i < z + T.l; i += NTB) { if (rkind[i] != RK_EQ) dv[i] = 0.0; }
    for (int c = tid; c < nq; c += NTB) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        if (ckind[c] == 0) { /* dv = dy */ }
        else if (ckind[c] == 1) { for (int i = r0; i < r1; i++) dv[i] = 0.0; }
        else {
            // one pass over the cone, four rows per step with their reads in flight together:  z.h  first, then d_i and z.d in the same sweep
            const double t0 = vv[r0], nz = cinfo[6 * c + 1], h0 = dv[r0];
            double zh = 0, zh1 = 0;
            for (int i = r0 + 1; i < r1; i += 4) {
                double w[4], hh[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int iu = min(i + u, r1 - 1); w[u] = vv[iu]; hh[u] = dv[iu]; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const double t = (i + u < r1) ? w[u] : 0.0; if (u & 1) zh1 = fma(t, hh[u], zh1); else zh = fma(t, hh[u], zh); }
            }
            zh += zh1;
            const double i2n = 1.0 / (2 * nz), cz = t0 * zh / (nz * nz);
            const double d0 = (nz * h0 + zh) * i2n;
            double zd = 0, zd1 = 0;
            for (int i = r0 + 1; i < r1; i += 4) {
                double w[4], hh[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int iu = min(i + u, r1 - 1); w[u] = vv[iu]; hh[u] = dv[iu]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (i + u < r1) {
                        const double di = (w[u] * h0 + (t0 + nz) * hh[u] - w[u] * cz) * i2n;
                        dv[i + u] = di;
                        if (u & 1) zd1 = fma(w[u], di, zd1); else zd = fma(w[u], di, zd);
                    }
                }
            }
            dv[r0] = d0;
            zd = (zd + zd1) / nz;
            cinfo[6 * c + 2] = (d0 + zd) * M_SQRT1_2;   // e_y . d
            cinfo[6 * c + 3] = (d0 - zd) * M_SQRT1_2;   // e_s . d
        }
    }
    __syncthreads();
    // ---- a_y, a_s for boundary cones
    for (int idx = tid; idx < nq * n; idx += NTB) {
        const int c = idx / n, j = idx - c * n;
        if (ckind[c] != 2) continue;
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        const double inz = 1.0 / cinfo[6 * c + 1];
        double a = 0, a1 = 0;
        const double a00 = A[r0 * lda + j];
        for (int i = r0 + 1; i < r1; i += 4) {       // four rows per step, reads requested together (index clamped, contribution masked): a dependent loop costs one LDS round trip per row
            double av[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int iu = min(i + u, r1 - 1); av[u] = A[iu * lda + j]; wv[u] = vv[iu]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const double w = (i + u < r1) ? wv[u] : 0.0; if (u & 1) a1 = fma(av[u], w, a1); else a = fma(av[u], w, a); }
        }
        a = (a + a1) * inz;
        ay[idx] = (a00 + a) * M_SQRT1_2;
        as[idx] = (a00 - a) * M_SQRT1_2;
    }
    __syncthreads();
    // ---- fvec[j] = sum_c [ a_s (e_s.d) + (A_c^T d - a_y (e_y.d) - a_s (e_s.d)) / (1 - lam) ]
    //      4 lanes per column, each takes every 4th cone; fixed summation order (deterministic)
    for (int j0 = 0; j0 < n; j0 += NTB / 4) {
        const int j = j0 + (tid >> 2), part = tid & 3;
        double acc = 0;
        if (j < n) {
            for (int c = part; c < nq; c += 4) {
                if (ckind[c] != 2) continue;
                const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
                const double lam = cinfo[6 * c], eyd = cinfo[6 * c + 2], esd = cinfo[6 * c + 3];
                double g = 0, g1 = 0;
                const double ayj = ay[c * n + j], asj = as[c * n + j];
                for (int i = r0; i < r1; i += 4) {
                    double av[4], wv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int iu = min(i + u, r1 - 1); av[u] = A[iu * lda + j]; wv[u] = dv[iu]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) { const double w = (i + u < r1) ? wv[u] : 0.0; if (u & 1) g1 = fma(av[u], w, g1); else g = fma(av[u], w, g); }
                }
                g += g1;
                const double a = g - ayj * eyd - asj * esd;      // A_c^T P d
                acc += asj * esd + a / (1 - lam);
            }
            if constexpr (PSD) {
                for (int t = T.soff[0] + part; t < T.eoff + 3 * (T.nep + T.np); t += 4)
                    if (rkind[t] == RK_MIX) acc = fma(A[t * lda + j], dv[t] / (1 - lamr[t]), acc);
            }
        }
        acc = group_reduce<4, false>(acc);
        if (j < n && part == 0) fvec[j] = acc;
    }
    __syncthreads();
    CE_STAMP(3);
    // ---- assemble the register tile of [K | rhs]
    double kt[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++) kt[i][j] = 0.0;
    // H block (r < n, c < n): H = sum_c theta_c (A_c^T A_c - a_y a_y^T - a_s a_s^T), accumulated straight into the tile
    if constexpr (BGR == 16) {
        // ... on the MATRIX CORES, with the accumulators landing in the 2-D cyclic tile layout directly.  H = sum_k w_k a_k a_k^T over the rows of the boundary
        // cones (w = theta_c) and, per cone, a_y and a_s (w = -theta_c).  v_mfma_f64_16x16x4_f64 computes D[M][N] += sum_k Aop[M][k] Bop[k][N] (4 rows k per
        // instruction) and leaves D[(l >> 4) + 4 r][l & 15] in register r of lane l.  Thread (ra, cb) = (l & 15, 4 wave + (l >> 4)) owns K[ra + 16 i][cb + 16 j]:
        // with N <-> the row residue (p = N + 16 i) and M <-> the columns of THIS wave (q = 4 wave + (M & 3) + 16 (M >> 2)), register r of the accumulator of
        // row block i IS kt[i][r] -- no staging, no exchange.  (Round 3 formed the same Gram matrix on the matrix cores but had to move it through global
        // memory into the tile layout, which ate the gain; the scalar accumulation it replaces was 52 k of the kernel's 297 k cycles: 8 LDS reads per 16 FMAs.)
        typedef double v4d __attribute__((ext_vector_type(4)));
        constexpr int NCS = (TH + 3) / 4;                     // column sets of four slots
        v4d acc[THI][NCS];
#pragma unroll
        for (int i = 0; i < THI; i++)
#pragma unroll
            for (int s2 = 0; s2 < NCS; s2++) acc[i][s2] = v4d{0.0, 0.0, 0.0, 0.0};
        const int lane = tid & 63, l15 = lane & 15, l4 = lane >> 4, wv = tid >> 6;
        int qa[NCS];
#pragma unroll
        for (int s2 = 0; s2 < NCS; s2++) qa[s2] = 4 * wv + (l15 & 3) + 16 * ((l15 >> 2) + 4 * s2);
        for (int c = 0; c < nq; c++) {
            if (ckind[c] != 2) continue;                       // uniform
            const double lam = cinfo[6 * c], th = lam / (1 - lam);
            const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
            for (int k0 = r0; k0 < r1 + 2; k0 += 4) {          // the cone's rows, then a_y, a_s; four per instruction, this lane supplies row k0 + (l >> 4)
                const int kk = k0 + l4;
                const double *src = kk < r1 ? A + kk * lda : (kk == r1 ? ay + c * n : as + c * n);
                const double wgt = kk < r1 ? th : (kk < r1 + 2 ? -th : 0.0);
                double av[NCS], bv[THI];
#pragma unroll
                for (int s2 = 0; s2 < NCS; s2++) { const double v = src[qa[s2] < n ? qa[s2] : 0]; av[s2] = qa[s2] < n ? wgt * v : 0.0; }
#pragma unroll
                for (int i = 0; i < THI; i++) { const int pb = l15 + 16 * i; const double v = src[pb < n ? pb : 0]; bv[i] = pb < n ? v : 0.0; }
#pragma unroll
                for (int i = 0; i < THI; i++)
#pragma unroll
                    for (int s2 = 0; s2 < NCS; s2++) acc[i][s2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[i], acc[i][s2], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < THI; i++)
#pragma unroll
            for (int s2 = 0; s2 < NCS; s2++)
#pragma unroll
                for (int r = 0; r < 4; r++) if (4 * s2 + r < TH) kt[i][4 * s2 + r] = acc[i][s2][r];
    } else {
    for (int c = 0; c < nq; c++) {
        if (ckind[c] != 2) continue;                       // uniform
        const double lam = cinfo[6 * c], th = lam / (1 - lam);
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
#pragma unroll 4
        for (int ii = r0; ii < r1; ii++) {
            const double *row = A + ii * lda;
            double ar[THI], ac[TH];
            // unguarded reads (a guarded read is a branch per entry, a select two more VALU operations per value): an index past the row's n entries stays
            // inside the LDS carve (the following row, or the vectors behind the last one), and the tile entries it pollutes -- row or column index
            // >= n -- are reset after the accumulation
#pragma unroll
            for (int i = 0; i < THI; i++) ar[i] = th * row[ra + BGR * i];
#pragma unroll
            for (int j = 0; j < TH; j++) ac[j] = row[cb + BGC * j];
#pragma unroll
            for (int i = 0; i < THI; i++)
#pragma unroll
                for (int j = 0; j < TH; j++) kt[i][j] = fma(ar[i], ac[j], kt[i][j]);
        }
        const double *ayc = ay + c * n, *asc = as + c * n;
#pragma unroll
        for (int i = 0; i < THI; i++)
#pragma unroll
            for (int j = 0; j < TH; j++) {
                const int r = ra + BGR * i, cc = cb + BGC * j;
                if (r < n && cc < n) kt[i][j] = fma(-th, ayc[r] * ayc[cc] + asc[r] * asc[cc], kt[i][j]);
            }
    }
    }
#pragma unroll
    for (int i = 0; i < THI; i++)
#pragma unroll
        for (int j = 0; j < TH; j++) if (ra + BGR * i >= n || cb + BGC * j >= n) kt[i][j] = 0.0;
    if constexpr (PSD) {   // weighted rows of rotated PSD blocks: H += theta_t a_t^T a_t
        for (int t = T.soff[0]; t < T.eoff + 3 * (T.nep + T.np); t++) {
            if (rkind[t] != RK_MIX) continue;              // uniform
            const double th = lamr[t] / (1 - lamr[t]);
            const double *row = A + t * lda;
            double ar[THI], ac[TH];
#pragma unroll
            for (int i = 0; i < THI; i++) ar[i] = (ra + BGR * i < n) ? th * row[ra + BGR * i] : 0.0;
#pragma unroll
            for (int j = 0; j < TH; j++) ac[j] = (cb + BGC * j < n) ? row[cb + BGC * j] : 0.0;
#pragma unroll
            for (int i = 0; i < THI; i++)
#pragma unroll
                for (int j = 0; j < TH; j++) kt[i][j] = fma(ar[i], ac[j], kt[i][j]);
        }
    }
    if (Pvals_g) {   // uniform
        const double *pv = Pvals_g + (size_t)inst * nnzP;
#pragma unroll
        for (int i = 0; i < THI; i++)
#pragma unroll
            for (int j = 0; j < TH; j++) {
                const int r = ra + BGR * i, cc = cb + BGC * j;
                if (r < n && cc < n) { const int ix = pmap[r * n + cc]; if (ix >= 0) kt[i][j] += pv[ix]; }
            }
    }
    CE_STAMP(8);
    // B / -B^T blocks and the right-hand side.  The source of equality e (a row of A or a_y of a boundary cone) is resolved
    // once per tile row / column into an LDS base pointer, so each entry costs one independent LDS read.
    {
        const double *prow_src[TI], *pcol_src[TJ];
        double rhs_eq[TI];
#pragma unroll
        for (int i = 0; i < TI; i++) {
            const int r = ra + BGR * i;
            prow_src[i] = A; rhs_eq[i] = 0.0;
            if (r >= n && r < NK) {
                const int src = esrc[r - n];
                prow_src[i] = src >= 0 ? A + src * lda : ay + (-1 - src) * n;
                rhs_eq[i] = src >= 0 ? dv[src] : cinfo[6 * (-1 - src) + 2];
            }
        }
#pragma unroll
        for (int j = 0; j < TJ; j++) {
            const int cc = cb + BGC * j;
            pcol_src[j] = A;
            if (cc >= n && cc < NK) { const int src = esrc[cc - n]; pcol_src[j] = src >= 0 ? A + src * lda : ay + (-1 - src) * n; }
        }
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++) {
                const int r = ra + BGR * i, cc = cb + BGC * j;
                if (r >= NK || cc > NK) continue;
                if (cc == NK) kt[i][j] = (r < n) ? dxg[(size_t)inst * n + r] + fvec[r] : rhs_eq[i];
                else if (r < n && cc >= n) kt[i][j] = -pcol_src[j][r];
                else if (r >= n && cc < n) kt[i][j] = prow_src[i][cc];
            }
    }
    CE_STAMP(4);
    // ---- pivot tolerance
    double ptol;
    {
        double r[1] = {0};
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++) r[0] = vmax_abs(r[0], (cb + BGC * j < NK) ? kt[i][j] : 0.0);      // (two alternating chains would not help: the block reduction behind it is what everybody waits for)
        block_reduce_n<1, NWB>(r, 1u, red);
        ptol = CE_RANK_TOL * (r[0] > 0 ? r[0] : 1.0);     // rank tolerance of the oracle's dense elimination (ce_common.h)
    }
    __syncthreads();                 // a_y / a_s are dead: the union region becomes colbuf / rowbuf
    for (int i = tid; i < 2 * BGR * TI; i += NTB) colbuf[i] = 0.0;
    // rank-deficient systems (redundant equality rows, degenerate active sets: the reference's LSQR returns a solution of the consistent
    // system there, diffcp_if.py:73-96): a column without an acceptable pivot is a FREE variable, set to zero and skipped; rows that never
    // serve as pivot keep colof = -1
    for (int i = tid; i < BGR * TI; i += NTB) colof[i] = -1;
    for (int j = tid; j < n; j += NTB) rx[j] = 0.0;
    for (int j = tid; j < NK - n; j += NTB) bv[j] = 0.0;
    __syncthreads();
    // ---- Gauss-Jordan with partial pivoting on the register tiles.  One workgroup barrier per pivot:
    //   (1) the 16 lanes owning column k (one DPP row) find the pivot with a DPP butterfly and publish the column
    //       (pivot entry zeroed, pivot value separately) -> barrier
    //   (2) the pivot row is broadcast INSIDE each 16-lane row with ds_bpermute (the lane with ra == prow % 16 holds exactly
    //       the entries K[prow][cb + 16 j] its row-mates need), so no second LDS round trip / barrier is required
    //   (3) rank-1 update of the live part of the tile (row slots >= ceil(NK/16) and column slots < jk are skipped)
#ifdef CE_TIMING
    bt0 = __builtin_readcyclecounter();
#endif
    unsigned rowdone = 0;     // bit i: row ra + 16 i has served as pivot
    unsigned nkmask = 0;      // bit i: row ra + BGR i < NK
#pragma unroll
    for (int i = 0; i < TI; i++) nkmask |= (ra + BGR * i < NK) ? (1u << i) : 0u;
    const int ridx0 = 255 - ra;            // search key of row ra + BGR i carries 255 - row in its eight lowest mantissa bits
    const int ilim = (NK + BGR - 1) / BGR;          // row slots in use
    const int lane_base = ((tid & 63) & ~(BGR - 1)) << 2;     // byte address of the first lane of this thread's row group for ds_bpermute
#pragma unroll
    for (int jk = 0; jk < TJ; jk++) {
        for (int ck = 0; ck < BGC; ck++) {
            const int k = BGC * jk + ck;
            if (k >= NK) break;
            const int buf = k & 1;
            double *cbuf = colbuf + buf * BGR * TI;
            if (cb == ck) {   // the 16 lanes owning column k: pivot search + publish the column
                // arg max |K[r][k]| over the rows not yet used, as ONE v_max_f64 per candidate: positive doubles order like their
                // bit patterns, so the row index rides in the 8 lowest mantissa bits (255 - r: ties go to the smallest row).
                // Branch-free (a used row contributes the key 0) and with the bare instruction: fmax() canonicalises both operands first
                // (two more v_max_f64 per step of a chain every other wave is waiting for).
                double best = 0.0;       // key 0: no candidate
                const unsigned alive = nkmask & ~rowdone;      // bit i: row ra + BGR i exists and has not served as pivot
#pragma unroll
                for (int i = 0; i < TI; i++) {
                    const int am = __builtin_amdgcn_sbfe((int)alive, i, 1);       // 0 or -1 (v_bfe_i32): a dead row's key is 0, with two ANDs instead of compares and selects
                    const double v = kt[i][jk];
                    const int lo = ((__double2loint(v) & ~0xFF) | (ridx0 - BGR * i)) & am;
                    best = vmax_abs(best, __hiloint2double(__double2hiint(v) & am, lo));      // (|.| is an operand modifier of the instruction: the sign bit rides along)
                }
                best = vmax_raw(best, dpp_mov<0xB1>(best));     // quad_perm [1,0,3,2]
                best = vmax_raw(best, dpp_mov<0x4E>(best));     // quad_perm [2,3,0,1]
                best = vmax_raw(best, dpp_mov<0x141>(best));    // row_half_mirror
                best = vmax_raw(best, dpp_mov<0x140>(best));    // row_mirror
                if constexpr (BGR == 32) best = vmax_raw(best, __shfl_xor(best, 16));      // the column's owners span two DPP rows
                const int bi = 255 - (__double2loint(best) & 0xFF);
                const bool tiny = best < ptol;               // no acceptable pivot in this column (best == 0: no candidate row left, bi = 255 is no row)
#pragma unroll
                for (int i = 0; i < TI; i++) cbuf[ra + BGR * i] = kt[i][jk];      // the whole column, unconditionally ...
                if (ra == (bi & (BGR - 1))) {                // the lane that holds row bi (bi = 255: some lane, value 0)
                    const int ib = bi / BGR;
                    double pv = 0.0;
#pragma unroll
                    for (int i = 0; i < TI; i++) pv = (ib == i) ? kt[i][jk] : pv;
                    if (bi < BGR * TI) cbuf[bi] = 0.0;       // ... then the pivot row's own entry is overwritten with 0 (same wave: LDS writes stay in program order)
                    pinfo[2 * buf] = tiny ? 0.0 : pv;
                    reinterpret_cast<int *>(pinfo + 2 * buf + 1)[0] = bi;
                }
                if (ra == 0 && tiny) misc[2] |= 4;
            }
            CE_BACC(0);      // pivot search + publish (the column's owners; everybody else arrives here at once)
            __syncthreads();
            CE_BACC(1);      // the barrier
            // the record and this thread's multipliers are requested together (the multipliers do not depend on the record): one LDS round trip
            // where the pivot row, then the pivot value, then the multipliers used to be three
            const double2 rec = *reinterpret_cast<const double2 *>(pinfo + 2 * buf);
            double cv[TI];
#pragma unroll
            for (int i = 0; i < TI; i++) cv[i] = cbuf[ra + BGR * i];
#pragma unroll
            for (int i = 0; i < TI; i++) asm volatile("" : "+v"(cv[i]));      // (pins the reads HERE: the optimiser otherwise sinks them below the free-variable branch, behind the record's round trip)
            const int prow = __builtin_amdgcn_readfirstlane(__double2loint(rec.y));
            const int ipv = prow / BGR;
            const double piv = rec.x;
            if (__builtin_amdgcn_readfirstlane(fabs(piv) < ptol ? 1 : 0)) continue;      // free variable (see above), no row is consumed.  readfirstlane: the value is the same in
                                                                                          // every lane, but only a scalar condition lets the compiler keep the loop body free of exec masking
            double pinv = __builtin_amdgcn_rcp(piv);           // hardware seed + two Newton steps (the IEEE divide expansion is
            pinv = fma(fma(-piv, pinv, 1.0), pinv, pinv);      // three times as long and sits on the critical path of every pivot)
            pinv = fma(fma(-piv, pinv, 1.0), pinv, pinv);
            if (ra == (prow & (BGR - 1))) {
                rowdone |= 1u << ipv;
                if (cb == 0) { colof[prow] = k; pivrow[prow] = piv; }
            }
            CE_BACC(2);      // record + multipliers read, reciprocal
            // pivot row: broadcast inside each 16-lane row
            double rw[TJ];
            const int src = lane_base + ((prow & (BGR - 1)) << 2);
#pragma unroll
            for (int i = 0; i < TI; i++) {
                if (ipv == i) {      // uniform
#pragma unroll
                    for (int j = jk; j < TJ; j++) {
                        const double v = kt[i][j];
                        const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
                        rw[j] = __hiloint2double(hi, lo);
                    }
                }
            }
            CE_BACC(3);      // pivot row by ds_bpermute
            if (cb <= ck) rw[jk] = 0.0;      // columns <= k of this slot are finished
#pragma unroll
            for (int i = 0; i < TI; i++) {
                if (i >= ilim) continue;     // uniform: pad row slots
                const double f = cv[i] * pinv;      // the pivot row's own entry was published as 0
#pragma unroll
                for (int j = jk; j < TJ; j++) kt[i][j] = fma(-f, rw[j], kt[i][j]);
            }
            CE_BACC(4);      // rank-1 update
        }
    }
    __syncthreads();
    CE_STAMP(5);
    // ---- solution: sol[colof[r]] = rhs[r] / pivot(r);  r_x -> rx, multipliers -> bv
#pragma unroll
    for (int j = 0; j < TJ; j++) {
        if (cb + BGC * j != NK) continue;
#pragma unroll
        for (int i = 0; i < TI; i++) {
            const int r = ra + BGR * i;
            if (r < NK) {
                const int k = colof[r];
                if (k < 0) continue;                           // row never served as pivot (rank-deficient system)
                const double sol = kt[i][j] / pivrow[r];
                if (k < n) rx[k] = sol; else bv[k - n] = sol;
            }
        }
    }
    __syncthreads();
    // ---- q = A r_x ; r_y
    mv_rows_partial(A, lda, m, n, rx, part);
    __syncthreads();
    for (int i = tid; i < m; i += NTB) qv2[i] = sum_parts(part, m, i);
    __syncthreads();
    for (int i = tid; i < z + T.l; i += NTB) vv[i] = (eqrow[i] >= 0) ? bv[eqrow[i]] : dv[i];
    for (int c = tid; c < nq; c += NTB) {
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
            const double cy = rhoy - il * (eyd - lam * eyq), cs = esd - il * (esd - lam * esq);
            const double k0 = (cy + cs) * M_SQRT1_2, kz = (cy - cs) * M_SQRT1_2;
            for (int i = r0 + 1; i < r1; i++) { const double zh = vv[i] * inz; vv[i] = il * (dv[i] - lam * qv2[i]) + kz * zh; }
            vv[r0] = il * (dv[r0] - lam * qv2[r0]) + k0;
        }
    }
    __syncthreads();
    if constexpr (PSD) {   // r~ in the rotated basis, then r_y,c = Q r~ = svec(U smat(r~) U^T)
        for (int t = T.soff[0] + tid; t < T.eoff + 3 * (T.nep + T.np); t += NTB) {
            const int rk = rkind[t];
            vv[t] = (rk == RK_EQ) ? bv[eqrow[t]] : (rk == RK_FREE ? dv[t] : (dv[t] - lamr[t] * qv2[t]) / (1 - lamr[t]));
        }
        __syncthreads();
        for (int c = 0; c < T.ns; c++) {
            const int k = T.sord[c], r0 = T.soff[c], d = k * (k + 1) / 2;
            const double *Um = psdU + c * T.maxs * T.maxs;
            double *X = psdScr, *W = X + T.maxs * T.maxs;
            for (int idx = tid; idx < k * k; idx += NTB) {
                const int i = idx / k, j = idx - i * k, a = i >= j ? i : j, b = i >= j ? j : i;
                const double v = vv[r0 + b * k - (b * (b - 1)) / 2 + (a - b)];
                X[idx] = (a == b) ? v : v * M_SQRT1_2;
            }
            __syncthreads();
            for (int idx = tid; idx < k * k; idx += NTB) {           // W = U X
                const int i = idx / k, j = idx - i * k;
                double acc = 0; for (int a = 0; a < k; a++) acc = fma(Um[i * k + a], X[a * k + j], acc);
                W[idx] = acc;
            }
            __syncthreads();
            for (int pos = tid; pos < d; pos += NTB) {               // T = W U^T
                int b = 0, rem = pos; while (rem >= k - b) { rem -= k - b; b++; }
                const int a = b + rem;
                double acc = 0; for (int e = 0; e < k; e++) acc = fma(W[a * k + e], Um[b * k + e], acc);
                vv[r0 + pos] = (a == b) ? acc : acc * M_SQRT2;
            }
            __syncthreads();
        }
        for (int c = tid; c < T.nep + T.np; c += NTB) {                     // r_y,c = W r~
            const int r0 = T.eoff + 3 * c;
            const double *W = expW + 9 * c;
            const double t0 = vv[r0], t1 = vv[r0 + 1], t2 = vv[r0 + 2];
#pragma unroll
            for (int a = 0; a < 3; a++) vv[r0 + a] = W[3 * a] * t0 + W[3 * a + 1] * t1 + W[3 * a + 2] * t2;
        }
        __syncthreads();
    }
    CE_STAMP(6);
    // ---- outputs in the boundary convention: dA_eval = [-dA.data, db[b_idx]], dq_eval = [dc, 0]   (diffcp_if.py:91-92)
    // x and y are staged in LDS (fvec and dv are dead): the gathers of the entry loop then are LDS reads behind ONE level of global loads (the
    // template's index pair), eight entries in flight per lane
    double *const xs = fvec, *const ys = dv;
    for (int j = tid; j < n; j += NTB) xs[j] = xg[(size_t)inst * n + j];
    for (int i = tid; i < m; i += NTB) ys[i] = yg[(size_t)inst * m + i];
    __syncthreads();
    {
        // eight entries per step in three fenced stages -- index pairs (global), operands (LDS, unconditional through a clamped column), stores: the plain loop
        // (even with `#pragma unroll 8`) compiled to one global round trip, a divergent branch on `j < n` around the LDS reads and a store PER ENTRY, twenty times
        // in series per lane (13 k cycles of a 220 k-cycle instance: profiles/r05/d_bwd_output_serialised.txt)
        constexpr int OU = 8;
        double *const dArow = dAo + (size_t)inst * T.nnz_aug;
        const int nnz = T.nnz_aug;
        for (int k0 = tid; k0 < nnz; k0 += OU * NTB) {
            int ii[OU], jj[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) { const int kk = min(k0 + u * NTB, nnz - 1); ii[u] = T.rowidx[kk]; jj[u] = T.colidx[kk]; }
            __builtin_amdgcn_sched_barrier(0);
            double xv[OU], rv[OU], vi[OU], yi[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) { const int jc = jj[u] < n ? jj[u] : 0; xv[u] = xs[jc]; rv[u] = rx[jc]; vi[u] = vv[ii[u]]; yi[u] = ys[ii[u]]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < OU; u++) {
                const double va = -(xv[u] * vi[u] - yi[u] * rv[u]);
                const double val = (jj[u] < n) ? va : -vi[u];
                if (k0 + u * NTB < nnz) dArow[k0 + u * NTB] = val;
            }
        }
    }
    for (int j = tid; j <= n; j += NTB) dqo[j * sdqk + inst * sdqb] = (j < n) ? -rx[j] : 0.0;
    if (dPo) {
        for (int k = tid; k < nnzP; k += NTB) {
            const int i = prow[k], j = pcol[k];
            const double v = -0.5 * (rx[i] * xs[j] + rx[j] * xs[i]);
            dPo[(size_t)inst * nnzP + k] = (p_tri && i != j) ? 2.0 * v : v;
        }
    }
    if (tid == 0 && adj_status) adj_status[inst] = misc[2];
    if (tid == 0 && fix && (misc[2] & 4)) fix[1 + atomicAdd(fix, 1)] = inst;      // rank-deficient system: diffcp's LSQR element replaces this basic solution (ce_vjp_qp)
#ifdef CE_TIMING
    CE_STAMP(7);
    if (tid < 7) dAo[(size_t)inst * T.nnz_aug + tid] = (double)(tstamp[tid + 1] - tstamp[tid]);
    if (tid == 8) dAo[(size_t)inst * T.nnz_aug + 8] = (double)(tstamp[8] - tstamp[3]);
    if (tid == 7) dAo[(size_t)inst * T.nnz_aug + 7] = (double)NK;
    if (tid == 9) for (int k = 0; k < 5; k++) dAo[(size_t)inst * T.nnz_aug + 9 + k] = (double)bacc[k];      // (wave 0's view)
    if (tid == 255) for (int k = 0; k < 5; k++) dAo[(size_t)inst * T.nnz_aug + 14 + k] = (double)bacc[k];     // (wave 3's view)
#endif
}
