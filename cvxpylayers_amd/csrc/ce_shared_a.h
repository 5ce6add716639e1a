// ce_shared_a.h -- SHARED-A adjoint: diffcp's adjoint solved by LSQR entirely inside ONE kernel, one workgroup per instance.
//
// For templates whose A does not depend on the parameters (only b, c vary: BASELINE configurations 4 and 5) the operator of diffcp's adjoint
// system  M^T r = dz,  r = (r_x, r_y, r_tau)  (oracle/cone_oracle.c apply_MT / apply_M / adjoint_one):
//        M^T r = ( -A^T r_y - c r_tau ,  DPi (A r_x - b r_tau - r_y) + r_y ,  c.r_x + b.r_y ),      dz = ( dx , DPi dy , -(x.dx + y.dy) )
// uses the SAME sparse matrix for every instance; only DPi = D Pi_K*(y - s), b and c are per instance.  The kernel runs LSQR on the FULL
// (n + m + 1) operator, as diffcp does: on rank-deficient systems (degenerate faces) LSQR returns the minimum-norm solution of THAT system, and
// pinning r_tau = 0 (rounds 1-4; still what a null q_vals selects) returns a different element -- equal gradients only where the system is regular.  Round 1 ran LSQR on it from Python (every
// operator application two rocBLAS GEMMs over the batch with the DENSE A plus ~30 small torch kernels: launch-bound, 250 ms for 1024
// instances of the 20 x 20 SDP).  Here every LSQR vector of an instance lives in LDS, A is applied from its sparse structure (CSR for A v,
// CSC for A^T v; values and indices stream from L2, shared by all workgroups), the convergence test is evaluated in the kernel, and the
// dense contractions of the PSD cone's derivative   DPi(V)[H] = U (B o (U^T H U)) U^T   run on the matrix cores (ce_psd_mfma.h).
// diffcp itself solves M^T r = dz with LSQR (its default mode); stopping rule and recurrences are Paige & Saunders', as in the oracle.
//
// Cones: zero / nonnegative / second-order / PSD / exponential / 3-d power.
#pragma once
#ifdef CE_TIMING   // debug build: shader cycles per phase of the LSQR iteration (thread 0, in registers), written over the first entries of the instance's dA row
#define LS_T(k) do { const long long t1_ = __builtin_readcyclecounter(); ls_tacc[k] += t1_ - ls_t0; ls_t0 = t1_; } while (0)
#else
#define LS_T(k) do { } while (0)
#endif
#include "ce_shared_a_ops.h"

struct SaStruct {            // sparse structure of the template's A part (device arrays, built once per engine)
    const int *csc_ptr;      // [n + 1]   column starts in the value order of the boundary (CSC of [A_cvx | b_cvx], first nnzA entries)
    const int *csc_row;      // [nnzA]
    const int *csr_ptr;      // [m + 1]
    const int *csr_col;      // [nnzA]
    const int *csr_src;      // [nnzA]    position of the entry in the value order
    int nnzA;
    const int *bpos;         // [m]       position of the row's b entry in the value order (-1: structurally zero)
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

// LDS doubles.  nvv: rows in front of the first PSD block (v = y - s is kept for those only; PSD blocks read y - s once, at the start).
// The partial sums of the dense-row products (2 NT doubles) share the PSD scratch matrices when the template has PSD blocks.
__host__ __device__ inline size_t sa_lsqr_lds_doubles(int n, int m, int nq, int ns, int maxs, int RP, int nvv, int ntri = 0, int lsmr = 0) {
    const int kp = ns > 0 ? psd_mfma_kp(maxs) : 0;
    return (size_t)(RP > 0 ? 2 * RP + (ns > 0 ? 0 : 2 * NT) : 0) + (size_t)(ns > 0 ? (2 * ns + 2) * kp * (kp + 1) + 2 * kp + 8 : 0) + NW * 8 +
           (size_t)(nvv + (nvv & 1)) + 6 * (size_t)m + 4 * (size_t)n + 5 * (size_t)(nq > 0 ? nq : 1) + 16 + 9 * (size_t)ntri + (ntri & 1) + (lsmr ? (size_t)m + n : 0);      // (LSMR: one more vector, h-bar)
}

// RP > 0: A is applied through its split into singleton rows and r <= RP dense rows (ce_shared_a_ops.h: balanced, wide loads); RP == 0: through
// the CSR / CSC structure (any sparsity pattern, but rows of very different lengths serialise on the longest).
// HPSD / HTRI: the template has PSD blocks / exponential-power triples (false: their code is compiled out -- the plain-cone instantiation carried 50 spilled VGPRs of it)
// LSMR: the same Golub-Kahan bidiagonalisation driven by Fong & Saunders' LSMR recurrences and stopping tests (diffcp's mode="lsmr"; oracle/cone_oracle.c lsmr_core,
// pinned on scipy.sparse.linalg.lsmr): every product, cone derivative and reduction below is shared, the solution update needs one more vector (h-bar) and |x|.
template <int RP, bool HPSD = true, bool HTRI = true, bool LSMR = false>
__global__ void __launch_bounds__(NT, 3)
k_sa_lsqr(DevT T, SaStruct S, SaSplit F, const double *__restrict__ Avals0, long sAb, int per_inst, const double *__restrict__ qg, long sqk, long sqb, const double *__restrict__ xg, const double *__restrict__ yg,
          const double *__restrict__ sg, const double *__restrict__ dxg, const double *__restrict__ dyg, double *__restrict__ dAo,
          double *__restrict__ dqo, long sdqk, long sdqb, int *__restrict__ adj_status, int *__restrict__ iters_o, double atol, double btol, double conlim, int itn_lim,
          const int *__restrict__ sel = nullptr, int status_or = 0, int a_lds = 0, int *__restrict__ sel_reset = nullptr) {
    // sel != nullptr (ce_vjp's re-solve of the instances its direct elimination flagged rank-deficient): sel[0] instances are listed in sel[1 ...] (appended by
    // the elimination kernel on the same stream); the grid is a fixed number of workgroups that walk the list -- the host never learns the count.  status_or is
    // OR-ed into the adj_status of every instance served (ce_vjp: 4 | 8 = "rank-deficient, re-solved by LSQR").
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    // ce_vjp alternates TWO lists: while this launch walks `sel`, it empties the other one (walked by the previous call's launch, appended to by the next call's
    // elimination kernel) -- no memset launch per call, and no exit ticket (768 atomics on one address cost 7 us: measured)
    if (sel_reset && blockIdx.x == 0 && tid == 0) *sel_reset = 0;
    const int n = T.n, m = T.m, z = T.z, nl = T.l, nq = T.nq, ns = HPSD ? T.ns : 0;
    const int KP = ns > 0 ? psd_mfma_kp(T.maxs) : 0, P = KP + 1, PM = KP * P;
    const float rKP = KP > 0 ? 1.0f / (float)KP : 1.0f;
    const int ntri = HTRI ? T.nep + T.np : 0;
    const int psd_first = ns > 0 ? T.soff[0] : T.eoff, nvv = psd_first + (psd_first & 1);      // rows in front of the PSD blocks / the triples
    double *p = sm;                                          // (everything read with 16-byte accesses sits at the start: even sizes only)
    double *wyd = p, *vd = p, *part = p;
    if constexpr (RP > 0) { wyd = p; p += RP; vd = p; p += RP; if (ns == 0) { part = p; p += 2 * NT; } }
    double *Hm = p; p += PM; double *Ym = p; p += PM;        // PSD scratch
    if (RP > 0 && ns > 0) part = Hm;                         // partial sums of the dense-row products: 2 NT <= 2 PM doubles, used between dproj calls only
    double *Um = p; p += (size_t)ns * PM;                    // eigenvectors of smat(v_c), per cone
    double *Bm = p; p += (size_t)ns * PM;                    // divided differences, per cone
    double *cs = p; p += (ns > 0 ? 2 * KP + 8 : 0);
    double *red = p; p += NW * 8;
    double *vv = p; p += nvv;          // v = y - s, rows in front of the PSD blocks
    double *uy = p; p += m; double *vy = p; p += m; double *wy = p; p += m; double *ry = p; p += m; double *ty = p; p += m; double *qv = p; p += m;
    double *ux = p; p += n; double *vx = p; p += n; double *wx = p; p += n; double *rx = p; p += n;
    double *hby = p; if (LSMR) p += m; double *hbx = p; if (LSMR) p += n;          // LSMR: h-bar (w plays h, r plays x)
    double *socs = p; p += 5 * (nq > 0 ? nq : 1);          // per cone: t, |z|, case, z.h ; then h_0 per cone
    p += (size_t)(p - sm) & 1;
    double *Jt = p; p += 9 * (size_t)ntri;                    // exponential / power triples: symmetrised 3 x 3 derivative of the dual-cone projection
    // a_lds (per-instance A only; the host grants it when m n doubles fit behind the vectors): the instance's A is staged DENSE in LDS once and both products of
    // every LSQR iteration read it there -- the CSR / CSC products stream the instance's 8 (nnzA) bytes from L2 / HBM twice per iteration, which is what an
    // instance of the re-solve list costs when it runs alone on its CU (302 iterations x ~35 k cycles at the metric shape).
    p += (size_t)(p - sm) & 1;
    double *Ad = p;                                          // [m][n], boundary sign (the products negate)
  for (int li = blockIdx.x;; li += gridDim.x) {              // (one pass without a list: instance = workgroup)
    int inst = li;
    if (sel) { if (li >= sel[0]) break; inst = sel[1 + li]; } else if (li != (int)blockIdx.x) break;
    const double *x = xg + (size_t)inst * n, *y = yg + (size_t)inst * m, *s = sg + (size_t)inst * m;
    // the tau row / column of the operator: c_j from the boundary's q values, b_i from this instance's value row (both stay in global memory: L2-resident, read
    // with the loads of the products they join; LDS has no room for two more vectors at three workgroups per CU).  qg == null: r_tau pinned to 0.
    const bool TAU = qg != nullptr;
    const double *cq = TAU ? qg + (size_t)inst * sqb : nullptr;
    const double *Ab = Avals0 + (size_t)inst * sAb;
    // per_inst (RP == 0 only): the A part differs between instances too -- the products read this instance's value row (diffcp's LSQR adjoint for per-instance
    // templates, solver_args mode="lsqr": ce_vjp_lsqr); else instance 0's values serve every workgroup and stay L2-resident
    const double *Aprod = per_inst ? Ab : Avals0;
    auto bval = [&](int i) -> double { if (!TAU) return 0.0; const int pb = S.bpos[i]; return pb >= 0 ? Ab[pb] : 0.0; };
    if (a_lds) {
        for (int i = tid; i < m * n; i += NT) Ad[i] = 0.0;
        __syncthreads();
        for (int k = tid; k < S.nnzA; k += NT) Ad[T.rowidx[k] * n + T.colidx[k]] = Aprod[k];
    }

    for (int i = tid; i < psd_first; i += NT) vv[i] = y[i] - s[i];
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
    if constexpr (HTRI) for (int c = tid; c < ntri; c += NT) {   // D Pi_K*(v) of every triple (k_ca_triple_jac; symmetrised like the batched path does)
        const int e0 = T.eoff + 3 * c;
        const double v3[3] = {y[e0] - s[e0], y[e0 + 1] - s[e0 + 1], y[e0 + 2] - s[e0 + 2]};
        double w3[3] = {-v3[0], -v3[1], -v3[2]}, J[9];
        if (c < T.nep) { exp_dproject(w3, J); for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i]; }
        else {
            const double a = T.pw[c - T.nep];
            if (a < 0) pow_dproject(v3, -a, J);
            else { pow_dproject(w3, a, J); for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i]; }
        }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jt[9 * c + 3 * i + j] = 0.5 * (J[3 * i + j] + J[3 * j + i]);
    }
    if constexpr (HPSD) for (int c = 0; c < ns; c++) {      // eigenvectors and divided differences of every PSD block (cold Jacobi, once)
        const int k = T.sord[c];
        double *U = Um + (size_t)c * PM, *Bc = Bm + (size_t)c * PM;
        const double *ys = y + T.soff[c], *ss = s + T.soff[c];
        for (int idx = tid; idx < KP * KP; idx += NT) {
            const int i = psd_fdiv(idx, rKP), j = idx - i * KP;
            double sv = 0.0;
            if (i < k && j < k) { const int a = i >= j ? i : j, b = i >= j ? j : i; const int e = b * k - (b * (b - 1)) / 2 + (a - b); const double v0 = ys[e] - ss[e]; sv = (a == b) ? v0 : v0 * M_SQRT1_2; }
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

    // the producers of the passes' y-side operand (v_y, q) also drop the entries of the dense rows into wyd[slot] (zero padded once): every thread calls it for the rows it writes
    auto to_wyd = [&](int i, double v) { if constexpr (RP > 0) { const int a = F.rowslot[i]; if (a >= 0) wyd[a] = v; } };
    if constexpr (RP > 0) { for (int a = tid; a < RP; a += NT) wyd[a] = 0.0; }
    // sink(i, DPi(hs * h)_i) for every row i.  h must be complete (synchronised) on entry; every thread calls it.  The value of row i depends on
    // h_i, on per-cone sums taken in a first phase (one barrier) and, for PSD blocks, on the whole block (copied to a matrix before any sink of
    // the block runs): sinks may therefore overwrite h_i.  No trailing barrier.
    auto dproj = [&](const double *h, double hs, auto &&sink) {
        if (nq > 0) {
            for (int c = tid >> 6; c < nq; c += NW) {       // z.h and h_0 per cone: one wave per cone
                double a = 0;
                for (int k = T.qoff[c] + 1 + (tid & 63); k < T.qoff[c + 1]; k += 64) a = fma(vv[k], h[k], a);
                a = wave_reduce_dpp<false>(a);
                if ((tid & 63) == 0) { socs[4 * c + 3] = a * hs; socs[4 * nq + c] = h[T.qoff[c]] * hs; }
            }
            __syncthreads();
        }
        for (int i = tid; i < psd_first; i += NT) {
            double o = h[i] * hs;
            if (i >= z && i < z + nl) o = (vv[i] > 0) ? o : 0.0;
            else {
                const int c = (i >= z + nl && nq > 0) ? T.rowcone[i] : -1;
                if (c >= 0) {
                    const double kase = socs[4 * c + 2];
                    if (kase == 1.0) o = 0.0;
                    else if (kase == 2.0) {
                        const int r0 = T.qoff[c];
                        const double t = socs[4 * c], nz = socs[4 * c + 1], zh = socs[4 * c + 3], h0 = socs[4 * nq + c];
                        const double nzs = fmax(nz, 1e-300);
                        if (i == r0) o = (nz * h0 + zh) / (2 * nzs);
                        else o = (vv[i] * h0 + (t + nz) * o - t * vv[i] * zh / (nzs * nzs)) / (2 * nzs);
                    }
                }
            }
            sink(i, o);
        }
        if constexpr (HPSD) for (int c = 0; c < ns; c++) {      // PSD block: Z = U (B o (U^T H U)) U^T on the matrix cores
            const int k = T.sord[c], off = T.soff[c], KT = KP / 16;
            const double *U = Um + (size_t)c * PM, *Bc = Bm + (size_t)c * PM;
            for (int idx = tid; idx < KP * KP; idx += NT) {
                const int i = psd_fdiv(idx, rKP), j = idx - i * KP;
                double sv = 0.0;
                if (i < k && j < k) { const int a = i >= j ? i : j, b = i >= j ? j : i; const double v0 = h[off + b * k - (b * (b - 1)) / 2 + (a - b)] * hs; sv = (a == b) ? v0 : v0 * M_SQRT1_2; }
                Hm[i * P + j] = sv;
            }
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return Hm[K * P + M]; }, [&](int K, int N) { return U[K * P + N]; }, [&](int M, int N, double v) { Ym[M * P + N] = v; }, (k + 3) >> 2);   // H U
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return U[K * P + M]; }, [&](int K, int N) { return Ym[K * P + N]; }, [&](int M, int N, double v) { Hm[M * P + N] = v * Bc[M * P + N]; }, (k + 3) >> 2);   // B o (U^T H U)
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return U[M * P + K]; }, [&](int K, int N) { return Hm[K * P + N]; }, [&](int M, int N, double v) { Ym[M * P + N] = v; }, (k + 3) >> 2);   // U Y
            __syncthreads();
            psd_mfma_gemm<NT>(KT, [&](int M, int K) { return Ym[M * P + K]; }, [&](int K, int N) { return U[N * P + K]; }, [&](int M, int N, double v) { Hm[M * P + N] = v; }, (k + 3) >> 2);   // (U Y) U^T
            __syncthreads();
            for (int idx = tid; idx < k * k; idx += NT) {          // lower triangle (a >= b) -> svec position
                const int a = psd_fdiv(idx, 1.0f / (float)k), b = idx - a * k;
                if (a < b) continue;
                const double v0 = 0.5 * (Hm[a * P + b] + Hm[b * P + a]);
                sink(off + b * k - (b * (b - 1)) / 2 + (a - b), (a == b) ? v0 : v0 * M_SQRT2);
            }
            if (c + 1 < ns) __syncthreads();                       // (Hm is reused by the next block)
        }
        if constexpr (HTRI) for (int c = tid; c < ntri; c += NT) {     // triples: o = J h (the thread reads its three entries before it sinks them)
            const int e0 = T.eoff + 3 * c;
            const double h0 = h[e0] * hs, h1 = h[e0 + 1] * hs, h2 = h[e0 + 2] * hs;
            const double *J = Jt + 9 * c;
            sink(e0, J[0] * h0 + J[1] * h1 + J[2] * h2); sink(e0 + 1, J[3] * h0 + J[4] * h1 + J[5] * h2); sink(e0 + 2, J[6] * h0 + J[7] * h1 + J[8] * h2);
        }
    };
    // the two products of one operator application share their barriers.  On return (synchronised): fx(j, (A^T yin)_j) was called for every
    // column and fy(i, (A xin)_i) for every row (solver-form A = -A_cvx: the stored values carry the boundary's sign).
    auto both_products = [&](const double *yin, const double *xin, auto &&fx, auto &&fy) {
        if constexpr (RP > 0) {
            // (wyd = the dense rows' entries of yin: written by whoever produced yin -- to_wyd below -- so that the pass starts without a gather phase and its barrier)
            sa_fused_pass<NT, RP>(F.AdT, n, wyd, xin,
                                  [&](int j, int k8) { double acc = 0; for (int k = F.scol_ptr[j] + k8; k < F.scol_ptr[j + 1]; k += 8) { const int i = F.scol_row[k]; acc = fma(F.srow_val[i], yin[i], acc); } return acc; },
                                  fx, part, F.sing_i, F.sing_v, yin, cq, sqk);
            __syncthreads();
            for (int i0 = tid; i0 < m; i0 += 4 * NT) {          // four rows per step: their index / value loads (global memory) are requested together
                int cc[4], aa[4], pb[4]; double sv4[4], bb[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = min(i0 + u * NT, m - 1); cc[u] = F.srow_col[i]; aa[u] = F.rowslot[i]; sv4[u] = F.srow_val[i]; pb[u] = TAU ? S.bpos[i] : -1; }
#pragma unroll
                for (int u = 0; u < 4; u++) bb[u] = pb[u] >= 0 ? Ab[pb[u]] : 0.0;          // (second level of the chain, the four of them together)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * NT;
                    if (i >= m) break;
                    if (cc[u] >= 0) fy(i, sv4[u] * xin[cc[u]], bb[u]);
                    else {
                        double s_ = 0;                               // slot of a dense row, -1: empty row
                        if (aa[u] >= 0) {
#pragma unroll
                            for (int w = 0; w < NW; w++) s_ += part[w * RP + aa[u]];
                        }
                        fy(i, s_, bb[u]);
                    }
                }
            }
            __syncthreads();
        } else if (a_lds) {
            // dense products from the LDS copy: eight lanes per output, four entries per lane and step in flight, DPP butterfly over the eight
            const int g = tid >> 3, c8 = tid & 7;
            for (int j0 = 0; j0 < n; j0 += NT / 8) {          // (A^T yin)_j
                const int j = j0 + g, jc = j < n ? j : n - 1;
                double a0 = 0, a1 = 0;
                for (int i = c8; i < m; i += 32) {
                    double av[4], yv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int iu = min(i + 8 * u, m - 1); av[u] = Ad[iu * n + jc]; yv[u] = yin[iu]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) { const double w = (i + 8 * u < m) ? yv[u] : 0.0; if (u & 1) a1 = fma(av[u], w, a1); else a0 = fma(av[u], w, a0); }
                }
                const double a = group_reduce<8, false>(a0 + a1);
                if (j < n && c8 == 0) fx(j, -a, TAU ? cq[(size_t)j * sqk] : 0.0);
            }
            for (int i0 = 0; i0 < m; i0 += NT / 8) {          // (A xin)_i
                const int i = i0 + g, ic = i < m ? i : m - 1;
                const double *row = Ad + ic * n;
                double a0 = 0, a1 = 0;
                for (int j = c8; j < n; j += 32) {
                    double av[4], xv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int ju = min(j + 8 * u, n - 1); av[u] = row[ju]; xv[u] = xin[ju]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) { const double w = (j + 8 * u < n) ? xv[u] : 0.0; if (u & 1) a1 = fma(av[u], w, a1); else a0 = fma(av[u], w, a0); }
                }
                const double a = group_reduce<8, false>(a0 + a1);
                if (i < m && c8 == 0) fy(i, -a, bval(i));
            }
            __syncthreads();
        } else {
            sa_spmv(S.csc_ptr, S.csc_row, (const int *)nullptr, Aprod, n, [&](int i) { return yin[i]; }, [&](int j, double a) { fx(j, -a, TAU ? cq[(size_t)j * sqk] : 0.0); });
            sa_spmv(S.csr_ptr, S.csr_col, S.csr_src, Aprod, m, [&](int j) { return xin[j]; }, [&](int i, double a) { fy(i, -a, bval(i)); });
            __syncthreads();
        }
    };
    auto safe = [](double t) -> double { return t > 0 ? t : 1.0; };
    // (sum of squares, tau-row dot product) over the workgroup in one reduction
    // (the sums come back workgroup-uniform: through readfirstlane they and every scalar of the recurrences computed from them are known to be uniform and live in
    // scalar registers across the phases -- ~20 loop-carried doubles that otherwise hold 40 VGPRs of a kernel that sits at its 168-VGPR ceiling)
    // One barrier per reduction: the two call sites of the iteration own separate halves of `red` (NW * 8 doubles), each written again an iteration (many barriers) later;
    // every phase that follows a reduction synchronises before it reads another thread's vector entries.
    auto sum_two = [&](double v, double w, double &wsum) -> double { double r[2] = {v, w}; block_reduce<2, false>(r, 0u, red); wsum = uniform_d(r[1]); return uniform_d(r[0]); };
    auto sum_three = [&](double v, double w, double u3, double &wsum, double &usum) -> double { double r[3] = {v, w, u3}; block_reduce<3, false>(r, 0u, red + NW * 4); wsum = uniform_d(r[1]); usum = uniform_d(r[2]); return uniform_d(r[0]); };

    // ---- LSQR (Paige & Saunders) on  N r = dz,  N = M^T  (the tau components ut, vt, wt, rt are workgroup-uniform scalars in registers)
    //      N   (r_x, r_y, r_t) = ( -A^T r_y - c r_t ,  DPi(A r_x - b r_t - r_y) + r_y ,  c.r_x + b.r_y )
    //      N^T (p_x, p_y, p_t) = (  A^T q + c p_t   , -A p_x + b p_t - q + p_y          , -c.p_x - b.q   ),     q = DPi(p_y)
    double acc = 0, acct = 0, dsum = 0;
    for (int j = tid; j < n; j += NT) { const double v = dxg[(size_t)inst * n + j]; ux[j] = v; rx[j] = 0.0; if constexpr (LSMR) hbx[j] = 0.0; acc = fma(v, v, acc); acct = fma(x[j], v, acct); }
    for (int i = tid; i < m; i += NT) { const double v = dyg[(size_t)inst * m + i]; ty[i] = v; ry[i] = 0.0; if constexpr (LSMR) hby[i] = 0.0; acct = fma(y[i], v, acct); }
    __syncthreads();
    dproj(ty, 1.0, [&](int i, double o) { uy[i] = o; acc = fma(o, o, acc); });
    acc = sum_two(acc, acct, dsum);
    double ut = TAU ? -dsum : 0.0, vt = 0.0, wt = 0.0, rt = 0.0;          // dz_tau = -(x.dx + y.dy)
    const double bnorm = sqrt(fma(ut, ut, acc));
    double beta = bnorm, ib = 1.0 / safe(beta);
    // u <- u / beta ;  q = DPi(uy) ;  v = N^T u
    for (int j = tid; j < n; j += NT) ux[j] *= ib;
    dproj(uy, ib, [&](int i, double o) { qv[i] = o; to_wyd(i, o); });
    if (ntri > 0) __syncthreads();                       // a triple's thread reads three rows of uy that other threads rescale below
    for (int i = tid; i < m; i += NT) uy[i] *= ib;       // (every other read of uy by dproj is behind one of its barriers, or by the row's own thread)
    ut *= ib;
    __syncthreads();
    acc = 0; acct = 0;
    both_products(qv, ux, [&](int j, double a, double cj) { const double v = fma(cj, ut, a); vx[j] = v; acc = fma(v, v, acc); acct = fma(cj, ux[j], acct); },
                  [&](int i, double a, double bi) { const double v = fma(bi, ut, -a - qv[i] + uy[i]); vy[i] = v; acc = fma(v, v, acc); acct = fma(bi, qv[i], acct); });
    acc = sum_two(acc, acct, dsum);
    vt = TAU ? -dsum : 0.0;
    double alfa = sqrt(fma(vt, vt, acc));
    // |w|^2 of the current search direction, as per-thread partial sums: LSQR's estimate of cond(N) (the `conlim` stopping test) needs sum_k |w_k|^2 / rho_k^2;
    // the partial sums ride on the first reduction of the NEXT iteration (no barrier of their own)
    double wsq = 0, ddnorm = 0;
    const double ctol = conlim > 0 ? 1.0 / conlim : 0.0;
    {
        const double ia = 1.0 / safe(alfa);
        for (int j = tid; j < n; j += NT) { const double v = vx[j] * ia; vx[j] = v; wx[j] = v; wsq = fma(v, v, wsq); }
        for (int i = tid; i < m; i += NT) { const double v = vy[i] * ia; vy[i] = v; wy[i] = v; wsq = fma(v, v, wsq); to_wyd(i, v); }
        vt *= ia; wt = vt;
    }
    __syncthreads();
    double rhobar = alfa, phibar = beta, anorm = 0, xxnorm = 0, zz = 0, cs2 = -1, sn2 = 0;
    // LSMR's scalars (scipy.sparse.linalg.lsmr's names; damp = 0)
    double zetabar = alfa * beta, alphabar = alfa, mrho = 1, mrhobar = 1, cbar = 1, sbar = 0, hbt = 0;
    double betadd = beta, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, normA2 = alfa * alfa, maxrbar = 0, minrbar = 1e100;
    auto sym_ortho = [](double a, double b, double &c, double &s_, double &r_) {
        if (b == 0) { c = a == 0 ? 1.0 : (a > 0 ? 1.0 : -1.0); s_ = 0; r_ = fabs(a); }
        else if (a == 0) { c = 0; s_ = b > 0 ? 1.0 : -1.0; r_ = fabs(b); }
        else if (fabs(b) > fabs(a)) { const double tau = a / b; s_ = (b > 0 ? 1.0 : -1.0) / sqrt(1 + tau * tau); c = s_ * tau; r_ = b / s_; }
        else { const double tau = b / a; c = (a > 0 ? 1.0 : -1.0) / sqrt(1 + tau * tau); s_ = c * tau; r_ = a / c; }
    };
    bool live = bnorm > 0 && alfa * beta > 0;
    int itn = 0;
#ifdef CE_TIMING
    long long ls_tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ls_t0 = __builtin_readcyclecounter();
#endif
    while (live && itn < itn_lim) {
        itn++;
        LS_T(7);
        // t = N v :  tx = -A^T vy - c vt ;  ty = DPi(A vx - b vt - vy) + vy ;  tt = c.vx + b.vy ;   u-hat = t - alfa u
        acc = 0; acct = 0;
        both_products(vy, vx, [&](int j, double a, double cj) { const double v = -a - cj * vt - alfa * ux[j]; ux[j] = v; acc = fma(v, v, acc); acct = fma(cj, vx[j], acct); },
                      [&](int i, double a, double bi) { const double vyi = vy[i]; ty[i] = a - bi * vt - vyi; acct = fma(bi, vyi, acct); });
        LS_T(0);
        dproj(ty, 1.0, [&](int i, double o) { const double v = o + vy[i] - alfa * uy[i]; uy[i] = v; acc = fma(v, v, acc); });
        LS_T(1);
        double wsum = 0;
        acc = sum_three(acc, acct, wsq, dsum, wsum);
        wsum = fma(wt, wt, wsum);                          // |w_{k-1}|^2, tau component included
        ut = TAU ? dsum - alfa * ut : 0.0;
        beta = sqrt(fma(ut, ut, acc));
        LS_T(2);
        ib = 1.0 / safe(beta);
        anorm = sqrt(anorm * anorm + alfa * alfa + beta * beta);
        // u = u-hat / beta ;  q = DPi(uy) ;  (tx, ty) = N^T u ;  v-hat = t - beta v
        for (int j = tid; j < n; j += NT) ux[j] *= ib;
        dproj(uy, ib, [&](int i, double o) { qv[i] = o; to_wyd(i, o); });
        if (ntri > 0) __syncthreads();
        for (int i = tid; i < m; i += NT) uy[i] *= ib;
        ut *= ib;
        __syncthreads();
        LS_T(3);
        acc = 0; acct = 0;
        both_products(qv, ux, [&](int j, double a, double cj) { const double v = fma(cj, ut, a) - beta * vx[j]; vx[j] = v; acc = fma(v, v, acc); acct = fma(cj, ux[j], acct); },
                      [&](int i, double a, double bi) { const double qi = qv[i]; const double v = fma(bi, ut, -a - qi + uy[i]) - beta * vy[i]; vy[i] = v; acc = fma(v, v, acc); acct = fma(bi, qi, acct); });
        LS_T(4);
        acc = sum_two(acc, acct, dsum);
        vt = TAU ? -dsum - beta * vt : 0.0;
        alfa = sqrt(fma(vt, vt, acc));
        LS_T(5);
        if constexpr (LSMR) {
            double chat, shat, alphahat; sym_ortho(alphabar, 0.0, chat, shat, alphahat);
            const double rhoold = mrho; double c_, s_; sym_ortho(alphahat, beta, c_, s_, mrho);
            const double thetanew = s_ * alfa; alphabar = c_ * alfa;
            const double rhobarold = mrhobar, zetaold = zeta, thetabar = sbar * mrho, rhotemp = cbar * mrho;
            { double cb, sb, rb; sym_ortho(cbar * mrho, thetanew, cb, sb, rb); cbar = cb; sbar = sb; mrhobar = rb; }
            zeta = cbar * zetabar; zetabar = -sbar * zetabar;
            const double f1 = -(thetabar * mrho / (rhoold * rhobarold)), f2 = zeta / (mrho * mrhobar), f3 = -(thetanew / mrho), ia = 1.0 / safe(alfa);
            double xx[1] = {0};
            for (int j = tid; j < n; j += NT) { const double v = vx[j] * ia, hh = wx[j], hb = fma(hbx[j], f1, hh), xv = fma(f2, hb, rx[j]); vx[j] = v; hbx[j] = hb; rx[j] = xv; wx[j] = fma(hh, f3, v); xx[0] = fma(xv, xv, xx[0]); }
            for (int i = tid; i < m; i += NT) { const double v = vy[i] * ia, hh = wy[i], hb = fma(hby[i], f1, hh), xv = fma(f2, hb, ry[i]); vy[i] = v; hby[i] = hb; ry[i] = xv; wy[i] = fma(hh, f3, v); xx[0] = fma(xv, xv, xx[0]); to_wyd(i, v); }
            { vt *= ia; hbt = fma(hbt, f1, wt); rt = fma(f2, hbt, rt); wt = fma(wt, f3, vt); }
            block_reduce<1>(xx, 0u, red + NW * 7);          // |x| (its own four doubles of `red`; ends synchronised: the vectors above are complete for the next products)
            const double normx = sqrt(fma(rt, rt, xx[0]));
            const double betaacute = chat * betadd, betacheck = -shat * betadd;
            const double betahat = c_ * betaacute; betadd = -s_ * betaacute;
            const double thetatildeold = thetatilde; double ctildeold, stildeold, rhotildeold; sym_ortho(rhodold, thetabar, ctildeold, stildeold, rhotildeold);
            thetatilde = stildeold * mrhobar; rhodold = ctildeold * mrhobar; betad = -stildeold * betad + ctildeold * betahat;
            tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
            const double taud = (zeta - thetatilde * tautildeold) / rhodold;
            ddnorm += betacheck * betacheck;          // (scipy's d)
            const double normr = sqrt(ddnorm + (betad - taud) * (betad - taud) + betadd * betadd);
            normA2 += beta * beta; const double normA = sqrt(normA2); normA2 += alfa * alfa;
            maxrbar = fmax(maxrbar, rhobarold);
            if (itn > 1) minrbar = fmin(minrbar, rhobarold);
            const double condA = fmax(maxrbar, rhotemp) / fmin(minrbar, rhotemp);
            const double normar = fabs(zetabar);
            const double test1 = normr / safe(bnorm), test2 = (normA * normr) != 0 ? normar / (normA * normr) : 1e300, test3 = 1.0 / condA;
            const double tt1 = test1 / (1.0 + normA * normx / safe(bnorm)), rtol = btol + atol * normA * normx / safe(bnorm);
            if (test1 <= rtol || test2 <= atol || test3 <= ctol || 1.0 + test3 <= 1.0 || 1.0 + test2 <= 1.0 || 1.0 + tt1 <= 1.0) live = false;
            zetabar = uniform_d(zetabar); alphabar = uniform_d(alphabar); mrho = uniform_d(mrho); mrhobar = uniform_d(mrhobar); cbar = uniform_d(cbar); sbar = uniform_d(sbar); hbt = uniform_d(hbt);
            betadd = uniform_d(betadd); betad = uniform_d(betad); rhodold = uniform_d(rhodold); tautildeold = uniform_d(tautildeold); thetatilde = uniform_d(thetatilde); zeta = uniform_d(zeta);
            normA2 = uniform_d(normA2); maxrbar = uniform_d(maxrbar); minrbar = uniform_d(minrbar);
        } else {
        const double rho = sqrt(rhobar * rhobar + beta * beta);
        const double cs_ = rhobar / safe(rho), sn = beta / safe(rho);
        const double theta = sn * alfa; rhobar = -cs_ * alfa; const double phi = cs_ * phibar; phibar = sn * phibar; const double tau = sn * phi;
        const double t1 = phi / safe(rho), t2 = -theta / safe(rho), ia = 1.0 / safe(alfa);
        wsq = 0;
        for (int j = tid; j < n; j += NT) { const double v = vx[j] * ia, w = wx[j], wn = v + t2 * w; vx[j] = v; rx[j] += t1 * w; wx[j] = wn; wsq = fma(wn, wn, wsq); }
        for (int i = tid; i < m; i += NT) { const double v = vy[i] * ia, w = wy[i], wn = v + t2 * w; vy[i] = v; ry[i] += t1 * w; wy[i] = wn; wsq = fma(wn, wn, wsq); to_wyd(i, v); }
        ddnorm += wsum / (safe(rho) * safe(rho));
        { vt *= ia; rt = fma(t1, wt, rt); wt = fma(t2, wt, vt); }
        __syncthreads();
        LS_T(6);
        const double delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * zz, zbar = rhs / safe(fabs(gambar)) * (gambar > 0 ? 1.0 : (gambar < 0 ? -1.0 : 0.0));
        const double xnorm = sqrt(xxnorm + zbar * zbar);
        const double gamma = sqrt(gambar * gambar + theta * theta);
        cs2 = gambar / safe(gamma); sn2 = theta / safe(gamma); zz = rhs / safe(gamma); xxnorm += zz * zz;
        const double rnorm = phibar, arnorm = alfa * fabs(tau);
        const double test1 = rnorm / safe(bnorm), test2 = arnorm / (anorm * rnorm + 1e-300);
        const double rtol = btol + atol * anorm * xnorm / safe(bnorm);
        // the remaining stopping tests of Paige & Saunders' LSQR as diffcp / the oracle run them (oracle/cone_oracle.c lsqr_MT): the condition estimate against
        // conlim (1e8: ill-conditioned systems stop HERE, long before atol / btol are met) and the three machine-precision tests
        const double test3 = 1.0 / (anorm * sqrt(ddnorm) + 1e-300), tt1 = test1 / (1.0 + anorm * xnorm / safe(bnorm));
        if (test1 <= rtol || test2 <= atol || test3 <= ctol || 1.0 + test3 <= 1.0 || 1.0 + test2 <= 1.0 || 1.0 + tt1 <= 1.0) live = false;
        }
        alfa = uniform_d(alfa); beta = uniform_d(beta); ut = uniform_d(ut); vt = uniform_d(vt); wt = uniform_d(wt); rt = uniform_d(rt);
        rhobar = uniform_d(rhobar); phibar = uniform_d(phibar); anorm = uniform_d(anorm); xxnorm = uniform_d(xxnorm); zz = uniform_d(zz);
        cs2 = uniform_d(cs2); sn2 = uniform_d(sn2); ddnorm = uniform_d(ddnorm);
    }
    __syncthreads();
    // ---- outputs in the boundary convention: dA_eval = [-dA.data, db[b_idx]], dq_eval = [dc, 0]  (diffcp_if.py:91-92);
    //      dA_ij = x_j r_y,i - y_i r_x,j ,  db = y r_tau - r_y ,  dc = x r_tau - r_x        (oracle/cone_oracle.c adjoint_one: dQ = r Pi(z)^T antisymmetrised)
    double *dA = dAo + (size_t)inst * T.nnz_aug;
    for (int k = tid; k < T.nnz_aug; k += NT) {
        const int r = T.rowidx[k], c = T.colidx[k];
        dA[k] = (c < n) ? -(x[c] * ry[r] - y[r] * rx[c]) : fma(y[r], rt, -ry[r]);
    }
    for (int j = tid; j <= n; j += NT) dqo[j * sdqk + inst * sdqb] = (j < n) ? fma(x[j], rt, -rx[j]) : 0.0;
#ifdef CE_TIMING
    __syncthreads();
    if (tid == 0) for (int k = 0; k < 8; k++) dA[k] = (double)ls_tacc[k];
#endif
    if (tid == 0) { if (adj_status) adj_status[inst] = (live ? 1 : 0) | status_or; if (iters_o) iters_o[inst] = itn; }
    __syncthreads();          // (the next listed instance reuses every LDS vector)
  }
}
