// ce_common.h -- shared device helpers of the cone engine (included inside an anonymous namespace)
#pragma once


constexpr int NT = 256;            // threads per workgroup
constexpr int NW = NT / 64;        // waves per workgroup
constexpr int CONVERGED_INTERVAL = 25;
constexpr int AA_MAX_REJECT = 10;     // safeguard rejections after which Anderson acceleration is switched off for the instance (oracle/cone_oracle.c)
constexpr int RESCALING_MIN_ITERS = 100;
constexpr int NUM_RUIZ_PASSES = 25;
constexpr int NUM_L2_PASSES = 1;
constexpr double MIN_SCALE = 1e-4, MAX_SCALE = 1e4;
constexpr double MIN_SCALE_VALUE = 1e-6, MAX_SCALE_VALUE = 1e6;
constexpr double TAU_FACTOR = 10.0, ZERO_CONE_FACTOR = 1000.0;
// rank tolerance of the adjoint's eliminations, relative to max |K|: ONE constant for k_backward_rt, both paths of the size-generic k_backward and the oracle's
// dense elimination (oracle/cone_oracle.c dense_solve_MT, `best <= 1e-11 * amax0`): a pivot below it is a FREE variable (diffcp's LSQR returns a solution of the
// consistent system there, diffcp_if.py:73-96), whatever kernel the template's size selects
constexpr double CE_RANK_TOL = 1e-11;

// DevT (the device-side template description) lives in ce_types.h, shared by all translation units

thread_local std::string g_err;

// compile-time loop: f(std::integral_constant<int, i>{}) for i < N (anything that indexes a register array by the loop variable: a plain `#pragma unroll` of a
// large body can silently stay rolled and put the array in scratch)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ------------------------------------------------------------------------------------------------
// workgroup reductions
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane has a valid source for these controls (quad_perm, row_mirror, row_half_mirror) and all rows / banks are enabled:
    // with bound_ctrl the "old" operand is dead, which spares the v_mov that would otherwise initialise the destination
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// all-reduce inside aligned groups of CH consecutive lanes (CH in 1,2,4,8,16); every lane of the wave must be active
template <int CH, bool MAX>
__device__ __forceinline__ double group_reduce(double v) {
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : a + b; };
    if constexpr (CH >= 2) v = op(v, dpp_mov<0xB1>(v));     // quad_perm [1,0,3,2]
    if constexpr (CH >= 4) v = op(v, dpp_mov<0x4E>(v));     // quad_perm [2,3,0,1]
    if constexpr (CH >= 8) v = op(v, dpp_mov<0x141>(v));    // row_half_mirror
    if constexpr (CH >= 16) v = op(v, dpp_mov<0x140>(v));   // row_mirror
    return v;
}

// DPP move with a row mask: rows outside the mask receive 0.0 (identity for sums and for maxima of non-negative values)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_rows(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// full wave64 reduction without the LDS crossbar: 4 intra-row DPP stages, row_bcast15 / row_bcast31, then lane 63 is
// read into scalar registers (the result is wave-uniform).  MAX is only used on non-negative values.
template <bool MAX>
__device__ __forceinline__ double wave_reduce_dpp(double v) {
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : a + b; };
    v = group_reduce<16, MAX>(v);
    v = op(v, dpp_mov_rows<0x142, 0xA>(v));    // row_bcast:15 into rows 1 and 3
    v = op(v, dpp_mov_rows<0x143, 0xC>(v));    // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// sum over the lanes of a wave that share (lane % LPR), LPR in {8, 16}: the result in every one of them.  The accumulators of the streaming passes
// (ce_shared_a_ops.h sa_fused_pass) are folded over the row groups of a wave with it.  Six to eight ds_bpermute round trips per value stood here (__shfl_xor
// 8 / 16 / 32); row_ror:8 stays inside a row of 16 lanes, and gfx950's v_permlane16_swap / v_permlane32_swap exchange rows / halves between two registers:
// with both operands the same value, register 0 + register 1 is v[lane] + v[lane ^ 16] (resp. ^ 32) -- no LDS crossbar, no wait.
#ifndef SA_COLSUM_DPP
#define SA_COLSUM_DPP 1
#endif
template <int LPR>
__device__ __forceinline__ double colsum_rows(double v) {
#if SA_COLSUM_DPP
    if constexpr (LPR == 8) v += dpp_mov<0x128>(v);          // row_ror:8
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v;
#else
    if constexpr (LPR == 8) v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
#endif
}

// sum over the wave, the same value in every lane.  Six ds_bpermute round trips (__shfl_xor) used to stand here: ~0.8 k cycles of LDS latency per reduction,
// several times per LSQR iteration of the shared-A adjoint kernel; the DPP form stays in the vector registers.
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce_dpp<false>(v); }
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// reduces K values over the workgroup; bit k of maxmask selects max instead of sum.  red: NW*K doubles of LDS.
// Must be called in workgroup-uniform control flow by ALL NT threads: wave_sum's DPP broadcast reads lane 63 of every wave (an inactive lane there is
// garbage, not a smaller sum), and the barriers below need every wave.
static_assert(NT % 64 == 0, "block_reduce: whole waves only");
// TRAIL false: no barrier behind the reads of `red` -- for callers that give every call site of a loop its own buffer (the next write to it is then at least
// one barrier away) and that synchronise before anybody else's LDS data is touched
template <int K, bool TRAIL = true>
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
        double t[NW];          // (requested together: see block_reduce_n)
#pragma unroll
        for (int w = 0; w < NW; w++) t[w] = red[w * K + k];
        double a = t[0];
#pragma unroll
        for (int w = 1; w < NW; w++) a = ((maxmask >> k) & 1u) ? fmax(a, t[w]) : a + t[w];
        v[k] = a;
    }
    if constexpr (TRAIL) __syncthreads();
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

// LDS doubles of the blocked Gauss-Jordan panels (G in global memory): column panel NP16 x 17, pivot block 16 x 17
__host__ __device__ inline size_t generic_gj_panel_doubles(int n) { const int np16 = 16 * ((n + 15) / 16); return (size_t)np16 * 17 + 16 * 17 + 2; }

// LDS doubles of the blocked pivoted elimination (generic backward kernel, K in global memory): column panel nkcap x 17, two 16 x 17 blocks, pivots
__host__ __device__ inline size_t generic_lu_panel_doubles(int nkcap) { return (size_t)nkcap * 17 + 2 * 16 * 17 + (size_t)nkcap + 2; }

#include "ce_math.h"

__device__ __forceinline__ double clamp_scale(double v) { return v < MIN_SCALE ? 1.0 : (v > MAX_SCALE ? MAX_SCALE : v); }

// scatter one instance's boundary values (batch-major row of [A_cvx | b_cvx] values) into dense solver form
template <int LU = 8>
__device__ __forceinline__ void load_instance(const DevT &T, const double *vals, double *A, double *bv) {
    const int n = T.n, m = T.m, lda = T.lda, nnz = T.nnz_aug;
    // The first chunk of (value, row, column) triples is requested BEFORE the zero fill and its barrier, and every chunk keeps 3 x LU loads per lane
    // in flight: with one workgroup per instance this phase is a chain of HBM latencies (5 chunks of 4 used to cost ~24 k cycles at the metric shape).
    double v0[LU]; int r0[LU], c0[LU];
#pragma unroll
    for (int u = 0; u < LU; u++) {
        const int k = threadIdx.x + u * NT, kc = k < nnz ? k : 0;
        v0[u] = vals[kc]; r0[u] = T.rowidx[kc]; c0[u] = k < nnz ? T.colidx[kc] : -1;      // c = -1: no entry
    }
    for (int i = threadIdx.x; i < m * lda; i += NT) A[i] = 0.0;
    for (int i = threadIdx.x; i < m; i += NT) bv[i] = 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < LU; u++) {
        if (c0[u] < 0) continue;
        if (c0[u] < n) A[r0[u] * lda + c0[u]] = -v0[u];      // solver sees A = -A_cvx (diffcp_if.py:65)
        else bv[r0[u]] = v0[u];                             // b = b_cvx          (diffcp_if.py:66)
    }
    for (int kb = LU * NT; kb < nnz; kb += LU * NT) {
#pragma unroll
        for (int u = 0; u < LU; u++) {
            const int k = kb + threadIdx.x + u * NT, kc = k < nnz ? k : 0;
            v0[u] = vals[kc]; r0[u] = T.rowidx[kc]; c0[u] = k < nnz ? T.colidx[kc] : -1;
        }
#pragma unroll
        for (int u = 0; u < LU; u++) {
            if (c0[u] < 0) continue;
            if (c0[u] < n) A[r0[u] * lda + c0[u]] = -v0[u];
            else bv[r0[u]] = v0[u];
        }
    }
    __syncthreads();
}

