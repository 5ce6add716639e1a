// ce_global_mv.h -- products with matrices resident in global memory (size-generic forward / backward kernels).  Needs the DPP helpers of
// ce_forward_rt.h (group_reduce).
#pragma once
// ------------------------------------------------------------------------------------------------
// Products with a matrix that lives in GLOBAL memory (L2 / HBM workspace; residency modes 1 and 2).  The LDS versions of
// ce_common.h walk a row per thread (conflict-free in LDS, but 64 different cache lines per load instruction in global memory) with two
// loads in flight; here lanes walk along rows (whole 128-byte lines per 16-lane group), UNR loads are in flight per lane, and the
// partial-sum layout stays the one sum_parts() reads.
// out indexed by COLUMN:  part[ch][j] = sum_{i in chunk ch} Mat[i][j] v[i]
template <int UNR = 16>
__device__ __forceinline__ void mv_cols_g(const double *__restrict__ Mat, int ld, int rows, int cols, const double *v, double *part) {
    const int CH = chunks_for(cols);
    const int len = (rows + CH - 1) / CH;
    for (int idx = threadIdx.x; idx < cols * CH; idx += NT) {
        const int j = idx % cols, ch = idx / cols;
        const int i0 = ch * len, i1 = min(rows, i0 + len);
        double a0 = 0, a1 = 0;
        for (int i = i0; i < i1; i += UNR) {
            double mv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) { const int ii = i + u < i1 ? i + u : i1 - 1; mv[u] = Mat[(size_t)ii * ld + j]; }
#pragma unroll
            for (int u = 0; u < UNR; u++) { const double x = i + u < i1 ? v[i + u] : 0.0; if (u & 1) a1 = fma(mv[u], x, a1); else a0 = fma(mv[u], x, a0); }
        }
        part[ch * cols + j] = a0 + a1;
    }
}
// out indexed by ROW:  part[0][i] = sum_j Mat[i][j] v[j]  (sixteen lanes per row, DPP reduction; the other chunks of sum_parts() are zeroed)
template <int UNR = 8>
__device__ __forceinline__ void mv_rows_g(const double *__restrict__ Mat, int ld, int rows, int cols, const double *v, double *part) {
    const int c16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
    for (int i0 = 0; i0 < rows; i0 += NT / 16) {            // uniform trip count (the DPP reduction needs whole rows of lanes)
        const int i = i0 + rg;
        const bool ok = i < rows;
        const double *r = Mat + (size_t)(ok ? i : rows - 1) * ld;
        double a0 = 0, a1 = 0;
        for (int j = c16; j < cols; j += 16 * UNR) {
            double mv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) { const int jj = j + 16 * u; mv[u] = r[jj < cols ? jj : cols - 1]; }
#pragma unroll
            for (int u = 0; u < UNR; u++) { const int jj = j + 16 * u; const double x = jj < cols ? v[jj] : 0.0; if (u & 1) a1 = fma(mv[u], x, a1); else a0 = fma(mv[u], x, a0); }
        }
        const double a = group_reduce<16, false>(a0 + a1);
        if (ok && c16 == 0) part[i] = a;
    }
    const int CH = chunks_for(rows);
    for (int idx = threadIdx.x + rows; idx < CH * rows; idx += NT) part[idx] = 0.0;
}
// row norms of a global-memory matrix (max |.| or sum of squares), same lane layout: part[i], other chunks untouched (callers read part[i] only)
__device__ __forceinline__ void row_norms_g(const double *__restrict__ Mat, int ld, int rows, int cols, bool l2, double *part) {
    const int c16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
    for (int i0 = 0; i0 < rows; i0 += NT / 16) {
        const int i = i0 + rg;
        const bool ok = i < rows;
        const double *r = Mat + (size_t)(ok ? i : rows - 1) * ld;
        double a = 0;
        for (int j = c16; j < cols; j += 16 * 8) {
            double mv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int jj = j + 16 * u; mv[u] = jj < cols ? r[jj] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; u++) a = l2 ? fma(mv[u], mv[u], a) : fmax(a, fabs(mv[u]));
        }
        a = l2 ? group_reduce<16, false>(a) : group_reduce<16, true>(a);
        if (ok && c16 == 0) part[i] = a;
    }
}

