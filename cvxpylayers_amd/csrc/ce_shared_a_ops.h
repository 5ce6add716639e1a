// ce_shared_a_ops.h -- products with a SHARED matrix split into rows with a single entry and r <= 64 "dense" rows (ce_shared_a_fwd.h,
// ce_shared_a.h).  The dense rows are stored transposed, A_d^T [n][RP] row-major and zero padded, and stream from L2 (every workgroup of
// the launch reads the same matrix); the singleton rows are a gather.  Both directions keep several wide, line-covering loads in flight
// per lane: with one workgroup per instance the latency of the L2 stream, not its bandwidth, is what the products cost.
#pragma once
#ifndef SA_UR64
#define SA_UR64 2
#endif

// fields the product routines read (SaFwd and SaSplit both carry them):
//   r, AdT, drow[r], srow_col[m] (-1: not a singleton row), srow_val[m], scol_ptr[n + 1], scol_row[]
struct SaSplit {
    int r, RP;
    const double *AdT;
    const int *drow, *srow_col;
    const double *srow_val;
    const int *scol_ptr, *scol_row;
    const int *rowslot;          // [m] slot a of a dense row, -1 otherwise
    const int *sing_i;           // [n] the singleton row of column j when it has exactly one (the rule: bounds, -I embeddings), -1: none, -2: several (walk scol_ptr / scol_row)
    const double *sing_v;        // [n] its value (filled with srow_val by k_sa_fill_split)
};

// out[a] = sum_j AdT[j][a] xin[j]  (a < RP).  Thread (a-pair, g) sums rows j = g, g + ng, ... with eight 16-byte loads in flight (RP / 2 lanes
// cover a row: a wave reads whole rows, 1 KB per instruction); partials through `part` (2 NTH doubles of LDS).  Ends synchronised.
template <int NTH, int RP>
__device__ __forceinline__ void sa_dense_partials(const double *__restrict__ AdT, int n, const double *xin, double *part) {
    constexpr int HP = RP / 2, ng = NTH / HP;
    const int tid = threadIdx.x, a2 = tid % HP, g = tid / HP;
    const double2 *base = reinterpret_cast<const double2 *>(AdT) + a2;
    double2 acc[2] = {{0, 0}, {0, 0}};
    for (int j = g; j < n; j += 8 * ng) {
        double2 mv[8]; double xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int jj = j + u * ng, jc = jj < n ? jj : n - 1; mv[u] = base[(size_t)jc * HP]; xv[u] = jj < n ? xin[jj] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) { acc[u & 1].x = fma(mv[u].x, xv[u], acc[u & 1].x); acc[u & 1].y = fma(mv[u].y, xv[u], acc[u & 1].y); }
    }
    reinterpret_cast<double2 *>(part)[tid] = double2{acc[0].x + acc[1].x, acc[0].y + acc[1].y};      // part[g][a]: g * RP + 2 a2 (+1), g < 2 NTH / RP
}
template <int NTH, int RP>
__device__ __forceinline__ void sa_dense_times(const double *__restrict__ AdT, int n, const double *xin, double *part, double *out) {
    constexpr int ng = 2 * NTH / RP;
    sa_dense_partials<NTH, RP>(AdT, n, xin, part);
    __syncthreads();
    if (threadIdx.x < RP) {
        constexpr int CHK = ng < 16 ? ng : 16;       // the partial sums are requested in batches (the plain loop waited for every pair of them)
        double s_ = 0;
#pragma unroll
        for (int g0 = 0; g0 < ng; g0 += CHK) {
            double pv[CHK];
#pragma unroll
            for (int gg = 0; gg < CHK; gg++) pv[gg] = part[(g0 + gg) * RP + threadIdx.x];
#pragma unroll
            for (int gg = 0; gg < CHK; gg++) s_ += pv[gg];
        }
        out[threadIdx.x] = s_;
    }
    __syncthreads();
}

// out(j, sum_a AdT[j][a] w[a] + sum over the row's eight lanes of extra(j, lane)) for every j < n.  Eight lanes per row; load i of lane k is
// the 16-byte piece 8 i + k of the row, so the eight lanes of a row read whole 128-byte lines and a wave covers eight rows per
// instruction; four rows (4 RP / 16 loads) are in flight per lane.  w: RP doubles in LDS, 16-byte aligned.  No trailing barrier.
template <int NTH, int RP, class FE, class FO>
__device__ __forceinline__ void sa_rows_dot(const double *__restrict__ AdT, int n, const double *w, FE &&extra, FO &&out) {
    constexpr int NL = RP / 16, RS = NTH / 8;
    const int tid = threadIdx.x, k8 = tid & 7;
    const double2 *w2 = reinterpret_cast<const double2 *>(w) + k8;
    // this lane's entries of w, ONCE: they do not change over the rows, but out() writes LDS, so the compiler re-read them for every row, one read at a time
    // (4 serialised LDS round trips per row group behind the row's own loads: profiles/r03/i_serial_chains_static.txt)
    double2 wr[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) wr[i] = w2[8 * i];
    for (int j0 = tid >> 3; j0 < n; j0 += 4 * RS) {
        double2 rv[4][NL];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * RS, jc = j < n ? j : n - 1;
            const double2 *row = reinterpret_cast<const double2 *>(AdT + (size_t)jc * RP) + k8;
#pragma unroll
            for (int i = 0; i < NL; i++) rv[u][i] = row[8 * i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * RS;
            const bool ok = j < n;                       // (uniform over the eight lanes of a row)
            double a0 = ok ? extra(j, k8) : 0.0, a1 = 0;
#pragma unroll
            for (int i = 0; i < NL; i++) { a0 = fma(rv[u][i].x, wr[i].x, a0); a1 = fma(rv[u][i].y, wr[i].y, a1); }
            const double acc = group_reduce<8, false>(a0 + a1);
            if (ok && k8 == 0) out(j, acc);
        }
    }
}

// Both directions from ONE stream over A_d^T:  out(j, sum_a AdT[j][a] w[a] + extra) for every j  AND  the partial sums of
// v[a] = sum_j AdT[j][a] xin[j]:  part[wave * RP + a], to be summed over the NTH / 64 waves by the caller after a barrier.  Lane layout of
// sa_rows_dot (eight lanes per row, lane k holds a = 16 i + 2 k, 16 i + 2 k + 1); every lane accumulates its 2 RP / 16 entries of v over the
// rows it visits, the eight row groups of a wave are folded with three shuffles per entry.  No trailing barrier.
// sing_i / sing_v / yin (may be null): the singleton part of column j is  sing_v[j] * yin[sing_i[j]]  when the column has exactly one singleton row -- two loads that
// depend on j only and are requested with the row's matrix loads, instead of the three-level chain scol_ptr -> scol_row -> srow_val inside extra() that every row
// group used to wait for (config 5: 33 k of an 81 k-cycle LSQR iteration per product).  extra() still serves columns with several singleton rows.
// cex / cex_stride (may be null): a per-column scalar of the caller (the adjoint's c_j, strided in the boundary layout), requested WITH the row's matrix loads and
// handed to out(j, value, cex_j) -- a load inside out() would sit behind the row group's reduction, one exposed L2 round trip per row group.
template <int NTH, int RP, class FE, class FO>
__device__ __forceinline__ void sa_fused_pass(const double *__restrict__ AdT, int n, const double *w, const double *xin, FE &&extra, FO &&out, double *part,
                                              const int *__restrict__ sing_i = nullptr, const double *__restrict__ sing_v = nullptr, const double *yin = nullptr,
                                              const double *__restrict__ cex = nullptr, long cex_stride = 0) {
    constexpr int NL = RP / 16, RS = NTH / 8, UR = RP == 64 ? SA_UR64 : 4;      // rows in flight per lane (RP = 64: two, the accumulators need the registers)
    const int tid = threadIdx.x, k8 = tid & 7;
    const double2 *w2 = reinterpret_cast<const double2 *>(w) + k8;
    double2 vacc[NL], wr[NL];          // (wr: this lane's entries of w, read once -- see sa_rows_dot)
#pragma unroll
    for (int i = 0; i < NL; i++) { vacc[i] = double2{0.0, 0.0}; wr[i] = w2[8 * i]; }
    for (int j0 = tid >> 3; j0 < n; j0 += UR * RS) {
        double2 rv[UR][NL];
        int si[UR]; double sv[UR], ce[UR];
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int j = j0 + u * RS, jc = j < n ? j : n - 1;
            const double2 *row = reinterpret_cast<const double2 *>(AdT + (size_t)jc * RP) + k8;
#pragma unroll
            for (int i = 0; i < NL; i++) rv[u][i] = row[8 * i];
            si[u] = sing_i ? sing_i[jc] : -2; sv[u] = sing_i ? sing_v[jc] : 0.0;
            ce[u] = cex ? cex[(size_t)jc * cex_stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int j = j0 + u * RS;
            const bool ok = j < n;
            const double xv = ok ? xin[j] : 0.0;
            double a0 = 0, a1 = 0;
            if (ok) { if (si[u] >= 0) a0 = (k8 == 0) ? sv[u] * yin[si[u]] : 0.0; else if (si[u] == -2) a0 = extra(j, k8); }
#pragma unroll
            for (int i = 0; i < NL; i++) {
                const double2 wv = wr[i];
                a0 = fma(rv[u][i].x, wv.x, a0); a1 = fma(rv[u][i].y, wv.y, a1);
                vacc[i].x = fma(rv[u][i].x, xv, vacc[i].x); vacc[i].y = fma(rv[u][i].y, xv, vacc[i].y);
            }
            const double acc = group_reduce<8, false>(a0 + a1);
            if (ok && k8 == 0) { if constexpr (std::is_invocable_v<FO, int, double, double>) out(j, acc, ce[u]); else out(j, acc); }
        }
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const double vx = colsum_rows<8>(vacc[i].x), vy = colsum_rows<8>(vacc[i].y);
        if ((tid & 63) < 8) reinterpret_cast<double2 *>(part + (tid >> 6) * RP)[8 * i + k8] = double2{vx, vy};
    }
}

// y = A x :  out(i, value) for every row i  (srow_col: >= 0 column of a single-entry row, -1 empty row, <= -2 dense row).  vd: RP doubles, part: 2 NTH doubles of LDS.  Ends synchronised.
template <int NTH, int RP, class ST, class FO>
__device__ __forceinline__ void sa_A_times(const ST &F, int n, int m, const double *xin, double *part, double *vd, FO &&out) {
    sa_dense_times<NTH, RP>(F.AdT, n, xin, part, vd);
    for (int i = threadIdx.x; i < m; i += NTH) { const int c = F.srow_col[i]; if (c >= 0) out(i, F.srow_val[i] * xin[c]); else if (c == -1) out(i, 0.0); }      // (-1: a row without entries; dense rows carry -2 - slot here)
    for (int a = threadIdx.x; a < F.r; a += NTH) out(F.drow[a], vd[a]);
    __syncthreads();
}

// x = A^T y :  out(j, value) for every column j.  wyd: RP doubles of LDS (16-byte aligned).  Ends synchronised.
template <int NTH, int RP, class ST, class FO>
__device__ __forceinline__ void sa_AT_times(const ST &F, int n, const double *yin, double *wyd, FO &&out) {
    for (int a = threadIdx.x; a < RP; a += NTH) wyd[a] = a < F.r ? yin[F.drow[a]] : 0.0;
    __syncthreads();
    sa_rows_dot<NTH, RP>(F.AdT, n, wyd,
                         [&](int j, int k8) { double acc = 0; for (int k = F.scol_ptr[j] + k8; k < F.scol_ptr[j + 1]; k += 8) { const int i = F.scol_row[k]; acc = fma(F.srow_val[i], yin[i], acc); } return acc; },
                         out);
    __syncthreads();
}

// fills A_d^T and the singleton values of a split from the boundary's value order (solver sign: A = -A_cvx); one thread per entry.
// AdT must be zeroed beforehand.  rowslot[i]: slot a of a dense row, -1 for a singleton / empty row.
__global__ void k_sa_fill_split(int nnzA, int RP, const int *__restrict__ rowidx, const int *__restrict__ colidx, const int *__restrict__ rowslot,
                                const double *__restrict__ vals, double *__restrict__ AdT, double *__restrict__ srow_val,
                                const int *__restrict__ sing_i = nullptr, double *__restrict__ sing_v = nullptr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnzA) return;
    const int i = rowidx[k], a = rowslot[i];
    if (a >= 0) AdT[(size_t)colidx[k] * RP + a] = -vals[k];
    else { srow_val[i] = -vals[k]; if (sing_i && sing_i[colidx[k]] == i) sing_v[colidx[k]] = -vals[k]; }
}
