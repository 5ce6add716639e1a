// ce_forward_v2.h -- forward kernel, second generation: one instance per 256-thread workgroup (4 wave64), 3 workgroups
// per CU, BOTH operand layouts of A-hat in registers, G in LDS.
//
//   at[T1] : thread (j1 = tid / CHT, c1 = tid % CHT) holds the column segment  A[T1*c1 + k][j1],  k < T1      -> A^T v
//   ar[T2] : thread (i2 = tid / CHA, c2 = tid % CHA) holds the row segment     A[i2][T2*c2 + k],  k < T2      -> A v
//   G      : (rho_x I + A^T Dy A)^{-1}, n x ldg doubles in LDS; thread (jg = tid / CHG, cg = tid % CHG) multiplies
//            the row segment G[jg][TG*cg + k], k < TG                                                          -> G v
// Segments are BLOCKED (contiguous, even length), so every vector operand is fetched with ds_read_b128 (two doubles per
// LDS instruction; segments of the lanes of one 16-lane LDS group are a multiple of 16 B apart and fall on distinct
// banks), and every product ends in a DPP butterfly over CHT / CHA / CHG (<= 16) adjacent lanes -- no LDS round trip for
// partial sums.  Compared with ce_forward_rt.h (512 threads, A rows in LDS): half the threads per instance (fewer
// reduction stages, fewer waves per barrier), no matrix traffic on the LDS pipe, 41 KB of LDS per instance -> 3
// instances per CU, and a register budget of 168 VGPRs with 104 of them holding the two tiles -> no scratch spills.
//
// The reduced KKT matrix is formed from row panels staged through the (not yet used) G region and inverted by
// Gauss-Jordan on a register tile (one barrier per pivot, pivot order k = kk + TG*cg so that register slots are static).
// Algorithm, constants and the order of operations per iterate are those of oracle/cone_oracle.c (SCS 3 restated).
#pragma once


template <int CHT, int T1, int CHA, int T2, int CHG, int TG, int NWARP = 4>
struct F2 {
    static constexpr int MP = CHT * T1;                       // padded rows
    static constexpr int NPa = CHA * T2, NPg = CHG * TG;
    static constexpr int NP = NPa > NPg ? NPa : NPg;          // padded columns
    static constexpr int VP = MP + NP + 2;                    // one (y | x | tau) vector
    static constexpr int OY = 0, OX = MP, OT = MP + NP;
    static constexpr int O_W = 0, O_UT = VP, O_U = 2 * VP, O_ZB = 3 * VP, O_GV = 4 * VP, O_PHI = 5 * VP;
    static constexpr int O_BV = 6 * VP, O_DV = O_BV + MP, O_CV = O_DV + MP, O_EV = O_CV + NP, O_TV = O_EV + NP, O_PX = O_TV + NP,
                         O_S1 = O_PX + NP, O_S2 = O_S1 + NP, O_S3 = O_S2 + NP, O_S4 = O_S3 + NP,
                         O_RED = O_S4 + NP, O_WP = O_RED + NWARP * 8, O_SC = O_WP + NWARP, O_MT = O_SC + 16, O_G = O_MT + 20;      // O_MT: ce_math.h coefficient table
    // S = A^T Dy A on the matrix cores: NTILE column tiles of 16, row panels of A-hat staged with pitch LDP (= 16 mod 32 doubles: the
    // two row groups of a 32-lane LDS pass fall 128 bytes apart).  The G region holds at least one panel of 4 rows.
    static constexpr int NTILE = (NPg + 15) / 16;
    static constexpr int LDP = (16 * NTILE) % 32 == 16 ? 16 * NTILE : 16 * NTILE + 16;
    static_assert(NTILE <= NWARP && LDP >= NPa, "one 16-row strip of S per wave; a panel row holds a whole row tile");
    static_assert(T1 % 2 == 0 && T2 % 2 == 0 && TG % 2 == 0, "segments must be even for 16-byte LDS reads");
    static_assert(CHT <= 16 && CHA <= 16 && CHG <= 16, "DPP butterflies stay inside a row of 16 lanes");
};

// structural zeros of a gathered tile (index -1: the load went through a clamped index, i.e. read entry 0 of THIS instance's values): multiplied by a 0 / 1 mask.
// The selecting form (-DF2_MASK_MUL=0; ADVICE round 5: a non-finite entry 0 then stays in its own slot instead of turning the tile's structural zeros into NaN)
// costs two v_cndmask per entry against one v_mul_f64: k_fwd2 1.526 against 1.498 ms, the step 2.18-2.22 against 2.11-2.16 ms on the same box
// (profiles/r06/o_ab_gather_mask_select_slower.log).  An instance whose entry 0 is not finite has no solution either way -- the mask only decides which of ITS
// slots carry the NaN -- so the faster form stays.
#ifndef F2_MASK_MUL
#define F2_MASK_MUL 1
#endif
#if F2_MASK_MUL
#define F2_SEL(ix, expr) ((expr) * ((ix) >= 0 ? 1.0 : 0.0))
#else
#define F2_SEL(ix, expr) ((ix) >= 0 ? (expr) : 0.0)
#endif
// a value that is equal in every lane, moved to scalar registers (frees VGPRs in the iteration loop)
__device__ __forceinline__ double uniform_d(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// FP32 all-reduce inside aligned groups of CH lanes (DPP, same stages as group_reduce)
template <int CH, bool MAX>
__device__ __forceinline__ float group_reduce_f(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    auto mov = [](float x, auto ctrl) { constexpr int CTRL = decltype(ctrl)::value; return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); };
    if constexpr (CH >= 2) v = op(v, mov(v, std::integral_constant<int, 0xB1>{}));
    if constexpr (CH >= 4) v = op(v, mov(v, std::integral_constant<int, 0x4E>{}));
    if constexpr (CH >= 8) v = op(v, mov(v, std::integral_constant<int, 0x141>{}));
    if constexpr (CH >= 16) v = op(v, mov(v, std::integral_constant<int, 0x140>{}));
    return v;
}

// sqrt(q) and 1/sqrt(q) to ~1 ulp without the fp64 sqrt + divide expansions (~60 VALU ops): hardware seed (v_rsq_f64) and two
// coupled Goldschmidt steps.  q > 0 and finite; callers guard q == 0.
__device__ __forceinline__ void sqrt_rsqrt(double q, double &s, double &rinv) {
    const double y = __builtin_amdgcn_rsq(q);
    double g = q * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    s = g; rinv = 2.0 * h;
}

// blocked register tile . LDS vector (vec already offset to the lane's segment, 16-byte aligned)
#ifndef F2_SEG_PARTS
#define F2_SEG_PARTS 1
#endif
template <int CH, int TT, int PARTS_ = F2_SEG_PARTS, bool REDUCE = true>
__device__ __forceinline__ double seg_dot(const double (&tile)[TT], const double *vec) {
    const double2 *v2 = reinterpret_cast<const double2 *>(vec);
    double a0 = 0, a1 = 0;       // (four chains were tried: the two extra accumulators push the iteration loop into scratch spills)
    // Long segments go in F2_SEG_PARTS fenced parts: the machine scheduler keeps only 1-4 of the 13 reads of a 26-term product in flight (it
    // serialised the A p_x phase completely: ten LDS round trips), with a fence every part's reads are issued together: one round trip per part.
    constexpr int NB = TT / 2, PARTS = NB > 8 ? PARTS_ : 1, H = (NB + PARTS - 1) / PARTS;
#pragma unroll
    for (int p = 0; p < PARTS; p++) {
#pragma unroll
        for (int k = p * H; k < (p + 1) * H && k < NB; k++) {
#ifdef F2_PROBE_HALFVEC      // (timing probe only, WRONG results: what the iteration would cost if a thread's vector operand were half as long -- a 2 x 13 tile instead of 1 x 26)
            const double2 v = v2[NB > 8 ? k % ((NB + 1) / 2) : k];
#else
            const double2 v = v2[k];
#endif
            a0 = fma(tile[2 * k], v.x, a0);
            a1 = fma(tile[2 * k + 1], v.y, a1);
        }
        if (p + 1 < PARTS) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (REDUCE) return group_reduce<CH, false>(a0 + a1); else return a0 + a1;
}
// LDS row segment . LDS vector
template <int CH, int TT, bool REDUCE = true>
__device__ __forceinline__ double seg_dot_lds(const double *row, const double *vec) {
    const double2 *r2 = reinterpret_cast<const double2 *>(row), *v2 = reinterpret_cast<const double2 *>(vec);
    double a0 = 0, a1 = 0;
    // two halves with a scheduling fence between them: left alone, the machine scheduler issues eight reads and then serialises the rest in pairs
    // (each pair a full LDS round trip); fenced, the second half's reads go out together once the first half's registers are free
    constexpr int H = (TT / 2 + 1) / 2;
#pragma unroll
    for (int k = 0; k < H; k++) {
        const double2 r = r2[k], v = v2[k];
        a0 = fma(r.x, v.x, a0);
        a1 = fma(r.y, v.y, a1);
    }
    if constexpr (TT / 2 > 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = H; k < TT / 2; k++) {
        const double2 r = r2[k], v = v2[k];
        a0 = fma(r.x, v.x, a0);
        a1 = fma(r.y, v.y, a1);
    }
    if constexpr (REDUCE) return group_reduce<CH, false>(a0 + a1); else return a0 + a1;
}

// Gather maps (template entry of every tile slot, -1 = structural zero) are THREAD-MAJOR: entry (t, k) at base[t * idx_stride<TT> + k],
// rows padded to a multiple of 4 ints (16-byte aligned), so that a thread fetches its row with dwordx4 loads off ONE base address
// (immediate offsets).  The slot-major layout of round 1 (base[k * NT + t]) needed one 64-bit address per slot beyond the
// 12-bit immediate range: 2 x 22 address VGPRs computed, spilled and reloaded in every refactor().
#ifndef F2_IDX_FENCE
#define F2_IDX_FENCE 8
#endif
template <int TT> constexpr int idx_stride = (TT + 3) & ~3;
// tile[k] = f(k, map entry) for k < TT, CHUNK loads of 4 entries in flight at a time
template <int TT, class F>
__device__ __forceinline__ void for_each_idx(const int *__restrict__ base, int t, F &&f) {
    // scalar base + unsigned 32-bit byte offset: the loads take the (SGPR base, VGPR offset) addressing form, no 64-bit vector address
    const int4 *ip = reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(base) + (unsigned)t * (unsigned)(idx_stride<TT> * sizeof(int)));
#pragma unroll
    for (int c4 = 0; c4 < idx_stride<TT> / 4; c4++) {
        const int4 ix = ip[c4];
        if (4 * c4 + 0 < TT) f(std::integral_constant<int, 0>{}, 4 * c4 + 0, ix.x);
        if (4 * c4 + 1 < TT) f(std::integral_constant<int, 1>{}, 4 * c4 + 1, ix.y);
        if (4 * c4 + 2 < TT) f(std::integral_constant<int, 2>{}, 4 * c4 + 2, ix.z);
        if (4 * c4 + 3 < TT) f(std::integral_constant<int, 3>{}, 4 * c4 + 3, ix.w);
        if (c4 % F2_IDX_FENCE == F2_IDX_FENCE - 1) __builtin_amdgcn_sched_barrier(0);       // bounded number of loads in flight (register peak)
    }
}

// Gather of one register tile in TWO fenced phases: every value load of the tile is issued (through a clamped index: structural zeros read entry 0 and are
// masked afterwards) before the first value is used.  Left to itself the compiler emitted `global_load ; s_waitcnt vmcnt(0) ; multiply` per entry -- and, for
// the guarded form `ix >= 0 ? vals[ix] : 0`, a branch around every load: 26 global round trips IN SERIES per gather, 25-36 k cycles of a 690 k-cycle instance
// (profiles/r05/c_gather_serialised.txt).   f(k, ix, value) consumes entry k.
#ifndef F2_GATHER_CHUNK
#define F2_GATHER_CHUNK 14
#endif
template <int TT, class F>
__device__ __forceinline__ void gather_tile(const int *__restrict__ base, int t, const double *__restrict__ vals, F &&f) {
    const int4 *ip = reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(base) + (unsigned)t * (unsigned)(idx_stride<TT> * sizeof(int)));
    int4 ix4[idx_stride<TT> / 4];
#pragma unroll
    for (int c4 = 0; c4 < idx_stride<TT> / 4; c4++) ix4[c4] = ip[c4];
    // in chunks of F2_GATHER_CHUNK entries: one global round trip per chunk, and a register peak of one chunk (the whole tile at once pushed loop-carried
    // values of the iteration into scratch)
    constexpr int CHK = F2_GATHER_CHUNK < TT ? F2_GATHER_CHUNK : TT;
#pragma unroll
    for (int k0 = 0; k0 < TT; k0 += CHK) {
        double raw[CHK];
#pragma unroll
        for (int u = 0; u < CHK; u++) {
            const int k = k0 + u < TT ? k0 + u : TT - 1;
            const int4 q = ix4[k >> 2];
            const int ix = (k & 3) == 0 ? q.x : ((k & 3) == 1 ? q.y : ((k & 3) == 2 ? q.z : q.w));
            raw[u] = vals[ix < 0 ? 0 : ix];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CHK; u++) {
            const int k = k0 + u;
            if (k < TT) {
                const int4 q = ix4[k >> 2];
                const int ix = (k & 3) == 0 ? q.x : ((k & 3) == 1 ? q.y : ((k & 3) == 2 ? q.z : q.w));
                // a structural zero read entry 0 through the clamped index: its bits are cleared here (two v_and_b32 with the index's sign mask -- integer
                // operations cannot be turned back into a branch around the load, and a non-finite entry 0 cannot leak into the tile as Inf * 0 = NaN)
                const int msk = ~(ix >> 31);
                f(k, ix, __hiloint2double(__double2hiint(raw[u]) & msk, __double2loint(raw[u]) & msk));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Orders LDS writes of this wave before LDS reads of this wave (other lanes' data).  LDS instructions of one wave execute in order,
// so no hardware wait is needed beyond what the compiler inserts for the data dependence; the fences keep the COMPILER from moving
// the reads above the writes.
__device__ __forceinline__ void wave_lds_exchange() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// thread coordinates, re-derived from an opaque copy of the thread id wherever they are needed: they then are short-lived
// values (a handful of VALU ops) instead of kernel-lived registers competing with the tiles
template <int CHT, int CHA, int CHG>
struct F2Co {
    int t, j1, c1, i2, c2, jg, cg;
    // wave = index of the wave in the workgroup (kept in a scalar register); the lane id is recomputed by volatile asm so
    // that no VGPR has to carry the thread id through the kernel (the allocator would spill it and reload it from scratch
    // at the top of every iteration)
    __device__ __forceinline__ static int thread_id(int wave) {
        int lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        return (wave << 6) + lane;
    }
    __device__ __forceinline__ explicit F2Co(int wave) {
        // unsigned: the signed quotient / remainder of a value the compiler cannot prove non-negative is a 6-instruction sequence per pair, and the products
        // T1 * c1 ... below become quarter-rate 32-bit multiplies; with the remainders' ranges known they are shifts / 24-bit multiplies
        const unsigned u = (unsigned)thread_id(wave);
        t = (int)u;
        j1 = (int)(u / CHT); c1 = (int)(u % CHT); i2 = (int)(u / CHA); c2 = (int)(u % CHA); jg = (int)(u / CHG); cg = (int)(u % CHG);
    }
};

// block_reduce_n (ce_forward_rt.h) with the wave index handed in as a scalar and the lane id recomputed on the spot: the
// per-wave slot address is then scalar, and no VGPR carries the thread id through the main loop for it (it used to be the
// kernel's last spill: 4 bytes per lane written to scratch at set-up, = 4 MB of HBM writes per launch of the metric batch)
template <int K, int NWV>
__device__ __forceinline__ void block_reduce_w(double (&v)[K], unsigned maxmask, double *red, int wave) {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = ((maxmask >> k) & 1u) ? wave_reduce_dpp<true>(v[k]) : wave_reduce_dpp<false>(v[k]);
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double t[NWV];          // the waves' partial results, requested together (the plain chain waited for each of them: NWV LDS round trips in series behind the barrier)
#pragma unroll
        for (int w = 0; w < NWV; w++) t[w] = red[w * K + k];
        double a = t[0];
#pragma unroll
        for (int w = 1; w < NWV; w++) a = ((maxmask >> k) & 1u) ? fmax(a, t[w]) : a + t[w];
        v[k] = a;
    }
    __syncthreads();
}

#ifdef CE_TIMING
#define F2_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) f2_tstamp[i] = __builtin_readcyclecounter(); } while (0)
// cycles per phase of the iteration, accumulated over the iterations of the instance (wave 0's clock; the values are the same in every lane: scalar registers)
#define F2_ACC(k) do { const long long t1_ = __builtin_readcyclecounter(); f2_tacc[k] += t1_ - f2_t0; f2_t0 = t1_; } while (0)
#define F2_EACC(k) do { const long long t1_ = __builtin_readcyclecounter(); f2_eacc[k] += t1_ - f2_t0; f2_t0 = t1_; } while (0)
#define F2_T0() do { f2_t0 = __builtin_readcyclecounter(); } while (0)
#else
#define F2_STAMP(i) do { } while (0)
#define F2_ACC(k) do { } while (0)
#define F2_EACC(k) do { } while (0)
#define F2_T0() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------------
// PSD cone: projection of svec(S) onto the PSD cone by a workgroup-parallel cyclic Jacobi eigensolver in LDS.
//   Sm, Vm : k x k scratch (row-major), cs : (c, s) per pair.  All NT threads take part; ends synchronised.
// Rounds follow the round-robin tournament (k-1 rounds of k/2 DISJOINT pairs, whose rotations commute): per round one thread
// per pair computes the rotation, then all threads apply S <- S J, V <- V J (column pass) and S <- J^T S (row pass).
// Same rotation formulas and svec convention (lower triangle, column-major, sqrt(2) off-diagonals) as oracle/cone_oracle.c.
// psd_jacobi: eigendecomposition only -- on return diag(Sm) holds the eigenvalues and the COLUMNS of Vm the eigenvectors.
template <int NTH = 256>
__device__ __forceinline__ void psd_jacobi(const double *zsvec, int k, double *Sm, double *Vm, double *cs, double *red) {
    constexpr int NT = NTH, NW = NTH / 64;
    const int tid = threadIdx.x;
    const int K = (k + 1) & ~1;               // players of the tournament (a dummy if k is odd)
    // svec -> symmetric matrix
    for (int idx = tid; idx < k * k; idx += NT) {
        const int i = idx / k, j = idx - i * k;
        const int a = i >= j ? i : j, b = i >= j ? j : i;                 // lower-triangle entry (a, b), column-major packed
        const int pos = b * k - (b * (b - 1)) / 2 + (a - b);
        const double v = zsvec[pos];
        Sm[idx] = (a == b) ? v : v * M_SQRT1_2;
        Vm[idx] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 40; sweep++) {
        double r[2] = {0, 0};                    // off-diagonal and diagonal squared norms
        for (int idx = tid; idx < k * k; idx += NT) { const int i = idx / k, j = idx - i * k; const double v = Sm[idx]; if (i == j) r[1] = fma(v, v, r[1]); else r[0] = fma(v, v, r[0]); }
        block_reduce_n<2, NW>(r, 0u, red);
        if (r[0] <= 1e-30 * (r[0] + r[1]) || r[0] == 0.0) break;          // uniform
        for (int rd = 0; rd < K - 1; rd++) {
            if (tid < K / 2) {
                int p = (tid == 0) ? K - 1 : (rd + tid) % (K - 1);
                int q = (rd + K - 1 - tid) % (K - 1);
                if (p > q) { const int t_ = p; p = q; q = t_; }
                double c = 1.0, sn = 0.0;
                if (q < k) {
                    const double apq = Sm[p * k + q];
                    if (apq != 0.0) {
                        const double theta = (Sm[q * k + q] - Sm[p * k + p]) / (2 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                        c = 1 / sqrt(t * t + 1); sn = t * c;
                    }
                } else { p = -1; }
                cs[4 * tid] = c; cs[4 * tid + 1] = sn; cs[4 * tid + 2] = (double)p; cs[4 * tid + 3] = (double)q;
            }
            __syncthreads();
            // column pass on S and V:  (x_p, x_q) <- (c x_p - s x_q, s x_p + c x_q) for every row
            for (int idx = tid; idx < (K / 2) * k * 2; idx += NT) {
                const int which = idx / ((K / 2) * k), rem = idx - which * (K / 2) * k;
                const int pi = rem / k, row = rem - pi * k;
                const int p = (int)cs[4 * pi + 2], q = (int)cs[4 * pi + 3];
                if (p < 0) continue;
                const double c = cs[4 * pi], sn = cs[4 * pi + 1];
                double *M = which ? Vm : Sm;
                const double a = M[row * k + p], b = M[row * k + q];
                M[row * k + p] = c * a - sn * b; M[row * k + q] = sn * a + c * b;
            }
            __syncthreads();
            // row pass on S
            for (int idx = tid; idx < (K / 2) * k; idx += NT) {
                const int pi = idx / k, col = idx - pi * k;
                const int p = (int)cs[4 * pi + 2], q = (int)cs[4 * pi + 3];
                if (p < 0) continue;
                const double c = cs[4 * pi], sn = cs[4 * pi + 1];
                const double a = Sm[p * k + col], b = Sm[q * k + col];
                Sm[p * k + col] = c * a - sn * b; Sm[q * k + col] = sn * a + c * b;
            }
            __syncthreads();
        }
    }
}
template <int NTH = 256>
__device__ __forceinline__ void psd_project(double *zsvec, int k, double *Sm, double *Vm, double *cs, double *red) {
    constexpr int NT = NTH;
    const int tid = threadIdx.x;
    psd_jacobi<NTH>(zsvec, k, Sm, Vm, cs, red);
    // eigenvalues -> cs (clipped at 0), then svec of V diag(w+) V^T
    for (int i = tid; i < k; i += NT) cs[i] = fmax(Sm[i * k + i], 0.0);
    __syncthreads();
    for (int pos = tid; pos < k * (k + 1) / 2; pos += NT) {
        // unpack pos -> (a, b), a >= b, column-major lower triangle
        int b = 0, rem = pos;
        while (rem >= k - b) { rem -= k - b; b++; }
        const int a = b + rem;
        double acc = 0;
        for (int e = 0; e < k; e++) acc = fma(Vm[a * k + e] * cs[e], Vm[b * k + e], acc);
        zsvec[pos] = (a == b) ? acc : acc * M_SQRT2;
    }
    __syncthreads();
}

#ifndef F2_WPS
#define F2_WPS 3
#endif
// F2_GJ_MFMA = 1 (default): the inversion of the reduced KKT matrix S runs as a blocked SWEEP on the matrix cores, on the accumulators the S formation left in
// registers (see refactor()); 0: the blocked Gauss-Jordan on the (jg, cg) register tile (rounds 2-4; still what the quadratic-objective instantiations run).
#ifndef F2_GJ_MFMA
#define F2_GJ_MFMA 1
#endif
// HASP: quadratic objective 1/2 x^T P x (SCS 3's QP embedding, oracle/cone_oracle.c solve_one): P-hat = E P E joins the reduced
// KKT matrix S = rho_x I + P-hat + A-hat^T Dy A-hat, tau-tilde becomes the positive root of a quadratic, the dual residual and
// the gap get their P terms.  Pvals: (B, nnzP) values in the template's P structure; idx_p: gather map of the (jg, cg) tile
// layout of G (row jg, columns TG*cg + k), -1 = structural zero (one-triangle structures map (i,j) and (j,i) to one entry).
// WL ("wave-local cones"): the host has ordered the rows so that no cone block straddles the rows of two waves in the (i2, c2) row
// layout (cone_engine.hip pack_rows; nonnegative rows are the filler and count as cones of dimension 1; row_perm maps kernel rows
// back to the template's rows).  A row thread then exchanges its cone's values with lanes of ITS OWN wave only -- LDS is in
// order per wave, no workgroup barrier -- which removes one of the two barriers of every equilibration pass and fuses the cone
// projection + relaxed update into the A p_x phase: 3 barriers per iteration instead of 4.
template <int CHT, int T1, int CHA, int T2, int CHG, int TG, bool PSD = false, int NTH = 256, bool HASP = false, bool WL = false>
__global__ void __launch_bounds__(NTH, (NTH == 256 ? ((HASP || PSD) ? 2 : F2_WPS) : 2))       // quadratic-objective / PSD variants: 2 workgroups per CU (256 VGPRs), their extra live values push the tiles of the 3-per-CU build into scratch
k_fwd2(DevT T, ce_settings S, const double *__restrict__ Avals, const double *__restrict__ qv, long sqk, long sqb,
       const int *__restrict__ idx_at, const int *__restrict__ idx_ar, const int *__restrict__ idx_b,
       double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ so, int *__restrict__ iters_o,
       int *__restrict__ status_o, double *__restrict__ resid_o,
       const double *__restrict__ Pvals_g = nullptr, int nnzP = 0, const int *__restrict__ idx_p = nullptr,
       const int *__restrict__ row_perm = nullptr, const int *__restrict__ order = nullptr, int *__restrict__ iters2 = nullptr) {
    static_assert(!(WL && (PSD || HASP)), "wave-local cone exchange: plain cones only");
    constexpr int NT = NTH, NW = NTH / 64;        // threads / waves per workgroup of this instantiation (shadow the file-level defaults)
    using L = F2<CHT, T1, CHA, T2, CHG, TG, NW>;
    using Co = F2Co<CHT, CHA, CHG>;
    constexpr int MP = L::MP, NP = L::NP, VP = L::VP, OY = L::OY, OX = L::OX, OT = L::OT;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    enum { SC_NB0 = 0, SC_NC0, SC_SIGMA, SC_SUMLOG, SC_RP, SC_RD, SC_GAP };
    double *const sc = sm + L::O_SC;
    double *const red = sm + L::O_RED;
    int *const socr = reinterpret_cast<int *>(sm + L::O_G);       // [MP] first row of the row's SOC (or -1)
    int *const socd = socr + MP;                                   // [MP] its dimension (0: not an SOC row)
    double *const Gm = sm + L::O_G + MP;                           // 2*MP ints = MP doubles

    // order: workgroup -> instance.  Workgroups are dispatched in index order, so a permutation sorted by EXPECTED duration, longest first, shortens the
    // tail of the launch (the last slots to drain run the short instances); the host derives it from the iteration counts of the previous call (ce_set_dispatch_history)
    // order[gridDim.x] says whether the history has been PREDICTIVE (k_dispatch_order: the iteration counts of the last two calls fell into the same check interval for
    // most instances); on unrelated batches the permutation would only scatter the instances' rows over HBM, and the workgroups keep the index order
    const int tid = threadIdx.x, inst = (order && order[gridDim.x]) ? order[blockIdx.x] : blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = T.n, m = T.m, l = n + m + 1, ldg = T.ldg, nq = T.nq, z = T.z;
    const int gsz = max(max(n * ldg, 4 * L::LDP), 16 * NP);        // doubles of the G region: G itself, one 4-row panel of the S formation, the exchange
                                                                   // buffers of the blocked inversion (cone_engine.hip f2_fits sizes it the same way)
    const double *const vals = Avals + (size_t)inst * T.nnz_aug;

#ifdef CE_TIMING
    __shared__ long long f2_tstamp[24], f2_tstamp2[8];
    if (threadIdx.x < 24) f2_tstamp[threadIdx.x] = 0;
    long long f2_tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, f2_t0 = 0;
    long long f2_eacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // equilibration pass [0..3], Gauss-Jordan block [4..7]
#endif
    F2_STAMP(0);
    for (int i = tid; i < L::O_G; i += NT) sm[i] = 0.0;
    const double *const mtab = sm + L::O_MT;
    for (int i = tid; i < MP; i += NT) {
        int r0 = -1, d = 0;
        if (i < m) { const int c = T.rowcone[i]; if (c >= 0) { r0 = T.qoff[c]; d = T.qoff[c + 1] - r0; } }
        if constexpr (PSD) {      // rows of PSD cones: block start and NEGATIVE block length (same block averaging, separate projection)
            for (int c = 0; c < T.ns; c++) if (i >= T.soff[c] && i < T.soff[c + 1]) { r0 = T.soff[c]; d = -(T.soff[c + 1] - r0); }
            if (i >= T.eoff && i < T.eoff + 3 * (T.nep + T.np)) { r0 = T.eoff + 3 * ((i - T.eoff) / 3); d = -3; }      // exponential / power cone triples
        }
        socr[i] = r0; socd[i] = d;
    }
    __syncthreads();
    ce_math_table_init(sm + L::O_MT, tid);
    for (int i = tid; i < m; i += NT) { const int ix = idx_b[i]; const double v = vals[ix < 0 ? 0 : ix]; sm[L::O_BV + i] = ix >= 0 ? v : 0.0; sm[L::O_DV + i] = 1.0; }
    for (int j = tid; j < n; j += NT) { sm[L::O_CV + j] = qv[j * sqk + inst * sqb]; sm[L::O_EV + j] = 1.0; }
    __syncthreads();
    {
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT) r[0] = fmax(r[0], fabs(sm[L::O_BV + i]));
        for (int j = tid; j < n; j += NT) r[1] = fmax(r[1], fabs(sm[L::O_CV + j]));
        block_reduce_w<2, NW>(r, 3u, red, wave);
        sc[SC_NB0] = r[0]; sc[SC_NC0] = r[1]; sc[SC_SIGMA] = 1.0; sc[9] = 0.0;      // sc[9]: safeguard rejections so far
        sc[10] = sqrt((double)l); sc[11] = 1e10; sc[12] = 1e-300;                                    // constants of rare paths of the iteration loop (read back through opaque indices)
    }
    F2_STAMP(1);
    // ---------------------------------------------------------------- equilibration (SCS normalize: 25 Ruiz passes + 1 l2 pass)
    // The passes run on FP32 copies of the tiles: D and E are preconditioners -- any positive diagonal scaling gives an equivalent
    // problem, and termination is tested on un-normalised residuals -- so the Ruiz factors only need single precision (they
    // are accumulated in double).  The fp64 iteration tiles are afterwards built as A * (D * E) in double from the final D, E
    // (materialize_*), so both layouts hold exactly the same matrix.  FP32 halves the VALU cost and the registers of this phase.
    if (S.normalize) {
        const Co co(wave);
        const int j1 = co.j1, c1 = co.c1, i2 = co.i2, c2 = co.c2;
        const bool own1 = (c1 == 0) && (j1 < n), own2 = (c2 == 0) && (i2 < m);
        // tiles as packed pairs: the scaling of a pass is one v_pk_mul_f32 per pair and factor (gfx950 packed fp32 runs at twice the
        // scalar fp32 rate), the inf-norms are v_max3_f32 chains, 1/sqrt is the hardware v_rsq_f32 (D, E are preconditioners:
        // any positive scaling is valid, 1 ulp of single precision is more than enough)
        typedef float f2v __attribute__((ext_vector_type(2)));
        f2v atv[T1 / 2], arv[T2 / 2];
        float pf[HASP ? TG : 1];
        if constexpr (HASP) {
            const double *pv = Pvals_g + (size_t)inst * nnzP;
#pragma unroll
            for (int k = 0; k < TG; k++) { const int ix = idx_p[tid * idx_stride<TG> + k]; const double v = pv[ix < 0 ? 0 : ix]; pf[k] = ix >= 0 ? (float)v : 0.0f; }
        }
        float *const fPn = reinterpret_cast<float *>(sm + L::O_S3);          // column norms of P-hat (= row norms: symmetric)
        // (every gather below loads UNCONDITIONALLY through a clamped index and selects afterwards: a guarded load `ix >= 0 ? vals[ix] : 0` is a branch around the load with
        //  its own `s_waitcnt vmcnt(0)` -- 26 or 52 global round trips IN SERIES per gather, 25-36 k cycles each: profiles/r05/c_gather_serialised.txt)
        gather_tile<T1>(idx_at, tid, vals, [&](int k, int ix, double v) { atv[k >> 1][k & 1] = ix >= 0 ? (float)(-v) : 0.0f; });   // A = -A_cvx (diffcp_if.py:65)
        gather_tile<T2>(idx_ar, tid, vals, [&](int k, int ix, double v) { arv[k >> 1][k & 1] = ix >= 0 ? (float)(-v) : 0.0f; });
        float *const fEt0 = reinterpret_cast<float *>(sm + L::O_S1), *const fEt1 = reinterpret_cast<float *>(sm + L::O_S2);
        float *const fDt0 = reinterpret_cast<float *>(sm + L::O_U + OY), *const fDt1 = reinterpret_cast<float *>(sm + L::O_UT + OY);
        float *const fRn = reinterpret_cast<float *>(sm + L::O_ZB + OY);
        auto clampf = [](float v) -> float { return v < (float)MIN_SCALE ? 1.0f : (v > (float)MAX_SCALE ? (float)MAX_SCALE : v); };
        double Eacc = 1.0, Dacc = 1.0;        // accumulated scalings of this thread's column / row (owners write them once, after the passes)
        // The column-layout tile belongs to ONE column and the row-layout tile to ONE row: their own factor is the same for all 26 entries, so it is
        // kept as a scalar (ecum, dcum) that multiplies the tile's norm instead of being multiplied into every entry in every pass (half the v_pk_mul_f32)
        float ecum = 1.0f, dcum = 1.0f;
        const int blk_r0 = (i2 < m) ? socr[i2] : 0, blk_d = (i2 < m) ? abs(socd[i2]) : 0;      // this row's cone block (read once: two LDS round trips less per pass)
        F2_STAMP(14);      // (the two FP32 tile gathers: first touch of the instance's values)
        F2_T0();
        for (int pass = 0; pass < NUM_RUIZ_PASSES + NUM_L2_PASSES; pass++) {
            const bool l2 = pass >= NUM_RUIZ_PASSES;
            float *const fEt = (pass & 1) ? fEt1 : fEt0;                  // column scaling of this pass (x-indexed)
            float *const fDt = (pass & 1) ? fDt1 : fDt0;                  // row scaling of this pass (y-indexed)
            float cn = 0, rn = 0;
            if (l2) {
#pragma unroll
                for (int k = 0; k < T1 / 2; k++) { cn = fmaf(atv[k].x, atv[k].x, cn); cn = fmaf(atv[k].y, atv[k].y, cn); }
#pragma unroll
                for (int k = 0; k < T2 / 2; k++) { rn = fmaf(arv[k].x, arv[k].x, rn); rn = fmaf(arv[k].y, arv[k].y, rn); }
                cn = ecum * sqrtf(group_reduce_f<CHT, false>(cn)); rn = dcum * sqrtf(group_reduce_f<CHA, false>(rn));
            } else {
                float c0 = 0, c1_ = 0, r0 = 0, r1 = 0;
#pragma unroll
                for (int k = 0; k < T1 / 2; k += 2) { c0 = fmaxf(fmaxf(c0, fabsf(atv[k].x)), fabsf(atv[k].y)); if (k + 1 < T1 / 2) c1_ = fmaxf(fmaxf(c1_, fabsf(atv[k + 1].x)), fabsf(atv[k + 1].y)); }
#pragma unroll
                for (int k = 0; k < T2 / 2; k += 2) { r0 = fmaxf(fmaxf(r0, fabsf(arv[k].x)), fabsf(arv[k].y)); if (k + 1 < T2 / 2) r1 = fmaxf(fmaxf(r1, fabsf(arv[k + 1].x)), fabsf(arv[k + 1].y)); }
                cn = ecum * group_reduce_f<CHT, true>(fmaxf(c0, c1_)); rn = dcum * group_reduce_f<CHA, true>(fmaxf(r0, r1));
            }
            if constexpr (HASP) {      // columns of [P-hat; A-hat]: the column norm of A-hat is combined with that of P-hat after the barrier
                const Co cop(wave);
                float pn = 0;
                if (l2) {
#pragma unroll
                    for (int k = 0; k < TG; k++) pn = fmaf(pf[k], pf[k], pn);
                    pn = group_reduce_f<CHG, false>(pn);          // squared
                } else {
#pragma unroll
                    for (int k = 0; k < TG; k++) pn = fmaxf(pn, fabsf(pf[k]));
                    pn = group_reduce_f<CHG, true>(pn);
                }
                if (cop.cg == 0 && cop.jg < n) fPn[cop.jg] = pn;
            } else {
                if (own1) fEt[j1] = __builtin_amdgcn_rsqf(clampf(cn));
            }
            if (own2) fRn[i2] = rn;          // raw row norms
            F2_EACC(0);      // norms, butterflies, column factor, row norms out
            if constexpr (WL) wave_lds_exchange(); else __syncthreads();
            if constexpr (HASP) {
                if (own1) { const float pn = fPn[j1]; fEt[j1] = __builtin_amdgcn_rsqf(clampf(l2 ? sqrtf(cn * cn + pn) : fmaxf(cn, pn))); }
            }
            // block sums of <= 12 rows: the CHA lanes of a row share the masked batch of reads (as in the iteration's cone norm) and add their shares with a DPP butterfly
            constexpr int NEQ = (12 + CHA - 1) / CHA;
            float ssh = 0;
            if (blk_d > 1 && blk_d <= 12) {   // (the reads past the block stay inside the vector)
                const int off = (int)__umul24((unsigned)c2, (unsigned)NEQ);
                const float *fr = fRn + blk_r0 + off;
                const int lim = blk_d - off;
                float v[NEQ];
#pragma unroll
                for (int u = 0; u < NEQ; u++) v[u] = fr[u];
#pragma unroll
                for (int u = 0; u < NEQ; u++) ssh += (u < lim) ? v[u] : 0.0f;
            }
            ssh = group_reduce_f<CHA, false>(ssh);
            if (own2) {
                float a = rn;
                const int r0 = blk_r0, d = blk_d;
                if (d > 1) {   // block-average inside the SOC / PSD block so the scaled cone is still the cone
                    float s0 = ssh, s1 = 0;
                    if (d <= 12) {
                    } else {
                        int i = 0;
                        for (; i + 1 < d; i += 2) { s0 += fRn[r0 + i]; s1 += fRn[r0 + i + 1]; }
                        if (i < d) s0 += fRn[r0 + i];
                    }
                    a = (s0 + s1) * __builtin_amdgcn_rcpf((float)d);
                }
                fDt[i2] = __builtin_amdgcn_rsqf(clampf(a));
            }
            F2_EACC(1);      // block sums, row factor
            __syncthreads();
            F2_EACC(2);      // the barrier
            {
                // every scaling factor of this pass is requested in ONE batch (the FP32 tiles leave the registers for it); multiplying as the values
                // arrive -- what the scheduler made of the plain loops -- kept two reads in flight: ~10 LDS round trips per pass instead of ~2
                const float ej = fEt[j1 < NP ? j1 : 0];            // pad entries are 0
                const float di = fDt[i2 < MP ? i2 : 0];
                const f2v *d2 = reinterpret_cast<const f2v *>(fDt + T1 * c1);
                const f2v *e2 = reinterpret_cast<const f2v *>(fEt + T2 * c2);
                f2v dd[T1 / 2], ee[T2 / 2];
#pragma unroll
                for (int k = 0; k < T1 / 2; k++) dd[k] = d2[k];
#pragma unroll
                for (int k = 0; k < T2 / 2; k++) ee[k] = e2[k];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < T1 / 2; k++) atv[k] *= dd[k];
#pragma unroll
                for (int k = 0; k < T2 / 2; k++) arv[k] *= ee[k];
                ecum *= ej; dcum *= di;
                if constexpr (HASP) {
                    const Co cop(wave);
                    const float eg = fEt[cop.jg < NP ? cop.jg : 0];
                    const float2 *g2 = reinterpret_cast<const float2 *>(fEt + TG * cop.cg);
#pragma unroll
                    for (int k = 0; k < TG / 2; k++) { const float2 ee = g2[k]; pf[2 * k] *= eg * ee.x; pf[2 * k + 1] *= eg * ee.y; }
                }
                Eacc *= (double)ej; Dacc *= (double)di;
            }
            F2_EACC(3);      // factor reads + scaling
            // no barrier: the next pass writes the other ping-pong buffers (and the row norms, last read before the barrier above)
        }
        F2_STAMP(15);      // (the 26 passes)
        if (own1) sm[L::O_EV + j1] = Eacc;
        if (own2) sm[L::O_DV + i2] = Dacc;
        __syncthreads();
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT) { const double v = sm[L::O_BV + i] * sm[L::O_DV + i]; sm[L::O_BV + i] = v; r[0] = fmax(r[0], fabs(v)); }
        for (int j = tid; j < n; j += NT) { const double v = sm[L::O_CV + j] * sm[L::O_EV + j]; sm[L::O_CV + j] = v; r[1] = fmax(r[1], fabs(v)); }
        block_reduce_w<2, NW>(r, 3u, red, wave);
        const double sigma = 1.0 / clamp_scale(fmax(r[0], r[1]));
        sc[SC_SIGMA] = sigma;
        for (int i = tid; i < m; i += NT) sm[L::O_BV + i] *= sigma;
        for (int j = tid; j < n; j += NT) sm[L::O_CV + j] *= sigma;
        for (int i = tid; i < VP; i += NT) { sm[L::O_U + i] = 0.0; sm[L::O_UT + i] = 0.0; sm[L::O_ZB + i] = 0.0; }
        for (int i = tid; i < NP; i += NT) { sm[L::O_S1 + i] = 0.0; sm[L::O_S2 + i] = 0.0; }
        __syncthreads();
    }

    F2_STAMP(2);
    double scale = S.scale, hg = 0, inv_den = 0;
    const double rho_x = S.rho_x, rtau = TAU_FACTOR, alpha = S.alpha;
    auto dyv = [&](int i) -> double { return (i < z) ? ZERO_CONE_FACTOR * scale : scale; };   // 1 / r_y

    // The iteration tiles.  They are (RE-)MATERIALISED from the instance's values (L2) and the final scalings D, E (LDS):
    // A-hat[r][j] = (-A_cvx[r][j]) * (D[r] * E[j]), the same expression for both layouts (bitwise consistent).  Not keeping
    // them alive across refactor() splits their live ranges around the register-hungry factorisation.
    double at[T1], ar[T2];
    auto materialize_at = [&](const Co &co) {
        const double ej = sm[L::O_EV + (co.j1 < NP ? co.j1 : 0)];
        const double *dv = sm + L::O_DV + T1 * co.c1;
        gather_tile<T1>(idx_at, co.t, vals, [&](int k, int ix, double v) { at[k] = F2_SEL(ix, -v * (dv[k] * ej)); });
    };
    auto materialize_ar = [&](const Co &co) {
        const double di = sm[L::O_DV + (co.i2 < MP ? co.i2 : 0)];
        const double *evs = sm + L::O_EV + T2 * co.c2;
        gather_tile<T2>(idx_ar, co.t, vals, [&](int k, int ix, double v) { ar[k] = F2_SEL(ix, -v * (di * evs[k])); });
    };
    // P-hat row segment of the (jg, cg) layout, re-materialised wherever it is needed (S formation, P-hat g_x, the residual check)
    double gPg = 0;                                                  // g_x^T P-hat g_x
    double *const PgV = Gm + gsz;                                // [NP] P-hat g_x   (HASP only; dynamic tail of the LDS carve)
    auto materialize_p = [&](const Co &co, double (&pg)[TG]) {
        const double *pv = Pvals_g + (size_t)inst * nnzP;
        const double ej = sm[L::O_EV + (co.jg < NP ? co.jg : 0)];
        const double *evs = sm + L::O_EV + TG * co.cg;
        gather_tile<TG>(idx_p, co.t, pv, [&](int k, int ix, double v) { pg[k] = F2_SEL(ix, v * (ej * evs[k])); });
    };
    // The column groups j1 == n and j1 == n + 1 (idle in the A^T product) carry phi as two extra "columns", so that the
    // A^T w_y phase also yields phi_y . w_y and phi_x . w_x (the numerator of tau-tilde) without a separate reduction.
    auto load_phi_tile = [&](const Co &co) {
        if (co.j1 == n) {
#pragma unroll
            for (int k = 0; k < T1; k++) at[k] = sm[L::O_PHI + OY + T1 * co.c1 + k];       // pads of PHI are zero
        } else if (co.j1 == n + 1) {
#pragma unroll
            for (int k = 0; k < T1; k++) at[k] = (T1 * co.c1 + k < n) ? sm[L::O_PHI + OX + T1 * co.c1 + k] : 0.0;
        }
    };

    // ---- (re)factor:  G <- (rho_x I + A^T Dy A)^{-1} (LDS);  g, h.g, phi.   Clobbers ZB, TV, PX, S1..S4.
    double g_scale = 0.0;          // the scale G (in LDS) was computed for; 0: none yet
    auto refactor = [&]() {
        F2_STAMP(7);
        const Co co(wave);
        const int tid = co.t;
        const int j1 = co.j1, c1 = co.c1, i2 = co.i2, c2 = co.c2, jg = co.jg, cg = co.cg;
        const bool own1 = (c1 == 0) && (j1 < n), own2 = (c2 == 0) && (i2 < m), owng = (cg == 0) && (jg < n);
        double sreg[TG];
#pragma unroll
        for (int s = 0; s < TG; s++) sreg[s] = 0.0;
        // RESCALE without refactoring.  Dy is proportional to the scale (zero-cone rows included), so with f = scale_new / scale_old
        //     S_new = rho I + f (S_old - rho I) = f (S_old + delta I),   delta = rho (1 - f) / f,        G_new = (1 / f) (I + delta G)^-1 G
        // and since rho_x = 1e-6 is tiny against the spectrum of A^T Dy A, x = |delta| |G|_F is ~1e-4: the Neumann series
        //     (I + delta G)^-1 G = G - delta G^2 + delta^2 G^3 - ...   =  Y_K,   Y_0 = G,  Y_{j+1} = G - delta G Y_j
        // reaches 1e-15 relative accuracy in K = 2-4 products of n x n matrices, against S formation + blocked Gauss-Jordan (three quarters of
        // a refactorisation, which costs as much as ~35 iterations and runs about once per instance after the initial one).  Y_j stays in the
        // (jg, cg) register tile of the inversion; a row of Y_j is spread over the CHG adjacent lanes of its row group and is broadcast from
        // there (ds_bpermute), G is read from LDS; only when x > 1e-2 (S nearly singular) the full refactorisation below runs.
        bool fast = false, g_in_lds = false;      // g_in_lds: the inversion has already written G to LDS (matrix-core sweep)
        if constexpr (!HASP && TG <= 14) {       // (the wide-tile variants have no registers to spare for Y, G and Z segments: they refactor)
            if (T.f2_neumann && g_scale > 0.0) {
                const double f = uniform_d(scale / g_scale), delta = uniform_d(rho_x * (1.0 - f) / f);
                double r[1] = {0};
                if (jg < n) {
                    const double2 *src = reinterpret_cast<const double2 *>(Gm + jg * ldg + TG * cg);
#pragma unroll
                    for (int s2 = 0; s2 < TG / 2; s2++) { const double2 v = src[s2]; r[0] = fma(v.x, v.x, fma(v.y, v.y, r[0])); }
                }
                block_reduce_w<1, NW>(r, 0u, red, wave);
                // x = |delta| |G|_F through its binary exponent (no fp64 literals: they would be hoisted into registers held across the iteration loop):
                // x < 2^-17 -> K = 2, < 2^-13 -> 3, < 2^-10 -> 4, < 2^-7 -> 7   (x^(K+1) <= ~1e-15), else the full refactorisation
                const int ex = __builtin_amdgcn_readfirstlane((__double2hiint(fabs(delta) * sqrt(r[0])) >> 20) & 0x7ff) - 1023;
                const int K = ex < -17 ? 2 : (ex < -13 ? 3 : (ex < -10 ? 4 : (ex < -7 ? 7 : 0)));
                if (K > 0) {
                    fast = true;
                    double greg[TG];
                    if (jg < n) {
                        const double2 *src = reinterpret_cast<const double2 *>(Gm + jg * ldg + TG * cg);
#pragma unroll
                        for (int s2 = 0; s2 < TG / 2; s2++) { const double2 v = src[s2]; sreg[2 * s2] = v.x; sreg[2 * s2 + 1] = v.y; }
                    }
#pragma unroll
                    for (int s = 0; s < TG; s++) greg[s] = sreg[s];
                    const int lane_base = (threadIdx.x & 63) & ~(CHG - 1);
                    for (int it = 0; it < K; it++) {
                        double zz[TG];
#pragma unroll
                        for (int s = 0; s < TG; s++) zz[s] = 0.0;
                        int goff = TG * cg;                       // LDS offset of G[kcol][TG cg], advanced row by row.  Opaque to the optimiser: the 56 row addresses are
                        asm volatile("" : "+v"(goff));            // invariant across `it` and would otherwise be hoisted into 56 VGPRs (the G / Y segments then spill)
#pragma unroll
                        for (int q = 0; q < CHG; q++) {
                            const int src_lane = (lane_base + q) << 2;
#pragma unroll
                            for (int s = 0; s < TG; s++) {
                                const int kcol = TG * q + s;
                                const int lo = __builtin_amdgcn_ds_bpermute(src_lane, __double2loint(sreg[s])), hi = __builtin_amdgcn_ds_bpermute(src_lane, __double2hiint(sreg[s]));
                                const double yk = __hiloint2double(hi, lo);                      // Y_j[jg][kcol]
                                if (kcol < n) {                                                   // uniform (rows of G beyond n do not exist)
                                    const double2 *gr = reinterpret_cast<const double2 *>(Gm + goff);
                                    goff += ldg;
#pragma unroll
                                    for (int s2 = 0; s2 < TG / 2; s2++) { const double2 v = gr[s2]; zz[2 * s2] = fma(yk, v.x, zz[2 * s2]); zz[2 * s2 + 1] = fma(yk, v.y, zz[2 * s2 + 1]); }
                                }
                            }
                        }
#pragma unroll
                        for (int s = 0; s < TG; s++) sreg[s] = fma(-delta, zz[s], greg[s]);
                    }
                    const double rf = 1.0 / f;
#pragma unroll
                    for (int s = 0; s < TG; s++) sreg[s] = (jg < n) ? sreg[s] * rf : 0.0;
                    __syncthreads();                 // every lane has finished reading the old G
                }
            }
        }
        if (!fast) {
        materialize_ar(co);
#pragma unroll
        for (int s = 0; s < TG; s++) sreg[s] = 0.0;
        // S = A-hat^T Dy A-hat on the matrix cores (v_mfma_f64_16x16x4_f64).  Row panels of A-hat are staged through the (not yet used)
        // G region with pitch LDP; wave w accumulates the 16-row strip S[16w .. 16w+15][:] as NTILE tiles of 16 x 16:
        //     D[M][N] += sum_K A[M][K] B[K][N],   A[M][K] = A-hat[i0 + K][16w + M] dy(i0 + K),   B[K][N] = A-hat[i0 + K][16J + N],
        // four panel rows per instruction; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15], i.e. BOTH operands are the panel
        // entry (row i0 + (l >> 4), column 16 * tile + (l & 15)): one ds_read_b64 per operand, 16 contiguous doubles per row group and
        // (LDP = 16 mod 32 doubles) the two row groups of a 32-lane LDS pass 128 bytes apart: conflict-free.  Accumulator layout
        // (MI355X guide, f64 MFMA): register r of lane l holds D[(l >> 4) + 4 r][l & 15].
        {
            constexpr int NTILE = L::NTILE, LDP = L::LDP;
            typedef double v4d __attribute__((ext_vector_type(4)));
            v4d acc[NTILE];
#pragma unroll
            for (int J = 0; J < NTILE; J++) acc[J] = v4d{0.0, 0.0, 0.0, 0.0};
            const int PR = (gsz / LDP) & ~3;                   // rows per panel (a multiple of the MFMA depth 4)
            const int lane = tid & 63, lg = lane >> 4, lc = lane & 15;
            for (int p0 = 0; p0 < m; p0 += PR) {
                const int p1 = min(m, p0 + PR), rows4 = (p1 - p0 + 3) & ~3;
                if (i2 >= p0 && i2 < p1) {
                    double2 *dst = reinterpret_cast<double2 *>(Gm + (i2 - p0) * LDP + T2 * c2);
#pragma unroll
                    for (int k = 0; k < T2 / 2; k++) dst[k] = make_double2(ar[2 * k], ar[2 * k + 1]);
                }
                // columns the row tiles do not cover, and the rows that pad the panel to a multiple of 4, are zero
                for (int i = tid; i < (p1 - p0) * (LDP - L::NPa); i += NT) Gm[(i / (LDP - L::NPa)) * LDP + L::NPa + i % (LDP - L::NPa)] = 0.0;
                for (int i = tid; i < (rows4 - (p1 - p0)) * LDP; i += NT) Gm[(p1 - p0) * LDP + i] = 0.0;
                __syncthreads();
                if (wave < NTILE) {
                    const double *prow = Gm + lg * LDP + lc;
                    for (int i0 = 0; i0 < rows4; i0 += 4, prow += 4 * LDP) {
                        const double a = prow[16 * wave] * dyv(p0 + i0 + lg);
#pragma unroll
                        for (int J = 0; J < NTILE; J++) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, prow[16 * J], acc[J], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
            if constexpr (F2_GJ_MFMA && !HASP) {
                // ---- inversion as a BLOCKED SWEEP on the matrix cores, in the accumulator layout the S formation ends in (no LDS round trip of S, no second
                // register tile).  Sweeping the symmetric matrix on the index block K = {k0 .. k0 + 3}, with P = S[K, K]^-1, R = S[K, :] and C = S[:, K] = R^T:
                //     S[i, j] <- S[i, j] - C_i P R[:, j]     S[i, K] <- C_i P     S[K, j] <- P R[:, j]     S[K, K] <- -P          (i, j outside K)
                // keeps S symmetric, and after every block has been swept S = -(S_0)^-1.  One update is ONE rank-4 MFMA per 16 x 16 tile:
                //     D = base + W R'',   W_i = -C_i P (rows outside K),  W_q = P[q, :] (rows of K),   R'' = R with its K columns replaced by -I,
                //     base = S with the K columns and the K rows zeroed
                // -- v_mfma_f64_16x16x4_f64 IS a rank-4 update.  Lane l = (lg, lc) supplies the A operand W[strip row lc][lg] and the B operand R''[lg][16 J + lc]:
                // R (4 rows, published by the wave that owns them: accumulator register r0 of its tiles IS the B-operand layout) is read from LDS by every wave,
                // and C comes from the SAME buffer by symmetry (C_i[p] = R[p][i]) -- no transposition inside the wave.  Every lane inverts the 4 x 4 pivot block
                // itself (no second barrier), as the blocked Gauss-Jordan did.  One barrier per block; rows / columns n .. 16 NTILE - 1 are padded with the identity.
                static_assert(8 * LDP <= 16 * NP, "two R buffers of four rows fit the exchange region");
                constexpr int MAXB = 4 * NTILE;                       // blocks of the padded matrix
                const int NB = (n + 3) >> 2;                          // blocks that hold a row < n
                if (wave < NTILE) {
                    static_for<NTILE>([&](auto Jc) {                  // diagonal: + rho_x (rows < n), identity (padding rows)
                        constexpr int J = decltype(Jc)::value;
                        if (wave == J) {
#pragma unroll
                            for (int r = 0; r < 4; r++) { const int row = 16 * J + lg + 4 * r; if (lc == lg + 4 * r) acc[J][r] += (row < n ? rho_x : 1.0); }
                        }
                    });
                    if (wave == 0) {
#pragma unroll
                        for (int J = 0; J < NTILE; J++) Gm[lg * LDP + 16 * J + lc] = acc[J][0];          // rows 0 .. 3 -> buffer 0
                    }
                }
                __syncthreads();
                F2_STAMP(8);
                F2_T0();
                static_for<MAXB>([&](auto bc) {
                    constexpr int b = decltype(bc)::value, k0 = 4 * b, wo = k0 / 16, r0 = (k0 % 16) / 4, c0 = k0 % 16;
                    constexpr int k1 = k0 + 4, wn = (k1 / 16) % NTILE, r1 = (k1 % 16) / 4;          // owner wave / register of the NEXT block's rows
                    if (b < NB) {                                     // (uniform)
                        const double *Rb = Gm + (b & 1) * (4 * LDP);
                        double *Rn = Gm + ((b + 1) & 1) * (4 * LDP);
                        if (wave < NTILE) {
                            double Bop[NTILE], Cop[4], a[4][4];
#pragma unroll
                            for (int J = 0; J < NTILE; J++) Bop[J] = Rb[lg * LDP + 16 * J + lc];
                            {
                                int coff = 16 * wave + lc;            // (opaque: keeps the four addresses from being hoisted out of the unrolled blocks)
                                asm volatile("" : "+v"(coff));
#pragma unroll
                                for (int pq = 0; pq < 4; pq++) Cop[pq] = Rb[pq * LDP + coff];
                            }
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const double2 v0 = *reinterpret_cast<const double2 *>(Rb + q * LDP + k0), v1 = *reinterpret_cast<const double2 *>(Rb + q * LDP + k0 + 2);
                                a[q][0] = v0.x; a[q][1] = v0.y; a[q][2] = v1.x; a[q][3] = v1.y;
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            // Column lg of P = S[K, K]^-1, and nothing else: lane (lg, lc) feeds the matrix cores W[strip row lc][lg] = -C_i . P[:, lg], so it solves
                            // S[K, K] xs = e_lg (elimination without pivoting -- the swept block is positive definite; the matrix is the same in every lane, the right-hand
                            // side is the lane's) instead of inverting the block and selecting a column: a third of the arithmetic, and no select chains (which the
                            // compiler turned into divergent branches, ~1.2 k cycles per block: profiles/r05/c_timing.log).  Every choice below is a BLEND with 0 / 1 lane
                            // constants for the same reason.
                            double e[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) e[k] = (lg == k) ? 1.0 : 0.0;
                            double pinv[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const double pv = a[k][k];
                                double pi_ = __builtin_amdgcn_rcp(pv);             // seed + two Newton steps instead of the IEEE divide
                                pi_ = fma(fma(-pv, pi_, 1.0), pi_, pi_);
                                pi_ = fma(fma(-pv, pi_, 1.0), pi_, pi_);
                                pinv[k] = pi_;
#pragma unroll
                                for (int i = k + 1; i < 4; i++) {
                                    const double lm = a[i][k] * pi_;
#pragma unroll
                                    for (int j = k + 1; j < 4; j++) a[i][j] = fma(-lm, a[k][j], a[i][j]);
                                    e[i] = fma(-lm, e[k], e[i]);
                                }
                            }
                            double xs[4];
                            xs[3] = e[3] * pinv[3];
                            xs[2] = fma(-a[2][3], xs[3], e[2]) * pinv[2];
                            xs[1] = fma(-a[1][3], xs[3], fma(-a[1][2], xs[2], e[1])) * pinv[1];
                            xs[0] = fma(-a[0][3], xs[3], fma(-a[0][2], xs[2], fma(-a[0][1], xs[1], e[0]))) * pinv[0];
                            F2_EACC(4);      // operand reads + the 4 x 4 solve
                            double wsel = -(Cop[0] * xs[0] + Cop[1] * xs[1] + Cop[2] * xs[2] + Cop[3] * xs[3]);
                            const int mr = lc - c0;                   // this lane's strip row, relative to the block
                            const bool kcol = mr >= 0 && mr < 4;
                            const double nk = kcol ? 0.0 : 1.0;
                            if (wave == wo) {                         // the strip that holds the rows of K: W_q = P[q, :] (= xs[q] by symmetry), base 0
                                double psel = 0.0;
#pragma unroll
                                for (int k = 0; k < 4; k++) psel = fma(xs[k], (mr == k) ? 1.0 : 0.0, psel);
                                wsel = fma(wsel, nk, psel);
#pragma unroll
                                for (int J = 0; J < NTILE; J++) acc[J][r0] = 0.0;
                            }
#pragma unroll
                            for (int r = 0; r < 4; r++) acc[wo][r] *= nk;                                  // the K columns of the base (tile J0 = wo)
                            Bop[wo] = fma(Bop[wo], nk, (kcol && mr == lg) ? -1.0 : 0.0);                   // ... and of R'': -I
                            F2_EACC(5);      // multipliers
#pragma unroll
                            for (int J = 0; J < NTILE; J++) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, Bop[J], acc[J], 0, 0, 0);
                            F2_EACC(6);      // rank-4 update
                            if (b + 1 < NB && wave == wn) {
#pragma unroll
                                for (int J = 0; J < NTILE; J++) Rn[lg * LDP + 16 * J + lc] = acc[J][r1];
                            }
                        }
                        __syncthreads();
                        F2_EACC(7);      // publish the next rows + the barrier
                    }
                });
                F2_STAMP(9);
                // G = -(swept S) to LDS, row-major with pitch ldg (the R buffers are dead: every block ended with a barrier)
                if (wave < NTILE) {
#pragma unroll
                    for (int J = 0; J < NTILE; J++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = 16 * wave + lg + 4 * r, col = 16 * J + lc;
                            if (row < n && col < L::NPg) Gm[row * ldg + col] = 0.0 - acc[J][r];
                        }
                }
                g_in_lds = true;
            } else {
                // S (rows and columns < n) to LDS, row-major with pitch ldg, then into the (jg, cg) register tile of the inversion
                if (wave < NTILE) {
#pragma unroll
                    for (int J = 0; J < NTILE; J++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int row = 16 * wave + lg + 4 * r, col = 16 * J + lc;
                            if (row < n && col < L::NPg) Gm[row * ldg + col] = acc[J][r] + (row == col ? rho_x : 0.0);
                        }
                }
                __syncthreads();
                if (jg < n) {
                    const double2 *src = reinterpret_cast<const double2 *>(Gm + jg * ldg + TG * cg);
#pragma unroll
                    for (int s = 0; s < TG / 2; s++) { const double2 v = src[s]; sreg[2 * s] = v.x; sreg[2 * s + 1] = v.y; }
                }
                __syncthreads();
                if constexpr (HASP) {
                    double pg[TG];
                    materialize_p(co, pg);
#pragma unroll
                    for (int s = 0; s < TG; s++) if (jg < n) sreg[s] += pg[s];
                }
            }
        }
        if (!g_in_lds) {
        F2_STAMP(8);
        // BLOCKED Gauss-Jordan inversion on the register tile: four pivots per workgroup barrier.  Block order: slots kk0 = 0, 4, 8, ...
        // (static), inside a slot block the lane groups cgk = 0, 1, ... that still hold a pivot < n; block K = columns / rows
        // k0 .. k0 + 3, k0 = TG cgk + kk0.  With C = S[:, K], R = S[K, :] (published through LDS, double buffered in the idle G
        // region) and P = S[K, K]^-1 (4 x 4, every thread inverts it itself: no second barrier):
        //     rows outside K :  S[i, :] += w R,  w = -C[i, :] P,   S[i, K] = w        rows in K :  S[q, :] = P[q, :] R,  S[q, K] = P[q, :]
        // A block that runs past n (or past the slot count TG) is padded with the identity.  One barrier per block instead of one
        // per pivot (round 1: 50 barriers, 72 k cycles of a 183 k cycle refactor at the metric configuration).
        constexpr int NBLK = (TG + 3) / 4;
        auto rbuf = [&](int b) -> double * { return Gm + b * (8 * NP); };              // 4 rows of NP
        auto cbuf = [&](int b) -> double * { return Gm + b * (8 * NP) + 4 * NP; };     // NP rows of 4
        auto publish = [&](auto blk_c, int cgn, int bufn) {
            constexpr int kk0 = 4 * decltype(blk_c)::value;
            const int k0 = TG * cgn + kk0;
            if (jg < n && cg == cgn) {           // (TG is even: a block has 4 or 2 slots; the missing pair is treated as zero by the reader)
                double2 *dst = reinterpret_cast<double2 *>(cbuf(bufn) + 4 * jg);
                dst[0] = make_double2(sreg[kk0], sreg[kk0 + 1]);
                if constexpr (kk0 + 3 < TG) dst[1] = make_double2(sreg[kk0 + 2], sreg[kk0 + 3]);
            }
            if (jg >= k0 && jg < k0 + (TG - kk0 < 4 ? TG - kk0 : 4) && jg < n) {
                double2 *dst = reinterpret_cast<double2 *>(rbuf(bufn) + (jg - k0) * NP + TG * cg);
#pragma unroll
                for (int s = 0; s < TG / 2; s++) dst[s] = make_double2(sreg[2 * s], sreg[2 * s + 1]);
            }
        };
        for (int i = tid; i < 16 * NP; i += NT) Gm[i] = 0.0;          // stale panel data out of the exchange buffers (padding rows are read)
        __syncthreads();
        int cnt = 0;
        publish(std::integral_constant<int, 0>{}, 0, 0);
        __syncthreads();
        F2_T0();
        static_for<NBLK>([&](auto blk_c) {
            constexpr int blk = decltype(blk_c)::value, kk0 = 4 * blk;
            constexpr int NBS = (TG - kk0) < 4 ? (TG - kk0) : 4;                         // slots of this block that exist
            const int nv = (kk0 < n) ? (n - 1 - kk0) / TG + 1 : 0;                       // lane groups with a pivot in this block
            const int nvn = (kk0 + 4 < TG && kk0 + 4 < n) ? 1 : 0;                       // does the next block have one?
            for (int cgk = 0; cgk < nv; cgk++) {
                const int k0 = TG * cgk + kk0, buf = cnt & 1;
                const int nbv = min(NBS, n - k0);                                        // pivots of this block (the rest: identity)
                const double *rb = rbuf(buf), *cb = cbuf(buf);
                if (jg < n) {
                    // the 4 x 4 pivot block: two 16-byte reads per row (k0 is even, the rows are 16-byte aligned; a block that runs past n reads into the next row / the column panel,
                    // which the fix-up below overwrites).  The guarded element-wise form was 16 scalar branches around 16 ds_read_b64 at the head of every block's dependency chain.
                    double a[4][4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const double2 v0 = *reinterpret_cast<const double2 *>(rb + q * NP + k0), v1 = *reinterpret_cast<const double2 *>(rb + q * NP + k0 + 2);
                        a[q][0] = v0.x; a[q][1] = v0.y; a[q][2] = v1.x; a[q][3] = v1.y;
                    }
                    if (nbv < 4) {      // (uniform, at most twice per inversion) pad with the identity
#pragma unroll
                        for (int q = 0; q < 4; q++)
#pragma unroll
                            for (int q2 = 0; q2 < 4; q2++) if (!(q < nbv && q2 < nbv)) a[q][q2] = (q == q2 ? 1.0 : 0.0);
                    }
                    bool bad = false;
#pragma unroll
                    for (int p = 0; p < 4; p++) {      // in-place inverse of the 4 x 4 block (no pivoting: S is positive definite)
                        const double pv = a[p][p];
                        bad = bad || !(pv > 0);
                        double pinv = __builtin_amdgcn_rcp(pv);                 // seed + two Newton steps instead of the IEEE divide
                        pinv = fma(fma(-pv, pinv, 1.0), pinv, pinv);
                        pinv = fma(fma(-pv, pinv, 1.0), pinv, pinv);
#pragma unroll
                        for (int j = 0; j < 4; j++) if (j != p) a[p][j] *= pinv;
#pragma unroll
                        for (int i = 0; i < 4; i++) if (i != p) {
                            const double f = a[i][p];
#pragma unroll
                            for (int j = 0; j < 4; j++) if (j != p) a[i][j] = fma(-f, a[p][j], a[i][j]);
                            a[i][p] = -f * pinv;
                        }
                        a[p][p] = pinv;
                    }
                    F2_EACC(4);      // pivot block read + its 4 x 4 inverse
                    if constexpr (HASP) { if (bad && jg == k0 && cg == 0) sc[7] = 1.0; }     // S not positive definite: P is not PSD
                    const double2 c01 = reinterpret_cast<const double2 *>(cb + 4 * jg)[0];
                    double2 c23 = make_double2(0.0, 0.0);
                    if constexpr (NBS > 2) c23 = reinterpret_cast<const double2 *>(cb + 4 * jg)[1];
                    const int qrow = jg - k0;
                    const bool prow_thread = (qrow >= 0 && qrow < NBS);
                    double w[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const double wc = -(c01.x * a[0][q] + c01.y * a[1][q] + c23.x * a[2][q] + c23.y * a[3][q]);
                        const double wp = qrow == 0 ? a[0][q] : (qrow == 1 ? a[1][q] : (qrow == 2 ? a[2][q] : a[3][q]));
                        w[q] = prow_thread ? wp : wc;
                    }
                    F2_EACC(5);      // multipliers
                    const double2 *r0 = reinterpret_cast<const double2 *>(rb + TG * cg), *r1 = reinterpret_cast<const double2 *>(rb + NP + TG * cg),
                                  *r2 = reinterpret_cast<const double2 *>(rb + 2 * NP + TG * cg), *r3 = reinterpret_cast<const double2 *>(rb + 3 * NP + TG * cg);
#pragma unroll
                    for (int s2 = 0; s2 < TG / 2; s2++) {
                        const double2 v0 = r0[s2], v1 = r1[s2], v2 = r2[s2], v3 = r3[s2];
                        const double bx = prow_thread ? 0.0 : sreg[2 * s2], by = prow_thread ? 0.0 : sreg[2 * s2 + 1];
                        sreg[2 * s2] = fma(w[3], v3.x, fma(w[2], v2.x, fma(w[1], v1.x, fma(w[0], v0.x, bx))));
                        sreg[2 * s2 + 1] = fma(w[3], v3.y, fma(w[2], v2.y, fma(w[1], v1.y, fma(w[0], v0.y, by))));
                    }
                    if (cg == cgk) {
#pragma unroll
                        for (int q = 0; q < NBS; q++) sreg[kk0 + q] = w[q];       // the block columns themselves
                    }
                }
                F2_EACC(6);      // rank-4 update of the tile
                if (cgk + 1 < nv) publish(std::integral_constant<int, blk>{}, cgk + 1, buf ^ 1);
                else if (nvn) publish(std::integral_constant<int, (blk + 1 < NBLK ? blk + 1 : blk)>{}, 0, buf ^ 1);
                cnt++;
                __syncthreads();
                F2_EACC(7);      // publish the next panel + the barrier
            }
        });
        F2_STAMP(9);
        }      // (!g_in_lds)
        }      // (!fast)
        g_scale = uniform_d(scale);
        if constexpr (HASP) { if (sc[7] != 0.0) return; }      // (uniform: read after the loop's last barrier) -> status FAILED below
        // G to LDS (the panel data in that region is dead), scratch back to zero
        if (!g_in_lds && jg < n) {
            double2 *dst = reinterpret_cast<double2 *>(Gm + jg * ldg + TG * cg);
#pragma unroll
            for (int s = 0; s < TG / 2; s++) dst[s] = make_double2(sreg[2 * s], sreg[2 * s + 1]);
        }
        for (int i = tid; i < NP; i += NT) { sm[L::O_S1 + i] = 0.0; sm[L::O_S2 + i] = 0.0; sm[L::O_S3 + i] = 0.0; sm[L::O_S4 + i] = 0.0; }
        for (int i = tid; i < m; i += NT) sm[L::O_ZB + OY + i] = dyv(i) * sm[L::O_BV + i];
        __syncthreads();
        materialize_at(co);
        {
            const double a = seg_dot<CHT, T1>(at, sm + L::O_ZB + OY + T1 * c1);
            if (own1) { const double cj = sm[L::O_CV + j1]; sm[L::O_S1 + j1] = cj - a; sm[L::O_S2 + j1] = cj + a; }   // rhs for g_x ; k = c + A^T Dy b
        }
        __syncthreads();
        F2_STAMP(11);
        {
            const double *grow = Gm + (jg < n ? jg : 0) * ldg + TG * cg;
            const double gx = seg_dot_lds<CHG, TG>(grow, sm + L::O_S1 + TG * cg), gk = seg_dot_lds<CHG, TG>(grow, sm + L::O_S2 + TG * cg);
            if (owng) { sm[L::O_GV + OX + jg] = gx; sm[L::O_PX + jg] = gk; }
        }
        __syncthreads();
        F2_STAMP(12);
        if constexpr (HASP) {      // P-hat g_x (vector, kept) and g_x^T P-hat g_x
            double pg[TG];
            materialize_p(co, pg);
            const double a = seg_dot<CHG, TG>(pg, sm + L::O_GV + OX + TG * cg);
            if (owng) PgV[jg] = a;
            __syncthreads();
            double rg[1] = {tid < n ? sm[L::O_GV + OX + tid] * PgV[tid] : 0.0};
            block_reduce_w<1, NW>(rg, 0u, red, wave);
            gPg = uniform_d(rg[0]);
        }
        materialize_ar(co);
        double r[1] = {0};
        {
            const double agx = seg_dot<CHA, T2>(ar, sm + L::O_GV + OX + T2 * c2);
            __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler interleaves the two products one read at a time: 20 serialised LDS round trips)
            const double agk = seg_dot<CHA, T2>(ar, sm + L::O_PX + T2 * c2);
            __builtin_amdgcn_sched_barrier(0);
            if (own2) {
                const double bi = sm[L::O_BV + i2];
                const double gy = dyv(i2) * (agx + bi);
                sm[L::O_GV + OY + i2] = gy; r[0] += bi * gy;
                sm[L::O_PHI + OY + i2] = bi - agk;
            }
            if (tid < n) { r[0] += sm[L::O_CV + tid] * sm[L::O_GV + OX + tid]; sm[L::O_PHI + OX + tid] = rho_x * sm[L::O_PX + tid]; }
        }
        F2_STAMP(13);
        block_reduce_w<1, NW>(r, 0u, red, wave);
        hg = uniform_d(r[0]);
        inv_den = uniform_d(1.0 / (rtau + hg));
        load_phi_tile(co);
        for (int i = tid; i < NP; i += NT) { sm[L::O_S1 + i] = 0.0; sm[L::O_S2 + i] = 0.0; sm[L::O_PX + i] = 0.0; }
        for (int i = tid; i < m; i += NT) sm[L::O_ZB + OY + i] = 0.0;
        __syncthreads();
        F2_STAMP(10);
    };

    if (threadIdx.x == 0) sm[L::O_W + OT] = 1.0;    // cold start: w = (0, 0, 1)
    if (S.warm_start) {
        // warm start from the caller's (x, y, s): u = (x^, y^, 1), v = (0, s^, 0) in the equilibrated space, and the fixed point
        // of the iteration map has w = u + R^-1 v.   x^ = sigma x / E, y^ = sigma y / D, s^ = sigma D s.
        const double sg = sc[SC_SIGMA];
        const int inst_ = inst;
        double wx = 0, wy = 0; bool bad = false;
        const int e = threadIdx.x;
        if (e < n) { wx = sg * xo[(size_t)inst_ * n + e] / sm[L::O_EV + e]; bad = !(fabs(wx) < 1e300); }
        for (int i = e; i < m; i += NT) {
            const double dvi = sm[L::O_DV + i];
            const int io = WL ? row_perm[i] : i;                 // kernel row -> template row
            const double v = sg * yo[(size_t)inst_ * m + io] / dvi + sg * dvi * so[(size_t)inst_ * m + io] * dyv(i);
            bad = bad || !(fabs(v) < 1e300);
        }
        double rb[1] = {bad ? 1.0 : 0.0};
        block_reduce_w<1, NW>(rb, 1u, red, wave);          // (max over the workgroup; __syncthreads_or would add static LDS)
        if (rb[0] == 0.0) {
            if (e < n) sm[L::O_W + OX + e] = wx;
            for (int i = e; i < m; i += NT) {
                const double dvi = sm[L::O_DV + i];
                const int io = WL ? row_perm[i] : i;
                sm[L::O_W + OY + i] = sg * yo[(size_t)inst_ * m + io] / dvi + sg * dvi * so[(size_t)inst_ * m + io] * dyv(i);
            }
        }
    }
    __syncthreads();

    int status = 0, iter = 0, last_scale_iter = 0, n_log = 0;
    // iter % acceleration_interval and iter % CONVERGED_INTERVAL as counters: the remainders of a run-time divisor were ~35 scalar instructions at the top of EVERY
    // iteration (two of them: `iter` and `iter + 1`), on a path where one wave issues one instruction per four cycles
    int aa_ph = 0, chk_ph = 0;
    auto next_iter = [&]() { iter++; aa_ph = (aa_ph + 1 >= S.acceleration_interval) ? 0 : aa_ph + 1; chk_ph = (chk_ph + 1 == CONVERGED_INTERVAL) ? 0 : chk_ph + 1; };
    // Anderson acceleration of the iteration map w -> F(w) (type I, one secant pair; oracle/cone_oracle.c with aa_mem = 1):
    // every aa_int iterations, with x = input and f = output of the last iteration, g = x - f, s = x - x_prev, y = g - g_prev,
    // d = f - f_prev:  w <- f - (s.g / (s.y + 1e-8 |s||y|)) d.   The next iteration's residual is the safeguard: if it exceeds |g| the
    // step is undone and the history dropped.  Vectors (VP doubles each) live in the dynamic tail of the LDS carve.
    bool aa_on = S.acceleration_lookback > 0;      // cleared after AA_MAX_REJECT safeguard rejections (robustness rule, see the oracle)
    const int aa_int = S.acceleration_interval;
    double *const aaWP = Gm + gsz + (PSD ? ((T.ns > 0 ? 2 * T.maxs * T.maxs + 2 * T.maxs + 8 : 0) + T.nep + T.np) : 0) + (HASP ? NP : 0);
    double *const aaXP = aaWP + VP, *const aaFP = aaXP + VP, *const aaFS = aaFP + VP, *const aaXS = aaFS + VP;
    int aa_iter = 0; bool aa_pending = false, aa_stale = false;      // (|g|^2 before the step lives in sc[8]: no register across the loop)
    const bool big_soc = T.maxq > SOC_SMALL;
    bool resume = false;     // true: the iteration interrupted by a rescale still owes its relaxed update

    // cone projection of element e (LDS slot ve) from ZB (pre-projection values); small cones: recomputed by every row thread
    auto project_e = [&](int e, int ve) -> double {
        double ue = sm[L::O_ZB + ve];
        const int soc_d = (e < m) ? socd[e] : 0;
        if (soc_d > 1 && !big_soc) {
            const int soc_r0 = socr[e];
            const double *zc = sm + L::O_ZB + OY + soc_r0;
            const double t0 = zc[0];
            double q0 = 0, q1 = 0;
            if (soc_d <= 13) {       // one batch of reads, masked (the reads past the cone stay inside the vector, pads included)
                double zv[12];
#pragma unroll
                for (int u = 0; u < 12; u++) zv[u] = zc[1 + u];
#pragma unroll
                for (int u = 0; u < 12; u += 2) {
                    const double z0 = (1 + u < soc_d) ? zv[u] : 0.0, z1 = (2 + u < soc_d) ? zv[u + 1] : 0.0;
                    q0 = fma(z0, z0, q0); q1 = fma(z1, z1, q1);
                }
            } else {
                for (int k = 1; k < soc_d; k += 4) {
                    const double z0 = zc[k], z1 = (k + 1 < soc_d) ? zc[k + 1] : 0.0, z2 = (k + 2 < soc_d) ? zc[k + 2] : 0.0, z3 = (k + 3 < soc_d) ? zc[k + 3] : 0.0;
                    q0 = fma(z0, z0, q0); q1 = fma(z1, z1, q1); q0 = fma(z2, z2, q0); q1 = fma(z3, z3, q1);
                }
            }
            const double q = q0 + q1;
            double nz = 0, rinv = 0;
            if (q > 0) sqrt_rsqrt(q, nz, rinv);
            if (nz <= t0) { /* inside */ }
            else if (nz <= -t0) ue = 0.0;
            else { const double c0 = 0.5 * (t0 + nz); ue = (e == soc_r0) ? c0 : ue * (c0 * rinv); }
        } else if (soc_d == 1) ue = fmax(ue, 0.0);
        return ue;
    };
    auto slot_of = [&](int e) -> int { return (e < m) ? OY + e : (e < m + n ? OX + (e - m) : OT); };

    for (bool done = false; !done;) {
    refactor();
    if constexpr (HASP) { if (sc[7] != 0.0) { status = -4; break; } }     // SCS_FAILED: the factorisation met a non-positive pivot
    F2_STAMP(3);
    if (resume) {   // relaxed update w += alpha (u - ut) owed by the iteration a rescale interrupted
        const int e = Co::thread_id(wave);
        if (e < l) { const int ve = slot_of(e); sm[L::O_W + ve] += alpha * (sm[L::O_U + ve] - sm[L::O_UT + ve]); }
        __syncthreads();
        resume = false; next_iter();
    }
    for (;;) {
        if (iter >= S.max_iters) { done = true; break; }
        const Co co(wave);
        const int j1 = co.j1, c1 = co.c1, i2 = co.i2, c2 = co.c2, jg = co.jg, cg = co.cg;
        const bool own1 = (c1 == 0) && (j1 < n), own2 = (c2 == 0) && (i2 < m), owng = (cg == 0) && (jg < n);
        const int e = co.t;
        const int ve = slot_of(e);
        const bool ev = e < l;
        const bool check = chk_ph == 0;
        const bool last = iter + 1 >= S.max_iters;
        F2_ACC(6);      // loop bookkeeping, thread coordinates
        if (aa_on) {      // (uniform)
            if (aa_pending) {      // safeguard: residual of the map at the accelerated point against the residual before the step
                const double dd = ev ? aaWP[ve] - sm[L::O_W + ve] : 0.0;
                double r[1] = {dd * dd};
                block_reduce_w<1, NW>(r, 0u, red, wave);
                if (!(uniform_d(r[0]) <= sc[8])) {      // (squared norms on both sides: the fp64 square root is a ~30-instruction dependent chain in every lane)
                    if (ev) { sm[L::O_W + ve] = aaFS[ve]; aaWP[ve] = aaXS[ve]; }
                    aa_iter = 0;
                    if (Co::thread_id(wave) == 0) sc[9] += 1.0;      // (the count lives in LDS: a register for it across the loop ends up in scratch)
                    __syncthreads();
                    if (uniform_d(sc[9]) >= AA_MAX_REJECT) aa_on = false;
                }
                aa_pending = false;
            }
            if (aa_on && iter > 0 && aa_ph == 0 && !aa_stale) {      // (aa_stale: the kept input predates a rescale -- with an interval that puts a step right behind a check iteration it would pair a pre-rescale input with a post-rescale output)
                const double xv = ev ? aaWP[ve] : 0.0, fv = ev ? sm[L::O_W + ve] : 0.0, gv = xv - fv;
                if (aa_iter > 0) {
                    const double xp = ev ? aaXP[ve] : 0.0, fp = ev ? aaFP[ve] : 0.0;
                    const double sv = xv - xp, yv = gv - (xp - fp), dv = fv - fp;
                    double r[5] = {sv * sv, yv * yv, sv * yv, sv * gv, gv * gv};
                    block_reduce_w<5, NW>(r, 0u, red, wave);
                    // scalar arithmetic of the step without the IEEE square-root / divide expansions (two sqrt + one divide were ~100 dependent VALU instructions in every lane,
                    // behind the reduction's barrier): hardware seeds + Newton / Goldschmidt steps (1-2 ulp; the 1e-8 term is a regulariser)
                    double s01 = 0, ri01 = 0;
                    { const double q01 = r[0] * r[1]; if (q01 > 0) sqrt_rsqrt(q01, s01, ri01); }
                    const double mm = uniform_d(r[2] + 1e-8 * s01);
                    double mi = __builtin_amdgcn_rcp(mm);
                    mi = fma(fma(-mm, mi, 1.0), mi, mi);
                    mi = fma(fma(-mm, mi, 1.0), mi, mi);
                    const double gam = uniform_d(r[3] * mi);
                    if (ev) { aaXP[ve] = xv; aaFP[ve] = fv; }
                    int o11 = 11; asm volatile("" : "+s"(o11));
                    if (fabs(mm) > sc[o11 + 1] /* 1e-300 */ && fabs(gam) < sc[o11] /* 1e10, from LDS: an fp64 literal is held in a VGPR pair across the loop */) {
                        if (ev) { aaFS[ve] = fv; aaXS[ve] = xv; sm[L::O_W + ve] = fv - gam * dv; }
                        if (Co::thread_id(wave) == 0) sc[8] = r[4];      // |g|^2 before the step (the safeguard compares squares)
                        aa_pending = true;
                    } else aa_iter = 0;
                } else if (ev) { aaXP[ve] = xv; aaFP[ve] = fv; }
                aa_iter++;
                __syncthreads();
            }
        }
        if (check && iter > 0) {   // keep the homogeneous iterate in range
            const double we = ev ? sm[L::O_W + ve] : 0.0;
            double r[1] = {we * we};
            block_reduce_w<1, NW>(r, 0u, red, wave);
            const double nw = uniform_d(sqrt(r[0]));
            if (nw > 0 && ev) {
                int o10 = 10; asm volatile("" : "+s"(o10));      // sqrt(l) from LDS through an opaque index: as a loop-invariant VALUE it is held in (and spilled from) a VGPR pair across the loop
                const double fsc = sc[o10] / nw;
                sm[L::O_W + ve] = we * fsc;
                if (aa_on) {
                    aaXP[ve] *= fsc; aaFP[ve] *= fsc; aaFS[ve] *= fsc; aaXS[ve] *= fsc;     // the map is positively homogeneous
                    if (e == 0) sc[8] *= fsc * fsc;                                        // (sc[8] is a squared norm)
                }
            }
            __syncthreads();
        }
        if (aa_on && (aa_pending || aa_ph + 1 == aa_int)) { aa_stale = false; if (ev) aaWP[ve] = sm[L::O_W + ve]; }      // input of this iteration, kept where the top of the next one reads it (the safeguard after a step, the step itself every aa_int iterations)
        F2_ACC(0);      // top of the iteration (acceleration bookkeeping, coordinates)
        // P1a: t = rho_x w_x - A^T w_y   (+ phi . w from the two spare column groups)
        {
            const double *wvec = sm + L::O_W + ((j1 == n + 1 && T1 * c1 < n) ? OX : OY) + T1 * c1;
            const double a = seg_dot<CHT, T1>(at, wvec);
            if (own1) sm[L::O_TV + j1] = rho_x * sm[L::O_W + OX + j1] - a;
            else if (c1 == 0 && j1 <= n + 1) sm[L::O_WP + (j1 - n)] = a;          // phi_y . w_y , phi_x . w_x
        }
        F2_ACC(1);      // P1a up to its barrier
        __syncthreads();
        F2_ACC(2);      // the barrier
        // P1b: p_x = G t
        {
            const double a = seg_dot_lds<CHG, TG>(Gm + __mul24(jg < n ? jg : 0, ldg) + TG * cg, sm + L::O_TV + TG * cg);      // (24-bit multiply: full rate)
            if (owng) sm[L::O_PX + jg] = a;
            if (WL && Co::thread_id(wave) == 0) sm[L::O_WP + 2] = sm[L::O_W + OT];   // snapshot of w_tau: the fused phase below rewrites it while other waves still need it
        }
        F2_ACC(3);      // P1b up to its barrier
        __syncthreads();
        F2_ACC(2);
        if constexpr (WL) {
            // P2 + P3 fused: q = A p_x ; tau-tilde ; u-tilde ; cone projection ; relaxed update.  The cone blocks of y are wave-local.
            // Latency is what this phase costs (three workgroups per CU hide some of it, not all): every LDS round trip that can be issued early is.
            // (1) operands that do not depend on this phase's product are requested BEFORE it and arrive while it runs;
            // (2) the product's own 13-read stream comes next, fenced, so that the scheduler pipelines it (left alone it serialised the reads: ten round trips);
            // (3) the cone's entries are fetched with one unrolled batch of reads (cones of <= 13 rows) instead of a loop of dependent four-packs.
            const int ee = OY + i2;
            const bool upd = !check && !last;          // fast path: the relaxed update happens here (else after the convergence check)
            double we = 0, gve = 0;
            int cd = 0, soc_r0 = 0;
            if (own2) { we = sm[L::O_W + ee]; gve = sm[L::O_GV + ee]; }
            if (i2 < m) { cd = socd[i2]; soc_r0 = socr[i2]; }      // cd 0: zero-cone row (dual free), 1: nonnegative row, > 1: row of an SOC.  (Every lane of the row: they share the norm's work below)
            const double wp0 = sm[L::O_WP], wp1 = sm[L::O_WP + 1], wp2 = sm[L::O_WP + 2];
            const double q = seg_dot<CHA, T2>(ar, sm + L::O_PX + T2 * c2);
            __builtin_amdgcn_sched_barrier(0);
            const double tau_t = (rtau * wp2 + wp0 + wp1) * inv_den;
            double ute = 0, ze = 0;
            if (own2) {
                const double py = we + dyv(i2) * q;
                ute = py - tau_t * gve;
                ze = 2 * ute - we;
                if (cd == 1 && ze < 0) ze = 0;
                sm[L::O_UT + ee] = ute; sm[L::O_ZB + ee] = ze;
            }
            F2_ACC(4);      // fused phase: product, tau, cone input
            wave_lds_exchange();
            // |tail|^2 of a cone of <= 13 rows: the CHA lanes of a row each sum NE of the 12 candidate entries and the shares are added with a DPP butterfly (every row used to
            // sum all 12 itself: 12 reads, 12 compares, 24 selects and 12 FMAs per lane and iteration on a kernel that is bound by the instructions it issues)
            constexpr int NE = (12 + CHA - 1) / CHA;
            double qsh = 0;
            if (cd > 1 && cd <= 13) {      // (the reads past the cone stay inside the vector -- its pads included -- and are masked)
                const int off = (int)__umul24((unsigned)c2, (unsigned)NE);      // (v_mul_u32_u24: full rate; the 32-bit multiply is a quarter-rate instruction)
                const double *zc = sm + L::O_ZB + OY + soc_r0 + 1 + off;
                const int lim = cd - 1 - off;            // this lane's entries u < lim belong to the cone
                double zv[NE];
#pragma unroll
                for (int u = 0; u < NE; u++) zv[u] = zc[u];
                double q0 = 0, q1 = 0;
#pragma unroll
                for (int u = 0; u < NE; u++) {
                    const double zz = (u < lim) ? zv[u] : 0.0;
                    if (u & 1) q1 = fma(zz, zz, q1); else q0 = fma(zz, zz, q0);
                }
                qsh = q0 + q1;
            }
            qsh = group_reduce<CHA, false>(qsh);
            if (own2) {
                double ue = ze;
                if (cd > 1) {
                    const double *zc = sm + L::O_ZB + OY + soc_r0;
                    const double t0 = zc[0];
                    double q0 = qsh, q1 = 0;
                    if (cd <= 13) {
                    } else {
                        for (int k = 1; k < cd; k += 4) {
                            const double z0 = zc[k], z1 = (k + 1 < cd) ? zc[k + 1] : 0.0, z2 = (k + 2 < cd) ? zc[k + 2] : 0.0, z3 = (k + 3 < cd) ? zc[k + 3] : 0.0;
                            q0 = fma(z0, z0, q0); q1 = fma(z1, z1, q1); q0 = fma(z2, z2, q0); q1 = fma(z3, z3, q1);
                        }
                    }
                    const double qq = q0 + q1;
                    double nz = 0, rinv = 0;
                    if (qq > 0) sqrt_rsqrt(qq, nz, rinv);
                    if (nz <= t0) { /* inside */ }
                    else if (nz <= -t0) ue = 0.0;
                    else { const double c0 = 0.5 * (t0 + nz); ue = (i2 == soc_r0) ? c0 : ue * (c0 * rinv); }
                }
                sm[L::O_U + ee] = ue;
                if (upd) sm[L::O_W + ee] = we + alpha * (ue - ute);
            }
            if (e < n) {
                const int ex = OX + e;
                const double wx = sm[L::O_W + ex];
                const double utx = sm[L::O_PX + e] - tau_t * sm[L::O_GV + ex];
                const double ux = 2 * utx - wx;
                sm[L::O_UT + ex] = utx; sm[L::O_U + ex] = ux;
                if (upd) sm[L::O_W + ex] = wx + alpha * (ux - utx);
            }
            if (e == NT - 1) {
                const double wt = sm[L::O_W + OT];
                const double ut = fmax(0.0, 2 * tau_t - wt);
                sm[L::O_UT + OT] = tau_t; sm[L::O_U + OT] = ut;
                if (upd) sm[L::O_W + OT] = wt + alpha * (ut - tau_t);
            }
            F2_ACC(5);      // fused phase: projection, relaxed update
            __syncthreads();
            F2_ACC(2);
            if (upd) { next_iter(); continue; }
        } else {
        // P2: q = A p_x ; tau-tilde ; u-tilde ; cone input
        {
            // (operands that do not depend on the product are requested before it, the product's stream is fenced: see the wave-local variant above)
            const double wt0 = sm[L::O_W + OT], wp0 = sm[L::O_WP], wp1 = sm[L::O_WP + 1];
            const double q = seg_dot<CHA, T2>(ar, sm + L::O_PX + T2 * c2);
            __builtin_amdgcn_sched_barrier(0);
            double tau_t = (rtau * wt0 + wp0 + wp1) * inv_den;
            if constexpr (HASP) {
                // positive root of (r_tau + h.g - g^T P g) t^2 + (-(r_tau w_tau + h.p) + 2 p^T P g) t - p^T P p = 0, with
                // p^T P p = rho_x p_x.(w_x - p_x) - (A p_x).p_y  (first block row of the linear system: no extra product)
                double r3[3] = {0, 0, 0};
                if (own2) { const double py = sm[L::O_W + OY + i2] + dyv(i2) * q; r3[0] = q * py; }
                if (e < n) { const double px = sm[L::O_PX + e]; r3[1] = px * (sm[L::O_W + OX + e] - px); r3[2] = px * PgV[e]; }
                block_reduce_w<3, NW>(r3, 0u, red, wave);
                const double pPp = rho_x * r3[1] - r3[0];
                const double qa = rtau + hg - gPg, qb = -(rtau * sm[L::O_W + OT] + sm[L::O_WP] + sm[L::O_WP + 1]) + 2 * r3[2];
                tau_t = uniform_d((-qb + sqrt(fmax(qb * qb + 4 * qa * fmax(pPp, 0.0), 0.0))) / (2 * qa));
            }
            if (own2) {
                const int ee = OY + i2;
                const double we = sm[L::O_W + ee];
                const double py = we + dyv(i2) * q;
                const double ute = py - tau_t * sm[L::O_GV + ee];
                double ze = 2 * ute - we;
                if (i2 >= z && i2 < z + T.l && ze < 0) ze = 0;        // nonnegative rows
                sm[L::O_UT + ee] = ute; sm[L::O_ZB + ee] = ze;
            }
            if (e < n) {
                const int ee = OX + e;
                const double ute = sm[L::O_PX + e] - tau_t * sm[L::O_GV + ee];
                sm[L::O_UT + ee] = ute; sm[L::O_ZB + ee] = 2 * ute - sm[L::O_W + ee];
            }
            if (e == NT - 1) { sm[L::O_UT + OT] = tau_t; sm[L::O_ZB + OT] = fmax(0.0, 2 * tau_t - sm[L::O_W + OT]); }
        }
        __syncthreads();
        if (big_soc) {   // large cones: one leader per cone computes (c0, f) -> S3/S4, then rows apply (uniform branch)
            for (int c = e; c < nq; c += NT) {
                const int r0 = OY + T.qoff[c], r1 = OY + T.qoff[c + 1];
                const double t0 = sm[L::O_ZB + r0]; double nz = 0;
                for (int k = r0 + 1; k < r1; k++) nz = fma(sm[L::O_ZB + k], sm[L::O_ZB + k], nz);
                nz = sqrt(nz);
                double c0, f;
                if (r1 - r0 == 1) { c0 = fmax(t0, 0.0); f = 0.0; }
                else if (nz <= t0) { c0 = t0; f = 1.0; }
                else if (nz <= -t0) { c0 = 0.0; f = 0.0; }
                else { c0 = 0.5 * (t0 + nz); f = c0 / nz; }
                sm[L::O_S3 + c] = c0; sm[L::O_S4 + c] = f;
            }
            __syncthreads();
            if (e < m && socd[e] > 0) { const int c = T.rowcone[e]; sm[L::O_ZB + ve] = (e == socr[e]) ? sm[L::O_S3 + c] : sm[L::O_S4 + c] * sm[L::O_ZB + ve]; }
            __syncthreads();
        }
        if constexpr (PSD) {   // PSD blocks of the cone input are projected in place (all threads, one cone after the other)
            double *psdS = Gm + gsz, *psdV = psdS + T.maxs * T.maxs, *psdC = psdV + T.maxs * T.maxs;
            for (int c = 0; c < T.ns; c++) psd_project<NTH>(sm + L::O_ZB + OY + T.soff[c], T.sord[c], psdS, psdV, psdC, red);
            if (T.nep + T.np > 0) {   // exponential / power cones: one thread per cone, root warm-started from the previous iteration (ce_expcone.h)
                double *expR = Gm + gsz + (T.ns > 0 ? 2 * T.maxs * T.maxs + 2 * T.maxs + 8 : 0);
                for (int c = Co::thread_id(wave); c < T.nep + T.np; c += NTH) {
                    double *zc = sm + L::O_ZB + OY + T.eoff + 3 * c;
                    if (c < T.nep) exp_project_dual(zc, expR + c); else pow_project_dual_of_entry(zc, T.pw[c - T.nep], expR + c);
                }
                __syncthreads();
            }
        }
        if (!check && !last) {
            // P3 (fast path): project, relaxed update
            if (ev) {
                const double ue = project_e(e, ve);
                sm[L::O_U + ve] = ue;
                sm[L::O_W + ve] += alpha * (ue - sm[L::O_UT + ve]);
            }
            __syncthreads();
            next_iter();
            continue;
        }
        // ---- slow path (every CONVERGED_INTERVAL iterations, and the last one)
        if (ev) sm[L::O_U + ve] = project_e(e, ve);
        __syncthreads();
        }      // (!WL)
        bool stop = false, rescale = false;
        if (check) {
            // The two products of the residual test are parked in LDS (ZB is free here) and the residuals are evaluated in a
            // separate elementwise phase: the tile-using code then has the register footprint of the iteration's own
            // products, and the check does not raise the loop's register peak.
            {
                const double ax_raw = seg_dot<CHA, T2>(ar, sm + L::O_U + OX + T2 * c2);       // A-hat x-hat
                if (own2) sm[L::O_ZB + OY + i2] = ax_raw;
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const double aty_raw = seg_dot<CHT, T1>(at, sm + L::O_U + OY + T1 * c1);      // A-hat^T y-hat
                if (own1) sm[L::O_ZB + OX + j1] = aty_raw;
            }
            if constexpr (HASP) {      // P-hat x-hat, parked in TV (free between the products of two iterations)
                __builtin_amdgcn_sched_barrier(0);
                double pg[TG];
                materialize_p(co, pg);
                const double px_raw = seg_dot<CHG, TG>(pg, sm + L::O_U + OX + TG * cg);
                if (owng) sm[L::O_TV + jg] = px_raw;
            }
            __syncthreads();
            const double tau = uniform_d(fabs(sm[L::O_U + OT]));
            const double isg = uniform_d(1.0 / sc[SC_SIGMA]);
            double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rp, nax, ns, naxs, rd, naty (max) ; ctx, bty (sum)
            double rP[2] = {0, 0};                    // |P x| (max) ; x^T P x (sum)
            if (e < m) {
                const int i = e;
                const double sc_ = isg / sm[L::O_DV + i];
                const double ax = sm[L::O_ZB + OY + i] * sc_;
                const double uy = sm[L::O_U + OY + i];
                const double sh = (uy + sm[L::O_W + OY + i] - 2 * sm[L::O_UT + OY + i]) / dyv(i) * sc_;
                const double bt = sm[L::O_BV + i] * tau * sc_;
                r[0] = fabs(ax + sh - bt); r[1] = fabs(ax); r[2] = fabs(sh); r[3] = fabs(ax + sh);
                r[7] = sm[L::O_BV + i] * uy * isg * isg;
            } else if (e < m + n) {
                const int j = e - m;
                const double sc_ = isg / sm[L::O_EV + j];
                const double aty = sm[L::O_ZB + OX + j] * sc_;
                const double cj = sm[L::O_CV + j];
                double pxj = 0;
                if constexpr (HASP) { pxj = sm[L::O_TV + j] * sc_; rP[0] = fabs(pxj); rP[1] = sm[L::O_TV + j] * sm[L::O_U + OX + j] * isg * isg; }
                r[4] = fabs(pxj + aty + cj * tau * sc_); r[5] = fabs(aty);
                r[6] = cj * sm[L::O_U + OX + j] * isg * isg;
            }
            block_reduce_w<8, NW>(r, 0x3Fu, red, wave);
            double nPx = 0, xPx = 0;
            if constexpr (HASP) { block_reduce_w<2, NW>(rP, 0x1u, red, wave); nPx = uniform_d(rP[0]); xPx = uniform_d(rP[1]); }
            // the reduced values are equal in every lane: move them to scalar registers so that the convergence logic below
            // is scalar code with uniform branches (and n_log / last_scale_iter / status stay scalar)
            const double rp = uniform_d(r[0]), nax = uniform_d(r[1]), ns = uniform_d(r[2]), naxs = uniform_d(r[3]), rd = uniform_d(r[4]),
                         naty = uniform_d(r[5]), ctx = uniform_d(r[6]), bty = uniform_d(r[7]);
            const double nrm_b0 = uniform_d(sc[SC_NB0]), nrm_c0 = uniform_d(sc[SC_NC0]);
            if (tau > 0) {
                const double xPxt = xPx / tau;            // (0 without a quadratic objective)
                const double res_pri = rp / tau, res_dual = rd / tau, gap = fabs(xPxt + ctx + bty) / tau;
                sc[SC_RP] = res_pri; sc[SC_RD] = res_dual; sc[SC_GAP] = gap;
                const double prl = fmax(fmax(nrm_b0 * tau, ns), nax) / tau, drl = fmax(fmax(nrm_c0 * tau, naty), nPx) / tau;
                const double grl = fmax(fmax(fabs(ctx), fabs(bty)), fabs(xPxt)) / tau;
                if (res_pri <= S.eps_abs + S.eps_rel * prl && res_dual <= S.eps_abs + S.eps_rel * drl &&
                    gap <= S.eps_abs + S.eps_rel * grl) { status = 1; stop = true; }
            }
            if (!stop && bty < 0 && naty / (-bty) <= S.eps_infeas) { status = -2; stop = true; }
            if (!stop && ctx < 0 && fmax(naxs, nPx) / (-ctx) <= S.eps_infeas) { status = -1; stop = true; }
            if (!stop && S.adaptive_scale && iter > 0) {
                const double dp = fmax(fmax(nax, ns), nrm_b0 * tau), dd = fmax(fmax(naty, nrm_c0 * tau), nPx);
                const double rel_p = rp / (dp > 0 ? dp : 1), rel_d = rd / (dd > 0 ? dd : 1);
                if (rel_p > 0 && rel_d > 0 && isfinite(rel_p) && isfinite(rel_d)) {
                    const double sum_log = uniform_d(sc[SC_SUMLOG]) + ce_log(rel_p, mtab) - ce_log(rel_d, mtab); n_log++;
                    __syncthreads();                 // everyone has read SC_SUMLOG before it is rewritten
                    sc[SC_SUMLOG] = sum_log;
                    const double factor = ce_exp(0.5 * sum_log / n_log, mtab);          // sqrt(exp(sum_log / n_log))
                    if (iter - last_scale_iter >= RESCALING_MIN_ITERS) {
                        const double ns2 = fmin(fmax(scale * factor, MIN_SCALE_VALUE), MAX_SCALE_VALUE);
                        if (ns2 != scale && (factor > sqrt(10.0) || factor < 1.0 / sqrt(10.0))) {
                            // keep (s, kappa):  w_y+ = rsk_y / r_y+ + 2 ut_y - u_y
                            const double dy_ratio = ns2 / scale;
                            if (e < m) {
                                const double ue = sm[L::O_U + ve], ute = sm[L::O_UT + ve];
                                const double d0 = ue + sm[L::O_W + ve] - 2 * ute;
                                sm[L::O_W + ve] = d0 * dy_ratio + 2 * ute - ue;
                            }
                            n_log = 0; last_scale_iter = iter; scale = uniform_d(ns2); aa_iter = 0; aa_pending = false; aa_stale = true;
                            __syncthreads();
                            sc[SC_SUMLOG] = 0.0;
                            rescale = true;
                        }
                    }
                }
            }
        }
        if (stop) { done = true; break; }
        if (last) { next_iter(); done = true; break; }
        if (rescale) { resume = true; break; }      // -> refactor() with the new scale, then finish this iteration
        if (ev) sm[L::O_W + ve] += alpha * (sm[L::O_U + ve] - sm[L::O_UT + ve]);
        __syncthreads();
        F2_ACC(7);      // the slow path of a check iteration (residual products, reductions, termination / rescale logic, relaxed update)
        next_iter();
    }
    }

    __syncthreads();
    F2_STAMP(4);
    const int tid_w = Co::thread_id(wave);
    const double tau = fabs(sm[L::O_U + OT]);
    const double sigma = sc[SC_SIGMA];
    if (status == 0) {   // ran out of iterations (SCS set_unfinished)
        const double kap = fabs(rtau * (sm[L::O_U + OT] + sm[L::O_W + OT] - 2 * sm[L::O_UT + OT]));
        double r[2] = {0, 0};
        const double isg = 1.0 / sigma;
        const int e = tid_w;
        if (e < m) r[1] = sm[L::O_BV + e] * sm[L::O_U + OY + e] * isg * isg;
        else if (e < m + n) r[0] = sm[L::O_CV + (e - m)] * sm[L::O_U + OX + (e - m)] * isg * isg;
        block_reduce_w<2, NW>(r, 0u, red, wave);
        if (tau > kap) status = 2; else if (r[1] < r[0]) status = -7; else status = -6;
    }
    // ---------------------------------------------------------------- write back (un-normalise)
    {
        const bool solved = (status == 1 || status == 2);
        const bool infeas = (status == -2 || status == -7 || status == -4);      // (failed: everything NaN)
        const double it = solved ? 1.0 / (sigma * tau) : (status == -4 ? NAN : 1.0 / sigma);
        for (int j = tid_w; j < n; j += NT) xo[(size_t)inst * n + j] = infeas ? NAN : sm[L::O_EV + j] * sm[L::O_U + OX + j] * it;
        for (int i = tid_w; i < m; i += NT) {
            const double uy = sm[L::O_U + OY + i], di = sm[L::O_DV + i];
            const double sh = (uy + sm[L::O_W + OY + i] - 2 * sm[L::O_UT + OY + i]) / dyv(i);
            const int io = WL ? row_perm[i] : i;                 // kernel row -> template row
            yo[(size_t)inst * m + io] = (solved || infeas) ? di * uy * it : NAN;
            so[(size_t)inst * m + io] = infeas ? NAN : sh / di * it;
        }
        if (tid_w == 0) {
            iters_o[inst] = iter; status_o[inst] = status;
            if (iters2) iters2[inst] = iter;
            if (resid_o) { resid_o[3 * inst] = sc[SC_RP]; resid_o[3 * inst + 1] = sc[SC_RD]; resid_o[3 * inst + 2] = sc[SC_GAP]; }
        }
    }
#ifdef CE_TIMING
    F2_STAMP(5);
    if (threadIdx.x < 8) f2_tstamp[16 + threadIdx.x] = f2_tacc[threadIdx.x];
    if (threadIdx.x < 8) f2_tstamp2[threadIdx.x] = f2_eacc[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 24) so[(size_t)inst * m + threadIdx.x] = (double)f2_tstamp[threadIdx.x];
    if (threadIdx.x < 8) so[(size_t)inst * m + 24 + threadIdx.x] = (double)f2_tstamp2[threadIdx.x];
#endif
}
