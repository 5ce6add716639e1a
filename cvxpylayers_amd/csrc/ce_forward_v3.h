// ce_forward_v3.h -- forward kernel, third generation (round 4): the iteration's three products take their VECTOR operand from
// other lanes, not from LDS.
//
// gfx950 keeps the DP-ALU DPP form of the fp64 multiply-add:
//     v_fmac_f64_dpp  vdst, src0 row_newbcast:k, src1        vdst += src0[lane k of my row of 16 lanes] * src1
// at the rate of a plain v_fma_f64 (scripts/probes/dpp_probe.hip, profiles/r04/a_dpp_probe.log).  A row of 16 lanes that holds a
// vector chunk (entry k in lane k) therefore multiplies a 16 x K block of a matrix with K instructions and NO operand traffic:
// lane l owns output l, its K matrix entries sit in K registers, the k-th instruction broadcasts entry k of the chunk.  k_fwd2
// streamed 26 operands per product and thread through LDS (13 ds_read_b128; 53 KB of LDS reads per product and instance: the
// LDS pipe was ~38 % busy, and the reads' round trips were the latency chains of round 3) and ended every product in a DPP
// butterfly.  Here a product costs its FMAs, one or two cross-row exchanges (v_permlane16_swap / v_permlane32_swap) and ONE
// ds_read_b64 per input register.
//
// Work split of one instance (256 threads = 4 waves = 16 rows of 16 lanes; rg = row of the wave, el = lane in the row):
//   A^T w_y  : wave w owns outputs j = XO w + el (el < XO), its four rows split the TY*4 input slots: at3[TY]; sum over the 4 rows of the wave
//   G t      : same outputs, inputs 13 per row: g3[XO] (G leaves LDS after the factorisation)
//   A p_x    : the pair of rows (2 pairs per wave, 8 per workgroup) owns y slots YO pair + el, each row half of the inputs: ar3[TA]; sum over the pair
// and the y rows are packed by the host (pack_rows3, cone_engine.hip) so that every second-order cone sits at the head of ITS OWN
// pair of rows: the cone's norm is a 16-lane DPP all-reduce and its head a row broadcast -- the projection needs no LDS either.
// Set-up (equilibration, S = rho I + A^T Dy A on the matrix cores, blocked Gauss-Jordan, g / phi) is k_fwd2's, on k_fwd2's tile
// layouts, which are dead when the iteration tiles are materialised.  Iterates are those of k_fwd2 up to the summation order of the
// products.  Shapes: plain cones, n <= 50, at most 8 second-order cones of <= 13 rows, m <= 104 after packing; everything else stays
// on k_fwd2.  OPT-IN (CE_FWD3=1): on MI355X the iteration is not faster than k_fwd2 (see the measurements quoted in DESIGN.md).
#pragma once

template <int K>
__device__ __forceinline__ void fmac_bcast(double &acc, double x, double m) {
    // (the compiler's hazard recogniser sees the operands of inline asm: it inserts the DPP wait states itself)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(K));
}
// sum_k tile[k] * x[k], x[k] = lane (k % 16) of register k / 16 of THIS row of 16 lanes.  Every lane of the wave must be active.
// Three accumulators: a register is re-used every third instruction, beyond the wait states of the DPP read-after-write hazard.
template <int N>
__device__ __forceinline__ double dpp_dot(const double (&tile)[N], double x0, double x1) {
    static_assert(N <= 32, "two input registers");
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    static_for<N>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        fmac_bcast<k % 16>(k % 3 == 0 ? a0 : (k % 3 == 1 ? a1 : a2), k < 16 ? x0 : x1, tile[k]);
    });
    return (a0 + a1) + a2;
}
// v[lane] + v[lane ^ 16]: a = b = v; v_permlane16_swap(a, b) leaves a = rows (0, 0, 2, 2) and b = rows (1, 1, 3, 3) of v
__device__ __forceinline__ double pairsum16(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(ph[0], pl[0]) + __hiloint2double(ph[1], pl[1]);
}
// sum over the four rows of the wave, the same value (and the same summation order) in every lane
__device__ __forceinline__ double rowsum4(double v) {
    const double p = pairsum16(v);
    const int lo = __double2loint(p), hi = __double2hiint(p);
    const auto ql = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), qh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(qh[0], ql[0]) + __hiloint2double(qh[1], ql[1]);
}
// lane 0 of the row of 16
__device__ __forceinline__ double row_bcast0(double v) {
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

#ifndef F3_WPS
#define F3_WPS 3
#endif
template <int CHT, int T1, int CHA, int T2, int CHG, int TG>
__global__ void __launch_bounds__(256, F3_WPS)
k_fwd3(DevT T, ce_settings S, const double *__restrict__ Avals, const double *__restrict__ qv, long sqk, long sqb,
       const int *__restrict__ idx_at, const int *__restrict__ idx_ar, const int *__restrict__ idx_b,
       const int *__restrict__ idx_at3, const int *__restrict__ idx_ar3, const int *__restrict__ slot_soc,
       double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ so, int *__restrict__ iters_o,
       int *__restrict__ status_o, double *__restrict__ resid_o, const int *__restrict__ row_perm) {
    constexpr bool PSD = false, HASP = false, WL = false;      // (the set-up code below is k_fwd2's: its other variants compile away)
    constexpr int NTH = 256, NT = 256, NW = 4;
    const double *const Pvals_g = nullptr; const int nnzP = 0; const int *const idx_p = nullptr;
    using L = F2<CHT, T1, CHA, T2, CHG, TG, NW>;
    using Co = F2Co<CHT, CHA, CHG>;
    constexpr int MP = L::MP, NP = L::NP, VP = L::VP, OY = L::OY, OX = L::OX, OT = L::OT;
    // y slots: 8 pairs of rows x YO; x outputs: 4 waves x XO
    constexpr int MS = MP, YO = MS / 8, TY = MS / 4, XO = (L::NPa + 3) / 4, TA = (4 * XO - 2 + 1) / 2;
    static_assert(MS % 8 == 0 && YO <= 16 && XO <= 16 && TY <= 32 && TA <= 32 && 4 * XO + 3 <= NP && 2 * TA + 32 <= NP + 34, "k_fwd3 layout");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    enum { SC_NB0 = 0, SC_NC0, SC_SIGMA, SC_SUMLOG, SC_RP, SC_RD, SC_GAP };
    double *const sc = sm + L::O_SC;
    double *const red = sm + L::O_RED;
    double *const wpx = sc + 10;                                   // [4] phi_x . w_x, one partial per wave
    int *const socr = reinterpret_cast<int *>(sm + L::O_G);       // [MP] first slot of the slot's SOC (or -1)
    int *const socd = socr + MP;                                   // [MP] 0: zero-cone row or padding slot, 1: nonnegative row, d > 1: row of an SOC of d rows
    double *const Gm = sm + L::O_G + MP;                           // 2*MP ints = MP doubles

    const int tid = threadIdx.x, inst = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // m counts the y SLOTS of the kernel (padding slots are all-zero rows: they stay zero in every vector and enter no sum or maximum);
    // the size of the embedding, l, is the template's
    const int n = T.n, mt = T.m, m = MS, l = n + mt + 1, lk = n + m + 1, ldg = T.ldg, nq = 0, z = 0;
    (void)nq; (void)z; (void)Pvals_g; (void)nnzP; (void)idx_p;
    const int gsz = max(max(n * ldg, 4 * L::LDP), 16 * NP);
    const double *const vals = Avals + (size_t)inst * T.nnz_aug;

#ifdef CE_TIMING
    __shared__ long long f2_tstamp[16];
    if (threadIdx.x < 16) f2_tstamp[threadIdx.x] = 0;
#endif
    F2_STAMP(0);
    for (int i = tid; i < L::O_G; i += NT) sm[i] = 0.0;
    const double *const mtab = sm + L::O_MT;
    for (int i = tid; i < MP; i += NT) { socr[i] = slot_soc[i]; socd[i] = slot_soc[MP + i]; }
    __syncthreads();
    ce_math_table_init(sm + L::O_MT, tid);
    for (int i = tid; i < m; i += NT) { const int ix = idx_b[i]; sm[L::O_BV + i] = ix >= 0 ? vals[ix] : 0.0; sm[L::O_DV + i] = 1.0; }
    for (int j = tid; j < n; j += NT) { sm[L::O_CV + j] = qv[j * sqk + inst * sqb]; sm[L::O_EV + j] = 1.0; }
    __syncthreads();
    {
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT) r[0] = fmax(r[0], fabs(sm[L::O_BV + i]));
        for (int j = tid; j < n; j += NT) r[1] = fmax(r[1], fabs(sm[L::O_CV + j]));
        block_reduce_n<2, NW>(r, 3u, red);
        sc[SC_NB0] = r[0]; sc[SC_NC0] = r[1]; sc[SC_SIGMA] = 1.0;
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
        float *const fPn = reinterpret_cast<float *>(sm + L::O_S3);          // column norms of P-hat (= row norms: symmetric)
        for_each_idx<T1>(idx_at, tid, [&](auto, int k, int ix) { atv[k >> 1][k & 1] = ix >= 0 ? (float)(-vals[ix]) : 0.0f; });   // A = -A_cvx (diffcp_if.py:65)
        for_each_idx<T2>(idx_ar, tid, [&](auto, int k, int ix) { arv[k >> 1][k & 1] = ix >= 0 ? (float)(-vals[ix]) : 0.0f; });
        float *const fEt0 = reinterpret_cast<float *>(sm + L::O_S1), *const fEt1 = reinterpret_cast<float *>(sm + L::O_S2);
        float *const fDt0 = reinterpret_cast<float *>(sm + L::O_U + OY), *const fDt1 = reinterpret_cast<float *>(sm + L::O_UT + OY);
        float *const fRn = reinterpret_cast<float *>(sm + L::O_ZB + OY);
        auto clampf = [](float v) -> float { return v < (float)MIN_SCALE ? 1.0f : (v > (float)MAX_SCALE ? (float)MAX_SCALE : v); };
        double Eacc = 1.0, Dacc = 1.0;        // accumulated scalings of this thread's column / row (owners write them once, after the passes)
        // The column-layout tile belongs to ONE column and the row-layout tile to ONE row: their own factor is the same for all 26 entries, so it is
        // kept as a scalar (ecum, dcum) that multiplies the tile's norm instead of being multiplied into every entry in every pass (half the v_pk_mul_f32)
        float ecum = 1.0f, dcum = 1.0f;
        const int blk_r0 = own2 ? socr[i2] : 0, blk_d = own2 ? abs(socd[i2]) : 0;      // this row's cone block (read once: two LDS round trips less per pass)
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
            {
                if (own1) fEt[j1] = __builtin_amdgcn_rsqf(clampf(cn));
            }
            if (own2) fRn[i2] = rn;          // raw row norms
            if constexpr (WL) wave_lds_exchange(); else __syncthreads();
            if (own2) {
                float a = rn;
                const int r0 = blk_r0, d = blk_d;
                if (d > 1) {   // block-average inside the SOC / PSD block so the scaled cone is still the cone
                    float s0 = 0, s1 = 0;
                    if (d <= 12) {   // one batch of reads, masked (a loop of dependent pairs costs d / 2 LDS round trips per pass; the reads past the block stay inside the vector)
                        float v[12];
#pragma unroll
                        for (int u = 0; u < 12; u++) v[u] = fRn[r0 + u];
#pragma unroll
                        for (int u = 0; u < 12; u += 2) { s0 += (u < d) ? v[u] : 0.0f; s1 += (u + 1 < d) ? v[u + 1] : 0.0f; }
                    } else {
                        int i = 0;
                        for (; i + 1 < d; i += 2) { s0 += fRn[r0 + i]; s1 += fRn[r0 + i + 1]; }
                        if (i < d) s0 += fRn[r0 + i];
                    }
                    a = (s0 + s1) * __builtin_amdgcn_rcpf((float)d);
                }
                fDt[i2] = __builtin_amdgcn_rsqf(clampf(a));
            }
            __syncthreads();
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
                Eacc *= (double)ej; Dacc *= (double)di;
            }
            // no barrier: the next pass writes the other ping-pong buffers (and the row norms, last read before the barrier above)
        }
        if (own1) sm[L::O_EV + j1] = Eacc;
        if (own2) sm[L::O_DV + i2] = Dacc;
        __syncthreads();
        double r[2] = {0, 0};
        for (int i = tid; i < m; i += NT) { const double v = sm[L::O_BV + i] * sm[L::O_DV + i]; sm[L::O_BV + i] = v; r[0] = fmax(r[0], fabs(v)); }
        for (int j = tid; j < n; j += NT) { const double v = sm[L::O_CV + j] * sm[L::O_EV + j]; sm[L::O_CV + j] = v; r[1] = fmax(r[1], fabs(v)); }
        block_reduce_n<2, NW>(r, 3u, red);
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
    auto dyv = [&](int i) -> double { return (socd[i] == 0) ? ZERO_CONE_FACTOR * scale : scale; };   // 1 / r_y  (slot kinds instead of k_fwd2's "i < z": zero-cone rows are fillers like any other single row; padding slots multiply zeros)

    // The ITERATION tiles (see the file header).  They are materialised at the end of refactor() from the instance's values (L2) and the final
    // scalings D, E (LDS) with the expression k_fwd2 uses for its layouts: A-hat[r][j] = (-A_cvx[r][j]) * (D[r] * E[j]).
    double at3[TY], ar3[TA];
    // ---- (re)factor:  G <- (rho_x I + A^T Dy A)^{-1} (LDS);  g, h.g, phi.   Clobbers ZB, TV, PX, S1..S4.
    double g_scale = 0.0;          // the scale G (in LDS) was computed for; 0: none yet
    auto refactor = [&]() {
        F2_STAMP(7);
        const Co co(wave);
        const int tid = co.t;
        const int j1 = co.j1, c1 = co.c1, i2 = co.i2, c2 = co.c2, jg = co.jg, cg = co.cg;
        const bool own1 = (c1 == 0) && (j1 < n), own2 = (c2 == 0) && (i2 < m), owng = (cg == 0) && (jg < n);
        // k_fwd2's tile layouts, alive inside refactor() only
        double at[T1], ar[T2];
        auto materialize_at = [&](const Co &co) {
            const double ej = sm[L::O_EV + (co.j1 < NP ? co.j1 : 0)];
            const double *dv = sm + L::O_DV + T1 * co.c1;
            for_each_idx<T1>(idx_at, co.t, [&](auto, int k, int ix) { at[k] = ix >= 0 ? -vals[ix] * (dv[k] * ej) : 0.0; });
        };
        auto materialize_ar = [&](const Co &co) {
            const double di = sm[L::O_DV + (co.i2 < MP ? co.i2 : 0)];
            const double *evs = sm + L::O_EV + T2 * co.c2;
            for_each_idx<T2>(idx_ar, co.t, [&](auto, int k, int ix) { ar[k] = ix >= 0 ? -vals[ix] * (di * evs[k]) : 0.0; });
        };
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
        bool fast = false;
        if constexpr (TG <= 14) {       // (the wide-tile variants have no registers to spare for Y, G and Z segments: they refactor)
            if (T.f2_neumann && g_scale > 0.0) {
                const double f = uniform_d(scale / g_scale), delta = uniform_d(rho_x * (1.0 - f) / f);
                double r[1] = {0};
                if (jg < n) {
                    const double2 *src = reinterpret_cast<const double2 *>(Gm + jg * ldg + TG * cg);
#pragma unroll
                    for (int s2 = 0; s2 < TG / 2; s2++) { const double2 v = src[s2]; r[0] = fma(v.x, v.x, fma(v.y, v.y, r[0])); }
                }
                block_reduce_n<1, NW>(r, 0u, red);
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
        }
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
                    double a[4][4];
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int q2 = 0; q2 < 4; q2++) a[q][q2] = (q < nbv && q2 < nbv) ? rb[q * NP + k0 + q2] : (q == q2 ? 1.0 : 0.0);
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
                if (cgk + 1 < nv) publish(std::integral_constant<int, blk>{}, cgk + 1, buf ^ 1);
                else if (nvn) publish(std::integral_constant<int, (blk + 1 < NBLK ? blk + 1 : blk)>{}, 0, buf ^ 1);
                cnt++;
                __syncthreads();
            }
        });
        }      // (!fast)
        F2_STAMP(9);
        g_scale = uniform_d(scale);
        // G to LDS (the panel data in that region is dead), scratch back to zero
        if (jg < n) {
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
        {
            const double *grow = Gm + (jg < n ? jg : 0) * ldg + TG * cg;
            const double gx = seg_dot_lds<CHG, TG>(grow, sm + L::O_S1 + TG * cg), gk = seg_dot_lds<CHG, TG>(grow, sm + L::O_S2 + TG * cg);
            if (owng) { sm[L::O_GV + OX + jg] = gx; sm[L::O_PX + jg] = gk; }
        }
        __syncthreads();
        materialize_ar(co);
        double r[1] = {0};
        {
            const double agx = seg_dot<CHA, T2>(ar, sm + L::O_GV + OX + T2 * c2), agk = seg_dot<CHA, T2>(ar, sm + L::O_PX + T2 * c2);
            if (own2) {
                const double bi = sm[L::O_BV + i2];
                const double gy = dyv(i2) * (agx + bi);
                sm[L::O_GV + OY + i2] = gy; r[0] += bi * gy;
                sm[L::O_PHI + OY + i2] = bi - agk;
            }
            if (tid < n) { r[0] += sm[L::O_CV + tid] * sm[L::O_GV + OX + tid]; sm[L::O_PHI + OX + tid] = rho_x * sm[L::O_PX + tid]; }
        }
        block_reduce_n<1, NW>(r, 0u, red);
        hg = uniform_d(r[0]);
        inv_den = uniform_d(1.0 / (rtau + hg));
        {   // the iteration tiles (file header).  Lanes el >= XO / YO, columns >= n and padding slots carry -1 in the gather maps: zeros.
            const int lane = tid & 63, rg = lane >> 4, el = lane & 15, jx = XO * wave + el;
            {
                const double ej = sm[L::O_EV + jx];                       // (jx <= 3 XO + 15 < NP; pads of E are never used: their map entries are -1)
                const double *dv = sm + L::O_DV + TY * rg;
                for_each_idx<TY>(idx_at3, tid, [&](auto, int k, int ix) { at3[k] = ix >= 0 ? -vals[ix] * (dv[k] * ej) : 0.0; });
                if (el < XO && jx == n) {                                  // the spare output carries phi_y: the A^T w_y phase also yields phi_y . w_y
#pragma unroll
                    for (int k = 0; k < TY; k++) at3[k] = sm[L::O_PHI + OY + TY * rg + k];
                }
            }
            {
                const int pair = tid >> 5, slot = YO * pair + (el < YO ? el : 0), h = rg & 1;
                const double di = sm[L::O_DV + slot];
                const double *evs = sm + L::O_EV + TA * h;
                for_each_idx<TA>(idx_ar3, tid, [&](auto, int k, int ix) { ar3[k] = ix >= 0 ? -vals[ix] * (di * evs[k]) : 0.0; });
            }
        }
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
        const int inst_ = blockIdx.x;
        double wx = 0, wy = 0; bool bad = false;
        const int e = threadIdx.x;
        if (e < n) { wx = sg * xo[(size_t)inst_ * n + e] / sm[L::O_EV + e]; bad = !(fabs(wx) < 1e300); }
        for (int i = e; i < m; i += NT) {
            const double dvi = sm[L::O_DV + i];
            const int io = row_perm[i];                          // y slot -> template row (-1: padding slot)
            if (io < 0) continue;
            const double v = sg * yo[(size_t)inst_ * mt + io] / dvi + sg * dvi * so[(size_t)inst_ * mt + io] * dyv(i);
            bad = bad || !(fabs(v) < 1e300);
        }
        double rb[1] = {bad ? 1.0 : 0.0};
        block_reduce_n<1, NW>(rb, 1u, red);          // (max over the workgroup; __syncthreads_or would add static LDS)
        if (rb[0] == 0.0) {
            if (e < n) sm[L::O_W + OX + e] = wx;
            for (int i = e; i < m; i += NT) {
                const double dvi = sm[L::O_DV + i];
                const int io = row_perm[i];
                if (io >= 0) sm[L::O_W + OY + i] = sg * yo[(size_t)inst_ * mt + io] / dvi + sg * dvi * so[(size_t)inst_ * mt + io] * dyv(i);
            }
        }
    }
    __syncthreads();

    int status = 0, iter = 0, last_scale_iter = 0, n_log = 0;
    // Anderson acceleration of the iteration map w -> F(w): k_fwd2's (type I, one secant pair, residual safeguard; oracle/cone_oracle.c with aa_mem = 1)
    bool aa_on = S.acceleration_lookback > 0;      // cleared after AA_MAX_REJECT safeguard rejections
    int aa_rej = 0;
    const int aa_int = S.acceleration_interval;
    double *const aaWP = Gm + gsz;
    double *const aaXP = aaWP + VP, *const aaFP = aaXP + VP, *const aaFS = aaFP + VP, *const aaXS = aaFS + VP;
    int aa_iter = 0; bool aa_pending = false, aa_stale = false;      // (|g| before the step lives in sc[8]: no register across the loop)
    bool resume = false;     // true: the iteration interrupted by a rescale still owes its relaxed update
    auto slot_of = [&](int e) -> int { return (e < m) ? OY + e : (e < m + n ? OX + (e - m) : OT); };

    for (bool done = false; !done;) {
    refactor();
    F2_STAMP(3);
    if (resume) {   // relaxed update w += alpha (u - ut) owed by the iteration a rescale interrupted
        const int e = Co::thread_id(wave);
        if (e < lk) { const int ve = slot_of(e); sm[L::O_W + ve] += alpha * (sm[L::O_U + ve] - sm[L::O_UT + ve]); }
        __syncthreads();
        resume = false; iter++;
    }
    for (;;) {
        if (iter >= S.max_iters) { done = true; break; }
        const int e = Co::thread_id(wave);
        const int ve = slot_of(e);
        const bool ev = e < lk;
        const bool check = (iter % CONVERGED_INTERVAL) == 0;
        const bool last = iter + 1 >= S.max_iters;
        if (aa_on) {      // (uniform)
            if (aa_pending) {      // safeguard: residual of the map at the accelerated point against the residual before the step
                const double dd = ev ? aaWP[ve] - sm[L::O_W + ve] : 0.0;
                double r[1] = {dd * dd};
                block_reduce_n<1, NW>(r, 0u, red);
                if (!(uniform_d(sqrt(r[0])) <= sc[8])) {
                    if (ev) { sm[L::O_W + ve] = aaFS[ve]; aaWP[ve] = aaXS[ve]; }
                    aa_iter = 0;
                    if (++aa_rej >= AA_MAX_REJECT) aa_on = false;
                    __syncthreads();
                }
                aa_pending = false;
            }
            if (aa_on && iter > 0 && iter % aa_int == 0 && !aa_stale) {      // (aa_stale: the kept input predates a rescale -- with an interval that puts a step right behind a check iteration it would pair a pre-rescale input with a post-rescale output)
                const double xv = ev ? aaWP[ve] : 0.0, fv = ev ? sm[L::O_W + ve] : 0.0, gv = xv - fv;
                if (aa_iter > 0) {
                    const double xp = ev ? aaXP[ve] : 0.0, fp = ev ? aaFP[ve] : 0.0;
                    const double sv = xv - xp, yv = gv - (xp - fp), dv = fv - fp;
                    double r[5] = {sv * sv, yv * yv, sv * yv, sv * gv, gv * gv};
                    block_reduce_n<5, NW>(r, 0u, red);
                    const double mm = uniform_d(r[2] + 1e-8 * sqrt(r[0]) * sqrt(r[1]));
                    const double gam = uniform_d(r[3] / mm);
                    if (ev) { aaXP[ve] = xv; aaFP[ve] = fv; }
                    if ((__double2hiint(mm) & 0x7fffffff) > 0x01b00000 /* |mm| > ~1e-300, without an fp64 literal */ && fabs(gam) < 1e10) {
                        if (ev) { aaFS[ve] = fv; aaXS[ve] = xv; sm[L::O_W + ve] = fv - gam * dv; }
                        if (threadIdx.x == 0) sc[8] = sqrt(r[4]);
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
            block_reduce_n<1, NW>(r, 0u, red);
            const double nw = uniform_d(sqrt(r[0]));
            if (nw > 0 && ev) {
                const double fsc = sqrt((double)l) / nw;
                sm[L::O_W + ve] = we * fsc;
                if (aa_on) { aaXP[ve] *= fsc; aaFP[ve] *= fsc; aaFS[ve] *= fsc; aaXS[ve] *= fsc; }     // the map is positively homogeneous
            }
            if (aa_on && nw > 0 && threadIdx.x == 0) sc[8] *= sqrt((double)l) / nw;
            __syncthreads();
        }
        if (aa_on && (aa_pending || (iter + 1) % aa_int == 0)) { aa_stale = false; if (ev) aaWP[ve] = sm[L::O_W + ve]; }      // input of this iteration, kept where the top of the next one reads it
        const int lane = e & 63, rg = lane >> 4, el = lane & 15, jx = XO * wave + el;
        const bool upd = !check && !last;          // fast path: the relaxed update happens inside the A p_x phase (else after the convergence check)
        // P1a: t = rho_x w_x - A^T w_y ; the spare output (column n) is phi_y . w_y ; phi_x . w_x as one partial per wave
        double g3[XO];
        {
            const double *wy = sm + L::O_W + OY + TY * rg + el;
            const double x0 = wy[0], x1 = wy[16];          // (lanes >= TY - 16 of x1 are never broadcast)
            const double wxj = sm[L::O_W + OX + jx], phx = sm[L::O_PHI + OX + jx];      // (both zero beyond n)
            const double acc = rowsum4(dpp_dot<TY>(at3, x0, x1));
            const double pw = group_reduce<16, false>(el < XO ? phx * wxj : 0.0);
            // the G segment of the next phase is requested here, before the barrier: it does not depend on t, and 26 registers are free at this point of the
            // iteration (held across the whole loop they pushed six tile entries into scratch).  ldg = 58: the 13 rows a row of lanes reads fall on distinct banks.
            {
                const double *grow = Gm + ((el < XO && jx < n) ? jx : 0) * ldg + XO * rg;
#pragma unroll
                for (int k = 0; k < XO; k++) g3[k] = grow[k];
            }
            if (rg == 0) {
                if (el < XO) { if (jx < n) sm[L::O_TV + jx] = rho_x * wxj - acc; else if (jx == n) sm[L::O_WP] = acc; }
                if (el == 15) wpx[wave] = pw;
            }
        }
        __syncthreads();
        // P1b: p_x = G t
        {
            const double x0 = sm[L::O_TV + XO * rg + el];
            double acc;
            {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0;
                static_for<XO>([&](auto kc) { constexpr int k = decltype(kc)::value; fmac_bcast<k>(k % 3 == 0 ? a0 : (k % 3 == 1 ? a1 : a2), x0, g3[k]); });
                acc = rowsum4((a0 + a1) + a2);
            }
            if (rg == 0 && el < XO && jx < n) sm[L::O_PX + jx] = acc;       // (rows >= n and lanes >= XO multiplied row 0 of G: not stored; columns >= n of G are zero)
            if (e == 0) sm[L::O_WP + 2] = sm[L::O_W + OT];   // snapshot of w_tau: the next phase rewrites it while other waves still need it
        }
        __syncthreads();
        // P2: q = A p_x ; tau-tilde ; u-tilde ; cone projection ; relaxed update -- all in registers, the cone's norm by a 16-lane DPP all-reduce
        {
            const int h = rg & 1, pair = e >> 5;
            const bool yv = el < YO;
            const int ee = OY + YO * pair + (yv ? el : 0);
            const double *pxv = sm + L::O_PX + TA * h + el;
            const double x0 = pxv[0], x1 = pxv[16];
            const double we = sm[L::O_W + ee], gve = sm[L::O_GV + ee];
            const int cd = socd[ee - OY];
            const double wp0 = sm[L::O_WP], wt = sm[L::O_WP + 2], px0 = wpx[0], px1 = wpx[1], px2 = wpx[2], px3 = wpx[3];
            const double q = pairsum16(dpp_dot<TA>(ar3, x0, x1));
            const double tau_t = (rtau * wt + wp0 + ((px0 + px1) + (px2 + px3))) * inv_den;
            const double py = we + ((cd == 0) ? ZERO_CONE_FACTOR * scale : scale) * q;
            const double ute = py - tau_t * gve;
            double ze = 2 * ute - we;
            if (cd == 1 && ze < 0) ze = 0;
            const bool soc = cd > 1 && yv;
            const double qq = group_reduce<16, false>((soc && el > 0) ? ze * ze : 0.0);      // |tail|^2 of the pair's cone (one cone per pair of rows, head at el == 0)
            const double t0 = row_bcast0(ze);
            double ue = ze;
            if (soc) {
                double nz = 0, rinv = 0;
                if (qq > 0) sqrt_rsqrt(qq, nz, rinv);
                if (nz <= t0) { /* inside */ }
                else if (nz <= -t0) ue = 0.0;
                else { const double c0 = 0.5 * (t0 + nz); ue = (el == 0) ? c0 : ue * (c0 * rinv); }
            }
            if (yv && h == 0) {
                if (upd) sm[L::O_W + ee] = we + alpha * (ue - ute);
                else { sm[L::O_UT + ee] = ute; sm[L::O_U + ee] = ue; }
            }
            if (e < 32) {      // the first pair of rows also holds all of p_x (its input registers): the x block of the update
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int jj = TA * h + 16 * r + el;
                    if (16 * r + el < TA && jj < n) {
                        const int ex = OX + jj;
                        const double wx = sm[L::O_W + ex];
                        const double utx = (r ? x1 : x0) - tau_t * sm[L::O_GV + ex];
                        const double ux = 2 * utx - wx;
                        if (upd) sm[L::O_W + ex] = wx + alpha * (ux - utx);
                        else { sm[L::O_UT + ex] = utx; sm[L::O_U + ex] = ux; }
                    }
                }
            }
            if (e == NT - 1) {
                const double ut = fmax(0.0, 2 * tau_t - wt);
                sm[L::O_UT + OT] = tau_t; sm[L::O_U + OT] = ut;
                if (upd) sm[L::O_W + OT] = wt + alpha * (ut - tau_t);
            }
        }
        __syncthreads();
        if (upd) { iter++; continue; }
        // ---- slow path (every CONVERGED_INTERVAL iterations, and the last one)
        bool stop = false, rescale = false;
        if (check) {
            // A-hat x-hat and A-hat^T y-hat with the iteration tiles, parked in ZB; the residuals are an elementwise phase
            {
                const int h = rg & 1, pair = e >> 5;
                const double *uxv = sm + L::O_U + OX + TA * h + el;
                const double ax_raw = pairsum16(dpp_dot<TA>(ar3, uxv[0], uxv[16]));
                if (el < YO && h == 0) sm[L::O_ZB + OY + YO * pair + el] = ax_raw;
                const double *uyv = sm + L::O_U + OY + TY * rg + el;
                const double aty_raw = rowsum4(dpp_dot<TY>(at3, uyv[0], uyv[16]));
                if (rg == 0 && el < XO && jx < n) sm[L::O_ZB + OX + jx] = aty_raw;
            }
            __syncthreads();
            const double tau = uniform_d(fabs(sm[L::O_U + OT]));
            const double isg = uniform_d(1.0 / sc[SC_SIGMA]);
            double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rp, nax, ns, naxs, rd, naty (max) ; ctx, bty (sum)
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
                r[4] = fabs(aty + cj * tau * sc_); r[5] = fabs(aty);
                r[6] = cj * sm[L::O_U + OX + j] * isg * isg;
            }
            block_reduce_n<8, NW>(r, 0x3Fu, red);
            const double rp = uniform_d(r[0]), nax = uniform_d(r[1]), ns = uniform_d(r[2]), naxs = uniform_d(r[3]), rd = uniform_d(r[4]),
                         naty = uniform_d(r[5]), ctx = uniform_d(r[6]), bty = uniform_d(r[7]);
            const double nrm_b0 = uniform_d(sc[SC_NB0]), nrm_c0 = uniform_d(sc[SC_NC0]);
            if (tau > 0) {
                const double res_pri = rp / tau, res_dual = rd / tau, gap = fabs(ctx + bty) / tau;
                sc[SC_RP] = res_pri; sc[SC_RD] = res_dual; sc[SC_GAP] = gap;
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
        if (last) { iter++; done = true; break; }
        if (rescale) { resume = true; break; }      // -> refactor() with the new scale, then finish this iteration
        if (ev) sm[L::O_W + ve] += alpha * (sm[L::O_U + ve] - sm[L::O_UT + ve]);
        __syncthreads();
        iter++;
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
        block_reduce_n<2, NW>(r, 0u, red);
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
            const int io = row_perm[i];                          // y slot -> template row (-1: padding slot)
            if (io < 0) continue;
            yo[(size_t)inst * mt + io] = (solved || infeas) ? di * uy * it : NAN;
            so[(size_t)inst * mt + io] = infeas ? NAN : sh / di * it;
        }
        if (tid_w == 0) {
            iters_o[inst] = iter; status_o[inst] = status;
            if (resid_o) { resid_o[3 * inst] = sc[SC_RP]; resid_o[3 * inst + 1] = sc[SC_RD]; resid_o[3 * inst + 2] = sc[SC_GAP]; }
        }
    }
#ifdef CE_TIMING
    F2_STAMP(5);
    if (threadIdx.x < 12) so[(size_t)inst * mt + threadIdx.x] = (double)f2_tstamp[threadIdx.x];
#endif
}
