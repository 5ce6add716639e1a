// ce_forward_generic.h -- size-generic LDS/L2-resident forward kernel (fallback path)
#pragma once
// ================================================================================================
// FORWARD
// ================================================================================================
template <bool A_LDS, bool G_LDS>
__global__ void __launch_bounds__(NT)
k_forward(DevT T, ce_settings S, const double *__restrict__ Avals, const double *__restrict__ qv, long sqk, long sqb,
          double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ so, int *__restrict__ iters_o,
          int *__restrict__ status_o, double *__restrict__ resid_o, double *gwsA, double *gwsG, double *aa_ws) {
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
    // PSD / exponential / power cones (the size-generic kernel serves every cone type: templates beyond the register-tiled kernels' sizes):
    // Jacobi scratch S, V, (c, s, p, q) per pair of ce_forward_v2.h's psd_project, and the previous root of every triple (Newton warm start, ce_expcone.h)
    double *psdS = p, *psdV = p, *psdC = p;
    if (T.ns > 0) { psdS = p; p += T.maxs * T.maxs; psdV = p; p += T.maxs * T.maxs; psdC = p; p += 2 * T.maxs + 8; }
    const int ntri = T.nep + T.np;
    double *troot = p; p += ntri;
    p += (size_t)(p - sm) & 1;   // (16-byte alignment of what follows)
    double *pan = p;             // !G_LDS: panels of the blocked inversion (generic_gj_panel_doubles(n))

    // products with A / G: LDS-resident -> ce_common.h, global-memory-resident -> the coalesced versions above
    auto AT_times = [&](const double *v, double *out) { if constexpr (A_LDS) mv_cols_partial(A, lda, m, n, v, out); else mv_cols_g(A, lda, m, n, v, out); };
    auto A_times = [&](const double *v, double *out) { if constexpr (A_LDS) mv_rows_partial(A, lda, m, n, v, out); else mv_rows_g(A, lda, m, n, v, out); };
    auto G_times = [&](const double *v, double *out) { if constexpr (G_LDS) mv_cols_partial(G, ldg, n, n, v, out); else mv_cols_g(G, ldg, n, n, v, out); };

    // ---------------------------------------------------------------- load
    for (int c = tid; c < ntri; c += NT) troot[c] = 0.0;
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
            if constexpr (!A_LDS) {   // A in global memory: lanes along the rows, several loads in flight (see mv_rows_g / mv_cols_g)
                row_norms_g(A, lda, m, n, l2, part);
                { const int CH = chunks_for(m); for (int idx = tid + m; idx < CH * m; idx += NT) part[idx] = 0.0; }
                const int CH2 = chunks_for(n), len2 = (m + CH2 - 1) / CH2;
                for (int idx = tid; idx < n * CH2; idx += NT) {
                    const int j = idx % n, ch = idx / n, i0 = ch * len2, i1 = min(m, i0 + len2);
                    double a = 0;
                    for (int i = i0; i < i1; i += 8) {
                        double mv[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) mv[u] = i + u < i1 ? A[(size_t)(i + u) * lda + j] : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; u++) a = l2 ? fma(mv[u], mv[u], a) : fmax(a, fabs(mv[u]));
                    }
                    part2[ch * n + j] = a;
                }
            } else
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
                tv[i] = (i < z + T.l) ? 1.0 / sqrt(clamp_scale(a)) : a;        // cone rows (SOC, PSD, triples): raw norm, averaged below
            }
            for (int j = tid; j < n; j += NT) {
                const int CH = chunks_for(n); double a = part2[j];
                for (int c = 1; c < CH; c++) a = l2 ? a + part2[c * n + j] : fmax(a, part2[c * n + j]);
                if (l2) a = sqrt(a);
                u[j] = 1.0 / sqrt(clamp_scale(a));                                // Et (u is free scratch here)
            }
            __syncthreads();
            if (nq + T.ns + ntri > 0) {   // block-average the row scaling inside each cone block so the scaled cone is still the cone
                for (int c = tid; c < nq + T.ns + ntri; c += NT) {
                    int r0, r1;
                    if (c < nq) { r0 = T.qoff[c]; r1 = T.qoff[c + 1]; }
                    else if (c < nq + T.ns) { r0 = T.soff[c - nq]; r1 = T.soff[c - nq + 1]; }
                    else { r0 = T.eoff + 3 * (c - nq - T.ns); r1 = r0 + 3; }
                    double a = 0;
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
        if constexpr (!A_LDS) {
            // S = rho_x I + A-hat^T Dy A-hat on the matrix cores, operands straight from the global-memory A (coalesced: sixteen lanes read one
            // 128-byte line of a row, four rows per instruction, eight loads in flight): the upper-triangular 16 x 16 tiles over the waves.
            //   D[M][N] += sum_K a[M][K] b[K][N],  a = A-hat[i0 + K][16 ti + M] dy(i0 + K),  b = A-hat[i0 + K][16 tj + N];  lane l supplies
            //   a[l & 15][l >> 4], b[l >> 4][l & 15] and holds D[(l >> 4) + 4 q][l & 15] in register q (MI355X guide, f64 MFMA).
            typedef double v4d __attribute__((ext_vector_type(4)));
            const int KT = (n + 15) / 16, wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lc = lane & 15;
            for (int t = wave; t < KT * (KT + 1) / 2; t += NW) {
                int ti = 0, rem = t;
                while (rem >= KT - ti) { rem -= KT - ti; ti++; }
                const int tj = ti + rem, ca = 16 * ti + lc, cb = 16 * tj + lc;
                const bool va = ca < n, vb = cb < n;
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                for (int i0 = 0; i0 < m; i0 += 16) {
                    double av[4], bv4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = i0 + 4 * u + lg;
                        const bool ok = i < m;
                        const double *row = A + (size_t)(ok ? i : 0) * lda;
                        av[u] = (ok && va) ? row[ca] * dyv(i) : 0.0;
                        bv4[u] = (ok && vb) ? row[cb] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv4[u], acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = 16 * ti + lg + 4 * q, col = 16 * tj + lc;
                    if (row < n && col < n) {
                        G[row * ldg + col] = acc[q] + (row == col ? rho_x : 0.0);
                        if (ti != tj) G[col * ldg + row] = acc[q];
                    }
                }
            }
        } else
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
        if (!G_LDS && T.gen_blocked_f) {
            // BLOCKED Gauss-Jordan inversion of the global-memory S, sixteen pivots per step, every product on the matrix cores.  For the pivot
            // block K (rows / columns k0 .. k0 + 15):  P = S[K,K]^-1 ;  C = S[:,K], R = S[K,:] (panels in LDS) ;  W = -C P for rows outside K, W[K] = P ;
            //     S[i, j] <- (i outside K ? S[i, j] : 0) + W[i,:] R[:, j]   for columns j outside K ;      S[:, K] <- W
            // (the 4-pivot register version of ce_forward_v2.h, one level up).  The scalar version read and wrote the whole matrix once per
            // pivot (n = 200: 128 MB of L2 traffic per inversion); this one once per sixteen.  S is symmetric positive definite: no pivoting.
            typedef double v4d __attribute__((ext_vector_type(4)));
            const int KT = (n + 15) / 16, NP16 = 16 * KT;
            double *Cp = pan, *Pb = Cp + (size_t)NP16 * 17;
            const int wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lc = lane & 15;
            for (int kb = 0; kb < KT; kb++) {
                const int k0 = 16 * kb, nb = min(16, n - k0);
                // column panel C[i][q] = S[i][k0 + q] (zero beyond n) and the pivot block with an identity tail; the row panel R = S[K,:] is read from
                // global memory by the tiles themselves (rows outside K first: they do not write R; then the rows of K, each tile from its own old entries)
                for (int idx = tid; idx < NP16 * 16; idx += NT) { const int i = idx >> 4, q = idx & 15; Cp[i * 17 + q] = (i < n && q < nb) ? G[(size_t)i * ldg + k0 + q] : 0.0; }
                __syncthreads();
                for (int idx = tid; idx < 256; idx += NT) { const int a = idx >> 4, b = idx & 15; Pb[a * 17 + b] = (a < nb && b < nb) ? Cp[(k0 + a) * 17 + b] : (a == b ? 1.0 : 0.0); }
                __syncthreads();
                for (int k = 0; k < 16; k++) {          // 16 x 16 inverse in LDS (all threads: 256 entries)
                    const double pinv = 1.0 / Pb[k * 17 + k];
                    const int a = tid >> 4, b = tid & 15;
                    double v = 0;
                    if (tid < 256) {
                        const double ck = Pb[a * 17 + k], rk = Pb[k * 17 + b];
                        if (a == k) v = (b == k) ? pinv : rk * pinv;
                        else if (b == k) v = -ck * pinv;
                        else v = fma(-ck * pinv, rk, Pb[a * 17 + b]);
                    }
                    __syncthreads();
                    if (tid < 256) Pb[a * 17 + b] = v;
                    __syncthreads();
                }
                // W = -C P (rows outside K), W[K] = P : one 16 x 16 tile per wave and pass, written over C
                for (int ti = wave; ti < KT; ti += NW) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
                    if (ti != kb) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Cp[(16 * ti + lc) * 17 + 4 * s4 + lg], Pb[(4 * s4 + lg) * 17 + lc], acc, 0, 0, 0);
                    }
                    // (a wave reads only its own tile's rows of C before overwriting them: no hazard between waves)
#pragma unroll
                    for (int q = 0; q < 4; q++) { const int row = lg + 4 * q; Cp[(16 * ti + row) * 17 + lc] = (ti != kb) ? -acc[q] : Pb[row * 17 + lc]; }
                }
                __syncthreads();
                // update of every tile (ti, tj): columns outside K get W R (+ the old entry for rows outside K), the block column becomes W
                for (int phase = 0; phase < 2; phase++) {
                    const int ntile = phase == 0 ? (KT - 1) * KT : KT;
                    for (int t = wave; t < ntile; t += NW) {
                        int ti, tj;
                        if (phase == 0) { ti = t / KT; tj = t - ti * KT; if (ti >= kb) ti++; } else { ti = kb; tj = t; }
                        v4d acc = {0.0, 0.0, 0.0, 0.0};
                        if (tj != kb) {
                            double bop[4];
#pragma unroll
                            for (int s4 = 0; s4 < 4; s4++) { const int q = 4 * s4 + lg, col = 16 * tj + lc; bop[s4] = (q < nb && col < n) ? G[(size_t)(k0 + q) * ldg + col] : 0.0; }      // R[q][col]
                            if (ti != kb) {
#pragma unroll
                                for (int q = 0; q < 4; q++) { const int row = 16 * ti + lg + 4 * q, col = 16 * tj + lc; acc[q] = (row < n && col < n) ? G[(size_t)row * ldg + col] : 0.0; }
                            }
#pragma unroll
                            for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Cp[(16 * ti + lc) * 17 + 4 * s4 + lg], bop[s4], acc, 0, 0, 0);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; q++) acc[q] = Cp[(16 * ti + lg + 4 * q) * 17 + lc];
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) { const int row = 16 * ti + lg + 4 * q, col = 16 * tj + lc; if (row < n && col < n) G[(size_t)row * ldg + col] = acc[q]; }
                    }
                    __syncthreads();
                }
            }
        } else {
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
        }
        // tv = Dy*b  (m) ;  part = A^T tv partials
        for (int i = tid; i < m; i += NT) tv[i] = dyv(i) * bv[i];
        __syncthreads();
        AT_times(tv, part);
        __syncthreads();
        // sc[0:n] = c - A^T Dy b   (rhs for g_x) ;  sc[n:2n] = c + A^T Dy b  (k, for phi)
        for (int j = tid; j < n; j += NT) { const double a = sum_parts(part, n, j); sc[j] = cv[j] - a; sc[n + j] = cv[j] + a; }
        __syncthreads();
        G_times(sc, part);      // G symmetric: column form == row form
        G_times(sc + n, part2);
        __syncthreads();
        for (int j = tid; j < n; j += NT) { g[j] = sum_parts(part, n, j); tv[j] = sum_parts(part2, n, j); }   // tv[0:n] = G k
        __syncthreads();
        A_times(g, part);      // A g_x
        A_times(tv, part2);    // A G k
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
    if (S.warm_start) {   // u = (x^, y^, 1), v = (0, s^, 0); w = u + R^-1 v  (see k_fwd2)
        bool bad = false;
        for (int j = tid; j < n; j += NT) bad = bad || !(fabs(xo[(size_t)inst * n + j]) < 1e300);
        for (int i = tid; i < m; i += NT) bad = bad || !(fabs(yo[(size_t)inst * m + i]) < 1e300) || !(fabs(so[(size_t)inst * m + i]) < 1e300);
        double rb[1] = {bad ? 1.0 : 0.0};
        block_reduce<1>(rb, 1u, red);                // (max over the workgroup; __syncthreads_or would add static LDS)
        if (rb[0] == 0.0) {
            for (int j = tid; j < n; j += NT) w[j] = sigma * xo[(size_t)inst * n + j] / Ev[j];
            for (int i = tid; i < m; i += NT) w[n + i] = sigma * yo[(size_t)inst * m + i] / Dv[i] + sigma * Dv[i] * so[(size_t)inst * m + i] * dyv(i);
            __syncthreads();
            phiw_partials();
        }
        __syncthreads();
    }

    int status = 0, iter = 0, last_scale_iter = 0, n_log = 0;
    double sum_log = 0, res_pri = NAN, res_dual = NAN, gap = NAN;
    double tau = 0, kap = 0, ctx = 0, bty = 0;

    // Anderson acceleration of the iteration map w -> F(w): type I, ONE secant pair, residual safeguard, switched off after AA_MAX_REJECT rejections -- the algorithm of
    // k_fwd2 / k_sa_fwd and of the oracle with aa_mem = 1 (oracle/cone_oracle.c solve_one; SCS's acceleration_lookback / acceleration_interval).  Every aa_int
    // iterations, with x = input and f = output of the last iteration, g = x - f, s = x - x_prev, y = g - g_prev, d = f - f_prev:  w <- f - (s.g / (s.y + 1e-8 |s||y|)) d;
    // the next iteration's residual is the safeguard.  The four history vectors (x_prev, f_prev, f_save, w_prev) live in GLOBAL memory (aa_ws: [B][4][lp]; this kernel
    // serves the templates whose iterates already fill LDS): they are touched on two of every aa_int iterations, each entry by the thread that owns it.
    const int lp = l + (l & 1);
    bool aa_on = S.acceleration_lookback > 0 && aa_ws != nullptr, aa_pending = false, aa_stale = false;
    const int aa_int = S.acceleration_interval > 0 ? S.acceleration_interval : 10;
    int aa_iter = 0, aa_rej = 0;
    double aa_normg = 0, aa_hs = 1.0;     // |g| before the step ; factor the stored history has to be scaled by (the renormalisations of w since it was stored)
    double *const aaXP = aa_ws ? aa_ws + (size_t)inst * 4 * lp : nullptr, *const aaFP = aaXP + lp, *const aaFS = aaFP + lp, *const aaWP = aaFS + lp;

    for (iter = 0; iter < S.max_iters; iter++) {
        const bool check = (iter % CONVERGED_INTERVAL) == 0;
        if (aa_on) {
            bool w_changed = false;
            if (aa_pending) {      // safeguard: residual of the map at the accelerated point against the residual before the step
                double rs[1] = {0};
                for (int e = tid; e < l; e += NT) { const double dd = aaWP[e] - w[e]; rs[0] = fma(dd, dd, rs[0]); }
                block_reduce<1>(rs, 0u, red);
                if (!(sqrt(rs[0]) <= aa_normg)) {
                    for (int e = tid; e < l; e += NT) w[e] = aaFS[e] * aa_hs;
                    aa_iter = 0; w_changed = true;
                    if (++aa_rej >= AA_MAX_REJECT) aa_on = false;
                }
                aa_pending = false;
            }
            if (aa_on && iter > 0 && iter % aa_int == 0 && !aa_stale) {      // (aa_stale: w_prev predates a rescale)
                if (aa_iter > 0) {
                    double rr[5] = {0, 0, 0, 0, 0};
                    for (int e = tid; e < l; e += NT) {
                        const double xv = aaWP[e], fv = w[e], gv = xv - fv, xp = aaXP[e] * aa_hs, fp = aaFP[e] * aa_hs;
                        const double sv = xv - xp, yv = gv - (xp - fp);
                        rr[0] = fma(sv, sv, rr[0]); rr[1] = fma(yv, yv, rr[1]); rr[2] = fma(sv, yv, rr[2]); rr[3] = fma(sv, gv, rr[3]); rr[4] = fma(gv, gv, rr[4]);
                    }
                    block_reduce<5>(rr, 0u, red);
                    const double mm = rr[2] + 1e-8 * sqrt(rr[0]) * sqrt(rr[1]), gam = rr[3] / mm;
                    const bool ok = fabs(mm) > 1e-300 && fabs(gam) < 1e10;
                    for (int e = tid; e < l; e += NT) {
                        const double xv = aaWP[e], fv = w[e], fp = aaFP[e] * aa_hs;
                        aaXP[e] = xv; aaFP[e] = fv;
                        if (ok) { aaFS[e] = fv; w[e] = fv - gam * (fv - fp); }
                    }
                    aa_hs = 1.0;
                    if (ok) { aa_normg = sqrt(rr[4]); aa_pending = true; w_changed = true; } else aa_iter = 0;
                } else {
                    for (int e = tid; e < l; e += NT) { aaXP[e] = aaWP[e]; aaFP[e] = w[e]; }
                    aa_hs = 1.0;
                }
                aa_iter++;
            }
            if (w_changed && !(check && iter > 0)) { __syncthreads(); phiw_partials(); __syncthreads(); }      // (the renormalisation below recomputes phi . w itself)
            else if (w_changed) __syncthreads();
        }
        if (check && iter > 0) {   // keep the homogeneous iterate in range
            double r[1] = {0};
            for (int e = tid; e < l; e += NT) r[0] += w[e] * w[e];
            block_reduce<1>(r, 0u, red);
            const double nw = sqrt(r[0]);
            if (nw > 0) {
                const double f = sqrt((double)l) / nw; for (int e = tid; e < l; e += NT) w[e] *= f;
                if (aa_on) { aa_hs *= f; aa_normg *= f; }      // the map is positively homogeneous: the stored history scales with w (lazily)
            }
            __syncthreads();
            phiw_partials();
            __syncthreads();
        }
        if (aa_on && (aa_pending || (iter + 1) % aa_int == 0)) { aa_stale = false; for (int e = tid; e < l; e += NT) aaWP[e] = w[e]; }      // input of this iteration, where the next one needs it
        // S1: A^T w_y
        AT_times(w + n, part);
        __syncthreads();
        // S2: t = rho_x w_x - A^T w_y
        for (int j = tid; j < n; j += NT) tv[j] = rho_x * w[j] - sum_parts(part, n, j);
        __syncthreads();
        // S3: G t
        G_times(tv, part);
        __syncthreads();
        // S4: p_x
        for (int j = tid; j < n; j += NT) ut[j] = sum_parts(part, n, j);
        __syncthreads();
        // S5: A p_x
        A_times(ut, part);
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
                if (i >= z && i < z + T.l && ue < 0) ue = 0;            // nonneg rows; zero-cone dual is free
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
                if (c >= 0) u[n + i] = (i == T.qoff[c]) ? socc[2 * c] : socc[2 * c + 1] * u[n + i];
            }
            __syncthreads();
        }
        // S8': PSD blocks (workgroup-parallel Jacobi in LDS, ce_forward_v2.h) and exponential / power triples (one thread per cone, ce_expcone.h), in place
        for (int c = 0; c < T.ns; c++) psd_project<NT>(u + n + T.soff[c], T.sord[c], psdS, psdV, psdC, red);
        if (ntri > 0) {
            for (int c = tid; c < ntri; c += NT) {
                double *zc = u + n + T.eoff + 3 * c;
                if (c < T.nep) exp_project_dual(zc, troot + c); else pow_project_dual_of_entry(zc, T.pw[c - T.nep], troot + c);
            }
            __syncthreads();
        }
        // ---- termination test / adaptive scale (uniform branch)
        bool stop = false;
        if (check) {
            A_times(u, part);          // A-hat x-hat
            AT_times(u + n, part2);     // A-hat^T y-hat
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
                            sum_log = 0; n_log = 0; last_scale_iter = iter; scale = ns2; aa_iter = 0; aa_pending = false; aa_stale = true;
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

