// ce_backward.h -- structured direct adjoint kernel
#pragma once
// ================================================================================================
// BACKWARD
// ================================================================================================
// row kinds after classifying DPi_{K*}(v), v = y - s
enum { RK_EQ = 0, RK_FREE = 1, RK_SOCB = 2, RK_MIX = 3 };   // RK_MIX: rotated PSD row with 0 < DPi eigenvalue < 1

template <bool A_LDS, bool K_LDS>
__global__ void __launch_bounds__(NT)
k_backward(DevT T, int nkcap, int ldk, const double *__restrict__ Avals, const double *__restrict__ xg,
           const double *__restrict__ yg, const double *__restrict__ sg, const double *__restrict__ dxg,
           const double *__restrict__ dyg, double *__restrict__ dAo, double *__restrict__ dqo, long sdqk, long sdqb,
           int *__restrict__ adj_status, double *gwsA, double *gwsK, int *__restrict__ fix = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, inst = blockIdx.x;
    const int n = T.n, m = T.m, lda = T.lda, nq = T.nq, z = T.z;
    const int PB = max(NT, max(n, m));
    const int nqs = nq > 0 ? nq : 1;

    double *p = sm;
    double *A, *K;
    if constexpr (A_LDS) { A = p; p += m * lda; } else { A = gwsA + (size_t)inst * m * lda; }
    if constexpr (K_LDS) { K = p; p += nkcap * ldk; } else { K = gwsK + (size_t)inst * nkcap * ldk; }
    double *bv = p; p += m;          // (unused values; load_instance fills b)
    double *xv = p; p += n;
    double *yv = p; p += m;
    double *vv = p; p += m;          // v = y - s ; later r_y
    double *dv = p; p += m;          // d = DPi dy
    double *qv2 = p; p += m;         // A r_x
    double *rx = p; p += n;
    double *ay = p; p += nqs * n;    // A_c^T e_y
    double *as = p; p += nqs * n;    // A_c^T e_s
    double *cinfo = p; p += 6 * nqs; // per cone: lambda, nz, e_y.d, e_s.d, (spare)
    double *part = p; p += PB;
    double *red = p; p += NW * 8;
    // PSD / exponential / power cones (same construction as k_backward_rt<PSD>, ce_backward_rt.h): eigenvectors per PSD cone, its eigenvalues, the DPi eigenvalue of every
    // rotated row, scratch of the Jacobi solver and of the column rotations (one (X, W) pair per wave), the 3 x 3 eigenvector matrices of the triples
    const bool has_psd = T.ns > 0 || T.nep + T.np > 0;
    double *psdU = p, *psdEv = p, *lamr = p, *psdScr = p, *expW = p;
    if (has_psd) { psdU = p; p += T.ns * T.maxs * T.maxs; psdEv = p; p += T.ns * T.maxs; lamr = p; p += m; psdScr = p; p += 2 * NW * T.maxs * T.maxs + 2 * T.maxs + 8; expW = p; p += 9 * (T.nep + T.np); }
    double *pan = p; if (!K_LDS && T.gen_blocked_b) p += generic_lu_panel_doubles(nkcap);      // panels of the blocked elimination (K in global memory)
    int *ip = (int *)p;
    int *rkind = ip; ip += m;        // row kind
    int *eqrow = ip; ip += m;        // equality index of row (RK_EQ) or -1
    int *ckind = ip; ip += nqs;      // cone kind: 0 = interior of K* (all EQ), 1 = in -K (all FREE), 2 = boundary
    int *ceq = ip; ip += nqs;        // equality index of the e_y row of a boundary cone
    int *perm = ip; ip += nkcap;
    int *misc = ip; ip += 4;         // [0] n_eq, [1] pivot row / pivot found, [2] flags
    int *colrow = ip; ip += nkcap;   // unblocked elimination: position (in perm) of the pivot row of column k, -1: free variable

    load_instance(T, Avals + (size_t)inst * T.nnz_aug, A, bv);
    for (int j = tid; j < n; j += NT) xv[j] = xg[(size_t)inst * n + j];
    for (int i = tid; i < m; i += NT) {
        const double yi = yg[(size_t)inst * m + i];
        yv[i] = yi; vv[i] = yi - sg[(size_t)inst * m + i];
    }
    __syncthreads();
    // ---- classify
    for (int i = tid; i < z + T.l; i += NT) rkind[i] = (i < z || vv[i] > 0) ? RK_EQ : RK_FREE;
    for (int c = tid; c < nq; c += NT) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1], d = r1 - r0;
        int kind; double lam = 0, nz = 0;
        if (d == 1) kind = vv[r0] >= 0 ? 0 : 1;
        else {
            for (int i = r0 + 1; i < r1; i++) nz = fma(vv[i], vv[i], nz);
            nz = sqrt(nz);
            const double t0 = vv[r0];
            if (nz <= t0) kind = 0; else if (nz <= -t0) kind = 1; else { kind = 2; lam = (t0 + nz) / (2 * nz); }
        }
        ckind[c] = kind; cinfo[6 * c] = lam; cinfo[6 * c + 1] = nz;
        for (int i = r0; i < r1; i++) rkind[i] = kind == 0 ? RK_EQ : (kind == 1 ? RK_FREE : RK_SOCB);
    }
    __syncthreads();
    if (has_psd) {
        // PSD cones: V = smat(v_c) = U Lambda U^T.  DPi(v) is diagonal in the orthonormal basis svec(sym(u_a u_b^T)) with eigenvalue B_ab; the rows of the cone are
        // ROTATED into that basis in place, A_c <- Q^T A_c, Q^T x = svec(U^T smat(x) U): afterwards every rotated row is an ordinary equality (B = 1) / free (B = 0) /
        // weighted (theta = B / (1 - B)) row and the machinery below applies.  (A may live in global memory here: the rotation streams the cone's rows once.)
        const int lane = tid & 63, wid = tid >> 6;
        for (int c = 0; c < T.ns; c++) {
            const int k = T.sord[c], r0 = T.soff[c], d = k * (k + 1) / 2;
            double *Um = psdU + c * T.maxs * T.maxs, *ev = psdEv + c * T.maxs;
            psd_jacobi<NT>(vv + r0, k, psdScr, Um, psdScr + 2 * T.maxs * T.maxs, red);     // eigenvalues on diag(psdScr), vectors in Um
            for (int i = tid; i < k; i += NT) ev[i] = psdScr[i * k + i];
            __syncthreads();
            for (int g0 = 0; g0 <= n; g0 += NW) {        // the n columns of A_c and (as column n) the incoming dy_c; one column per wave at a time
                const int col = g0 + wid;
                double *X = psdScr + wid * 2 * T.maxs * T.maxs, *W = X + T.maxs * T.maxs;
                if (col <= n) {
                    for (int idx = lane; idx < k * k; idx += 64) {
                        const int i = idx / k, j = idx - i * k, a = i >= j ? i : j, b = i >= j ? j : i;
                        const int pos = b * k - (b * (b - 1)) / 2 + (a - b);
                        const double v = (col < n) ? A[(size_t)(r0 + pos) * lda + col] : dyg[(size_t)inst * m + r0 + pos];
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
                        if (col < n) A[(size_t)(r0 + pos) * lda + col] = t;
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
        // Exponential / power cones: S = D Pi_K*(v_c) = W diag(theta) W^T (3 x 3, one thread per cone); the triple's rows are rotated, A_c <- W^T A_c
        if (T.nep + T.np > 0) {
            for (int c = tid; c < T.nep + T.np; c += NT) {
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
            for (int idx = tid; idx < (T.nep + T.np) * n; idx += NT) {
                const int c = idx / n, j = idx - c * n, r0 = T.eoff + 3 * c;
                const double *W = expW + 9 * c;
                const double a0 = A[(size_t)r0 * lda + j], a1 = A[(size_t)(r0 + 1) * lda + j], a2 = A[(size_t)(r0 + 2) * lda + j];
#pragma unroll
                for (int a = 0; a < 3; a++) A[(size_t)(r0 + a) * lda + j] = W[a] * a0 + W[3 + a] * a1 + W[6 + a] * a2;
            }
            __syncthreads();
        }
    }
    if (tid == 0) {   // equality numbering (serial scan; m is small)
        int ne = 0;
        for (int i = 0; i < m; i++) eqrow[i] = (rkind[i] == RK_EQ) ? ne++ : -1;
        for (int c = 0; c < nq; c++) ceq[c] = (ckind[c] == 2) ? ne++ : -1;
        misc[0] = ne; misc[2] = 0;
    }
    __syncthreads();
    const int neq = misc[0];
    const int NK = n + neq;
    if (NK > nkcap) {   // more active rows than the direct solve holds: degenerate instance (flagged, zero gradient)
        for (int k = tid; k < T.nnz_aug; k += NT) dAo[(size_t)inst * T.nnz_aug + k] = 0.0;
        for (int j = tid; j <= n; j += NT) dqo[j * sdqk + inst * sdqb] = 0.0;
        if (tid == 0 && adj_status) adj_status[inst] = 2;
        if (tid == 0 && fix) fix[1 + atomicAdd(fix, 1)] = inst;
        return;
    }
    // ---- d = DPi(v) dy   (symmetric), per-cone scalars e_y.d, e_s.d
    for (int i = tid; i < z + T.l; i += NT) dv[i] = rkind[i] == RK_EQ ? dyg[(size_t)inst * m + i] : 0.0;
    for (int c = tid; c < nq; c += NT) {
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        const double *h = dyg + (size_t)inst * m;
        if (ckind[c] == 0) { for (int i = r0; i < r1; i++) dv[i] = h[i]; }
        else if (ckind[c] == 1) { for (int i = r0; i < r1; i++) dv[i] = 0.0; }
        else {
            const double t0 = vv[r0], nz = cinfo[6 * c + 1];
            double zh = 0; for (int i = r0 + 1; i < r1; i++) zh = fma(vv[i], h[i], zh);
            dv[r0] = (nz * h[r0] + zh) / (2 * nz);
            for (int i = r0 + 1; i < r1; i++) dv[i] = (vv[i] * h[r0] + (t0 + nz) * h[i] - t0 * vv[i] * zh / (nz * nz)) / (2 * nz);
            // e_y = (1, zhat)/sqrt2, e_s = (1, -zhat)/sqrt2
            double zd = 0; for (int i = r0 + 1; i < r1; i++) zd = fma(vv[i], dv[i], zd);
            zd /= nz;
            cinfo[6 * c + 2] = (dv[r0] + zd) * M_SQRT1_2;   // e_y . d
            cinfo[6 * c + 3] = (dv[r0] - zd) * M_SQRT1_2;   // e_s . d
        }
    }
    __syncthreads();
    // ---- a_y, a_s for boundary cones
    for (int idx = tid; idx < nq * n; idx += NT) {
        const int c = idx / n, j = idx % n;
        if (ckind[c] != 2) continue;
        const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
        const double inz = 1.0 / cinfo[6 * c + 1];
        double a = 0; for (int i = r0 + 1; i < r1; i++) a = fma(A[i * lda + j], vv[i], a);
        a *= inz;
        ay[c * n + j] = (A[r0 * lda + j] + a) * M_SQRT1_2;
        as[c * n + j] = (A[r0 * lda + j] - a) * M_SQRT1_2;
    }
    __syncthreads();
    // ---- f = dx + sum_FREE a_i d_i + sum_B [ a_s (e_s.d) + A_c^T P d / (1-lam) ],  A_c^T P d = A_c^T d - a_y (e_y.d) - a_s (e_s.d):
    //      one product A^T w with w_i = d_i (free rows, cones in -K), d_i / (1 - lam_c) (boundary cones), 0 (equality rows), plus the per-cone
    //      a_y / a_s terms.  (Was a serial loop over all rows inside the single thread that owned each right-hand-side entry.)  f lands in rx.
    for (int i = tid; i < m; i += NT) {
        double w = 0;
        if (i < z + T.l) w = (rkind[i] == RK_FREE) ? dv[i] : 0.0;
        else { const int c = T.rowcone[i]; if (c >= 0) { if (ckind[c] == 1) w = dv[i]; else if (ckind[c] == 2) w = dv[i] / (1 - cinfo[6 * c]); }
               else if (has_psd && rkind[i] == RK_MIX) w = dv[i] / (1 - lamr[i]); }      // rotated PSD / triple rows: weighted rows (free rows carry d = 0)
        qv2[i] = w;
    }
    __syncthreads();
    if constexpr (A_LDS) mv_cols_partial(A, lda, m, n, qv2, part); else mv_cols_g(A, lda, m, n, qv2, part);
    __syncthreads();
    for (int r = tid; r < n; r += NT) {
        double val = dxg[(size_t)inst * n + r] + sum_parts(part, n, r);
        for (int c = 0; c < nq; c++) {
            if (ckind[c] != 2) continue;
            const double il = 1.0 / (1 - cinfo[6 * c]), eyd = cinfo[6 * c + 2], esd = cinfo[6 * c + 3];
            val += as[c * n + r] * esd * (1 - il) - ay[c * n + r] * eyd * il;
        }
        rx[r] = val;
    }
    __syncthreads();
    // ---- assemble K = [[H, -B^T],[B, 0]] | rhs   (a lambda: the blocked elimination below re-assembles when it meets a rank-deficient system)
    auto assemble = [&]() {
    for (int idx = tid; idx < NK * (NK + 1); idx += NT) {
        const int r = idx / (NK + 1), cidx = idx % (NK + 1);
        double val = 0;
        if (r < n && cidx < n) {            // H[a][b] = sum_c theta_c (A_c^T A_c - a_y a_y^T - a_s a_s^T)
            if constexpr (!A_LDS) continue;                    // (global-memory A: formed on the matrix cores below)
            for (int c = 0; c < nq; c++) {
                if (ckind[c] != 2) continue;
                const double lam = cinfo[6 * c], th = lam / (1 - lam);
                const int r0 = T.qoff[c], r1 = T.qoff[c + 1];
                double a = 0; for (int i = r0; i < r1; i++) a = fma(A[i * lda + r], A[i * lda + cidx], a);
                a -= ay[c * n + r] * ay[c * n + cidx] + as[c * n + r] * as[c * n + cidx];
                val = fma(th, a, val);
            }
            if (has_psd) {      // weighted rows of rotated PSD blocks / triples: H += theta_t a_t^T a_t
                for (int t = (T.ns > 0 ? T.soff[0] : T.eoff); t < T.eoff + 3 * (T.nep + T.np); t++)
                    if (rkind[t] == RK_MIX) val = fma(lamr[t] / (1 - lamr[t]), A[t * lda + r] * A[t * lda + cidx], val);
            }
        } else if (r < n && cidx == NK) {   // f (computed above as one product with A^T)
            val = rx[r];
        } else if (r >= n && cidx == NK) {  // d_B  (filled below by the owning row / cone)
            val = 0;
        } else val = 0;
        K[r * ldk + cidx] = val;
    }
    if constexpr (!A_LDS) {
        // H = X^T W X on the matrix cores: X = the rows of the boundary cones (weight theta_c), then a_y and a_s of every boundary cone as two more
        // rows with weight -theta_c; operands straight from the global-memory A (sixteen lanes read one 128-byte line, four rows per
        // instruction), upper-triangular 16 x 16 tiles over the waves, mirrored on store.  (Was n^2 scalar dot products over the cone rows.)
        typedef double v4d __attribute__((ext_vector_type(4)));
        double *wrow = qv2;                                    // per-row weight (free again: f has been formed)
        for (int i = tid; i < m; i += NT) { const int c = (i >= z + T.l) ? T.rowcone[i] : -1; wrow[i] = (c >= 0 && ckind[c] == 2) ? cinfo[6 * c] / (1 - cinfo[6 * c]) : ((has_psd && i >= z + T.l && c < 0 && rkind[i] == RK_MIX) ? lamr[i] / (1 - lamr[i]) : 0.0); }
        __syncthreads();
        const int KT = (n + 15) / 16, wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lc = lane & 15;
        const int i_first = z + T.l;                            // cone rows only
        for (int t = wave; t < KT * (KT + 1) / 2; t += NW) {
            int ti = 0, rem = t;
            while (rem >= KT - ti) { rem -= KT - ti; ti++; }
            const int tj = ti + rem, ca = 16 * ti + lc, cb = 16 * tj + lc;
            const bool va = ca < n, vb = cb < n;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int i0 = i_first; i0 < m; i0 += 16) {
                double av[4], bv4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + 4 * u + lg;
                    const bool ok = i < m;
                    const double *row = A + (size_t)(ok ? i : 0) * lda;
                    const double w = ok ? wrow[i] : 0.0;
                    av[u] = (ok && va && w != 0.0) ? row[ca] * w : 0.0;
                    bv4[u] = (ok && vb && w != 0.0) ? row[cb] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv4[u], acc, 0, 0, 0);
            }
            for (int c0 = 0; c0 < 2 * nq; c0 += 4) {           // a_y (even) / a_s (odd) of cone c0 / 2 ...
                const int e = c0 + lg, c = e >> 1;
                double a = 0.0, b = 0.0;
                if (e < 2 * nq && ckind[c] == 2) {
                    const double th = cinfo[6 * c] / (1 - cinfo[6 * c]);
                    const double *vec = ((e & 1) ? as : ay) + c * n;
                    if (va) a = -th * vec[ca];
                    if (vb) b = vec[cb];
                }
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = 16 * ti + lg + 4 * q, col = 16 * tj + lc;
                if (row < n && col < n) { K[(size_t)row * ldk + col] = acc[q]; if (ti != tj) K[(size_t)col * ldk + row] = acc[q]; }
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < m * n; idx += NT) {   // B rows of plain equality rows
        const int i = idx / n, j = idx % n; const int e = eqrow[i];
        if (e >= 0) { const double a = A[i * lda + j]; K[(n + e) * ldk + j] = a; K[j * ldk + (n + e)] = -a; }
    }
    for (int idx = tid; idx < nq * n; idx += NT) {  // B rows of boundary cones (e_y rows)
        const int c = idx / n, j = idx % n; const int e = ceq[c];
        if (e >= 0) { const double a = ay[c * n + j]; K[(n + e) * ldk + j] = a; K[j * ldk + (n + e)] = -a; }
    }
    for (int i = tid; i < m; i += NT) if (eqrow[i] >= 0) K[(n + eqrow[i]) * ldk + NK] = dv[i];
    for (int c = tid; c < nq; c += NT) if (ceq[c] >= 0) K[(n + ceq[c]) * ldk + NK] = cinfo[6 * c + 2];
    for (int i = tid; i < NK; i += NT) perm[i] = i;
    __syncthreads();
    };
    assemble();
    // ---- Gauss-Jordan with partial pivoting on [K | rhs]
    double kmax;
    {
        double r[1] = {0};
        for (int idx = tid; idx < NK * NK; idx += NT) r[0] = fmax(r[0], fabs(K[(idx / NK) * ldk + idx % NK]));
        block_reduce<1>(r, 1u, red);
        kmax = r[0];
    }
    const double ptol = CE_RANK_TOL * (kmax > 0 ? kmax : 1.0);
    bool unblocked = !(!K_LDS && T.gen_blocked_b);
    if (!unblocked) {
        // BLOCKED Gauss-Jordan with partial pivoting, sixteen pivots per pass over the global-memory matrix (the unblocked loop below streams the
        // whole [K | rhs] once per pivot: at n = 200 that is 0.98 MB x NK steps per instance, HBM-bound).  Per block of columns k0 .. k0 + nb - 1:
        //   1. the column panel (all rows) goes to LDS and the nb pivot steps run on it alone: pivot search, multipliers L[i][kk] for every other row;
        //   2. with Lp = the multipliers at the block's pivot rows, T = I + strict_lower(Lp):  U~ = T^-1 Kp  are the pivot rows as they were when
        //      they were used (Kp: their trailing entries), the pivot rows end as (I - strict_upper(Lp)) U~, every other row as K_i - L_i U~;
        //   3. one wave per strip of sixteen trailing columns (the right-hand side included) does that with MFMA products: U~ stays in the
        //      accumulator layout, which is the B-operand layout of the next product (row lg + 4 q of lane l = K index 4 s + lg for q = s).
        typedef double v4d __attribute__((ext_vector_type(4)));
        double *Pn = pan, *Tm = Pn + (size_t)nkcap * 17, *M2 = Tm + 16 * 17, *pivv = M2 + 16 * 17;
        const int wave = tid >> 6, lane = tid & 63, lg = lane >> 4, lc = lane & 15;
        for (int k0 = 0; k0 < NK; k0 += 16) {
            const int nb = min(16, NK - k0), kb = k0 >> 4;
            for (int idx = tid; idx < NK * 16; idx += NT) { const int pr = idx >> 4, q = idx & 15; Pn[pr * 17 + q] = (q < nb) ? K[(size_t)pr * ldk + k0 + q] : 0.0; }
            __syncthreads();
            for (int kk = 0; kk < nb; kk++) {
                const int k = k0 + kk;
                if (tid < 64) {   // pivot search by wave 0 over logical rows k..NK-1
                    double best = -1; int bi = k;
                    for (int i = k + tid; i < NK; i += 64) { const double v = fabs(Pn[perm[i] * 17 + kk]); if (v > best) { best = v; bi = i; } }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
                        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                    }
                    if (tid == 0) {
                        const int t = perm[k]; perm[k] = perm[bi]; perm[bi] = t;
                        if (best < ptol) { misc[2] = 1; }
                    }
                }
                __syncthreads();
                const int pk = perm[k];
                double piv = Pn[pk * 17 + kk];
                if (fabs(piv) < ptol) piv = (piv < 0 ? -ptol : ptol);
                const double pinv = 1.0 / piv;
                if (tid == 0) pivv[k] = piv;
                for (int i = tid; i < NK; i += NT) {
                    if (i == k) continue;
                    double *row = Pn + perm[i] * 17;
                    const double f = row[kk] * pinv;
                    row[kk] = f;
                    if (f != 0.0) for (int c = kk + 1; c < nb; c++) row[c] = fma(-f, Pn[pk * 17 + c], row[c]);
                }
                __syncthreads();
            }
            auto Lp = [&](int t, int u) -> double { return (t < nb && u < nb) ? Pn[perm[k0 + t] * 17 + u] : 0.0; };     // multiplier of pivot row t at step u (t != u)
            if (tid < 16) {   // T^-1 by forward substitution, one column per thread
                const int c = tid;
                double x[16];
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    double v = (t == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int u = 0; u < 16; u++) if (u < t) v = fma(-Lp(t, u), x[u], v);
                    x[t] = v;
                }
#pragma unroll
                for (int t = 0; t < 16; t++) Tm[t * 17 + c] = x[t];
            }
            for (int idx = tid; idx < 256; idx += NT) { const int t = idx >> 4, u = idx & 15; M2[t * 17 + u] = (t == u) ? 1.0 : (u > t ? -Lp(t, u) : 0.0); }
            __syncthreads();
            const int jt = k0 + nb;                         // first trailing column; the right-hand side is column NK
            for (int ct = wave; jt + 16 * ct <= NK; ct += NW) {
                const int col = jt + 16 * ct + lc; const bool cok = col <= NK;
                v4d ut = {0.0, 0.0, 0.0, 0.0}, pf = {0.0, 0.0, 0.0, 0.0};
                {
                    double kp[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) { const int t = 4 * s4 + lg; kp[s4] = (t < nb && cok) ? K[(size_t)perm[k0 + t] * ldk + col] : 0.0; }
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) ut = __builtin_amdgcn_mfma_f64_16x16x4f64(Tm[lc * 17 + 4 * s4 + lg], kp[s4], ut, 0, 0, 0);
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) pf = __builtin_amdgcn_mfma_f64_16x16x4f64(M2[lc * 17 + 4 * s4 + lg], ut[s4], pf, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; q++) { const int t = lg + 4 * q; if (t < nb && cok) K[(size_t)perm[k0 + t] * ldk + col] = pf[q]; }
                }
                for (int ti = 0; 16 * ti < NK; ti++) {
                    if (ti == kb) continue;
                    v4d acc; int prw[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { const int i = 16 * ti + lg + 4 * q; prw[q] = i < NK ? perm[i] : -1; acc[q] = (prw[q] >= 0 && cok) ? K[(size_t)prw[q] * ldk + col] : 0.0; }
                    const int ia = 16 * ti + lc;
                    const double *lrow = Pn + (ia < NK ? perm[ia] : 0) * 17;
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) {
                        const double a = (ia < NK && 4 * s4 + lg < nb) ? -lrow[4 * s4 + lg] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ut[s4], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) if (prw[q] >= 0 && cok) K[(size_t)prw[q] * ldk + col] = acc[q];
                }
            }
            __syncthreads();
        }
        if (misc[2] & 1) {
            // A vanishing pivot (redundant equality rows, a degenerate active set): sixteen pivots per pass cannot skip a column.  K is rebuilt (f still sits in rx,
            // the solution has not been written) and the rank-revealing elimination below takes over -- rare, and correct instead of flagged.
            __syncthreads();
            if (tid == 0) misc[2] = 0;
            assemble();
            unblocked = true;
        } else {
            for (int k = tid; k < NK; k += NT) {
                const double sol = K[(size_t)perm[k] * ldk + NK] / pivv[k];
                if (k < n) rx[k] = sol; else bv[k - n] = sol;
            }
            __syncthreads();
        }
    }
    if (unblocked) {
    // Rank-revealing like k_backward_rt (and the oracle's dense elimination): a column without an acceptable pivot among the unused rows is a FREE
    // variable (set to zero, skipped, no row consumed) -- redundant equality rows / degenerate active sets, where the reference's LSQR returns a
    // solution of the consistent system.  `rcur` = number of rows used so far (the pivot row of column k sits at position colrow[k] of perm).
    int rcur = 0;
    for (int k = 0; k < NK; k++) {
        if (tid < 64) {   // pivot search by wave 0 over the unused rows rcur..NK-1
            double best = -1; int bi = rcur;
            for (int i = rcur + tid; i < NK; i += 64) { const double v = fabs(K[perm[i] * ldk + k]); if (v > best) { best = v; bi = i; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) {
                if (best >= ptol && rcur < NK) { const int t = perm[rcur]; perm[rcur] = perm[bi]; perm[bi] = t; misc[1] = 1; }
                else { misc[1] = 0; misc[2] |= 4; }
            }
        }
        __syncthreads();
        const bool have = misc[1] != 0;                        // uniform
        if (tid == 0) colrow[k] = have ? rcur : -1;
        if (!have) { __syncthreads(); continue; }              // (misc[1] is rewritten by the next search)
        const int pk = perm[rcur];
        const double piv = K[pk * ldk + k];
        const double pinv = 1.0 / piv;
        // rank-1 update of columns k+1 .. NK: a 16 x 16 thread grid walks rows / columns (no integer division in the loop)
        {
            // (K may live in global memory: the pivot row's entries of a column chunk stay in registers for all rows, and every row chunk is
            //  eight independent loads -- the plain loop had one dependent load / store pair in flight per lane)
            const int ty = tid >> 4, tx = tid & 15;
            const double *prow = K + pk * ldk;
            for (int j0 = k + 1 + tx; j0 <= NK; j0 += 128) {
                double pr[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int jj = j0 + 16 * u; pr[u] = jj <= NK ? prow[jj] : 0.0; }
                for (int i = ty; i < NK; i += 16) {
                    if (i == rcur) continue;
                    double *row = K + perm[i] * ldk;
                    const double f = row[k] * pinv;          // (column k itself is not touched by this step)
                    if (f != 0.0) {
                        double rv[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { const int jj = j0 + 16 * u; rv[u] = jj <= NK ? row[jj] : 0.0; }
#pragma unroll
                        for (int u = 0; u < 8; u++) { const int jj = j0 + 16 * u; if (jj <= NK) row[jj] = fma(-f, pr[u], rv[u]); }
                    }
                }
            }
        }
        rcur++;
        __syncthreads();
    }
    // solution: sol_k = rhs[pivot row of k] / K[that row][k] (0 for a free variable);  r_x -> rx, multipliers rho -> bv (b is not needed by the adjoint)
    for (int k = tid; k < NK; k += NT) {
        const int rk_ = colrow[k];
        double sol = 0.0;
        if (rk_ >= 0) { const int pk = perm[rk_]; sol = K[pk * ldk + NK] / K[pk * ldk + k]; }
        if (k < n) rx[k] = sol; else bv[k - n] = sol;
    }
    __syncthreads();
    }
    // ---- q = A r_x ; r_y
    mv_rows_partial(A, lda, m, n, rx, part);
    __syncthreads();
    for (int i = tid; i < m; i += NT) qv2[i] = sum_parts(part, m, i);
    __syncthreads();
    for (int i = tid; i < z + T.l; i += NT) vv[i] = (eqrow[i] >= 0) ? bv[eqrow[i]] : dv[i];
    for (int c = tid; c < nq; c += NT) {
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
            // coefficients on e_y and e_s after expanding P
            const double cy = rhoy - il * (eyd - lam * eyq), cs = esd - il * (esd - lam * esq);
            // component form: e_y = (1, zhat)/sqrt2 ; e_s = (1, -zhat)/sqrt2
            const double k0 = (cy + cs) * M_SQRT1_2, kz = (cy - cs) * M_SQRT1_2;
            // careful: vv[] (zbar) is overwritten in place -> do row 0 last, scale zhat on the fly
            for (int i = r0 + 1; i < r1; i++) { const double zh = vv[i] * inz; vv[i] = il * (dv[i] - lam * qv2[i]) + kz * zh; }
            vv[r0] = il * (dv[r0] - lam * qv2[r0]) + k0;
        }
    }
    __syncthreads();
    if (has_psd) {   // r~ in the rotated basis, then r_y,c = Q r~ = svec(U smat(r~) U^T)  /  W r~
        for (int t = (T.ns > 0 ? T.soff[0] : T.eoff) + tid; t < T.eoff + 3 * (T.nep + T.np); t += NT) {
            const int rk = rkind[t];
            vv[t] = (rk == RK_EQ) ? bv[eqrow[t]] : (rk == RK_FREE ? dv[t] : (dv[t] - lamr[t] * qv2[t]) / (1 - lamr[t]));
        }
        __syncthreads();
        for (int c = 0; c < T.ns; c++) {
            const int k = T.sord[c], r0 = T.soff[c], d = k * (k + 1) / 2;
            const double *Um = psdU + c * T.maxs * T.maxs;
            double *X = psdScr, *W = X + T.maxs * T.maxs;
            for (int idx = tid; idx < k * k; idx += NT) {
                const int i = idx / k, j = idx - i * k, a = i >= j ? i : j, b = i >= j ? j : i;
                const double v = vv[r0 + b * k - (b * (b - 1)) / 2 + (a - b)];
                X[idx] = (a == b) ? v : v * M_SQRT1_2;
            }
            __syncthreads();
            for (int idx = tid; idx < k * k; idx += NT) {           // W = U X
                const int i = idx / k, j = idx - i * k;
                double acc = 0; for (int a = 0; a < k; a++) acc = fma(Um[i * k + a], X[a * k + j], acc);
                W[idx] = acc;
            }
            __syncthreads();
            for (int pos = tid; pos < d; pos += NT) {               // T = W U^T
                int b = 0, rem = pos; while (rem >= k - b) { rem -= k - b; b++; }
                const int a = b + rem;
                double acc = 0; for (int e = 0; e < k; e++) acc = fma(W[a * k + e], Um[b * k + e], acc);
                vv[r0 + pos] = (a == b) ? acc : acc * M_SQRT2;
            }
            __syncthreads();
        }
        for (int c = tid; c < T.nep + T.np; c += NT) {                     // r_y,c = W r~
            const int r0 = T.eoff + 3 * c;
            const double *W = expW + 9 * c;
            const double t0 = vv[r0], t1 = vv[r0 + 1], t2 = vv[r0 + 2];
#pragma unroll
            for (int a = 0; a < 3; a++) vv[r0 + a] = W[3 * a] * t0 + W[3 * a + 1] * t1 + W[3 * a + 2] * t2;
        }
        __syncthreads();
    }
    // ---- outputs in the boundary convention: dA_eval = [-dA.data, db[b_idx]], dq_eval = [dc, 0]
    //      dA_ij = x_j r_y,i - y_i r_x,j ; db = -r_y ; dc = -r_x     (r_tau pinned to 0)
    for (int k = tid; k < T.nnz_aug; k += NT) {
        const int i = T.rowidx[k], j = T.colidx[k];
        const double val = (j < n) ? -(xv[j] * vv[i] - yv[i] * rx[j]) : -vv[i];
        dAo[(size_t)inst * T.nnz_aug + k] = val;
    }
    for (int j = tid; j <= n; j += NT) dqo[j * sdqk + inst * sdqb] = (j < n) ? -rx[j] : 0.0;
    if (tid == 0 && adj_status) adj_status[inst] = misc[2];
    if (tid == 0 && fix && (misc[2] & 4)) fix[1 + atomicAdd(fix, 1)] = inst;      // rank-deficient: re-solved by the LSQR launch behind this kernel (ce_vjp_qp)
}

// ================================================================================================
// layout kernels: (R x C) row-major <-> (C x R) row-major, fp64, 32x32 LDS tiles (+1 pad)
// ================================================================================================
// TS x TS tiles.  TS = 64 (the default of the launch sites, CE_TR_TILE): a wave reads and writes whole 512-byte row segments (with 32 a wave touches two 256-byte pieces of
// different rows) and every thread keeps 16 independent loads in flight before the barrier.
template <int TS>
__global__ void __launch_bounds__(256) k_transpose(const double *__restrict__ in, double *__restrict__ out, int R, int C) {
    __shared__ double tile[TS][TS + 1];
    constexpr int RS = 256 / TS;          // rows of the tile per pass
    const int bx = blockIdx.x * TS, by = blockIdx.y * TS;
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
    double v[TS / RS];
#pragma unroll
    for (int u = 0; u < TS / RS; u++) { const int rr = by + ty + RS * u, cc = bx + tx; v[u] = (rr < R && cc < C) ? in[(size_t)rr * C + cc] : 0.0; }
#pragma unroll
    for (int u = 0; u < TS / RS; u++) tile[ty + RS * u][tx] = v[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TS / RS; u++) { const int cc = bx + ty + RS * u, rr = by + tx; if (rr < R && cc < C) out[(size_t)cc * R + rr] = tile[tx][ty + RS * u]; }
}


// ================================================================================================
// parameter-map evaluation, batch-major:  out (B x rows) = P (B x cols) . map^T,  map in CSR (rows x cols)
// one thread per (row, instance); lanes walk rows -> coalesced 8-byte stores, gathers of P stay inside one instance's row
// ================================================================================================
// Parameter map with the instance's source row staged in LDS: one workgroup = one instance.  The source row (cols doubles) is
// read once, coalesced; every map row then gathers from LDS, so maps that transpose a matrix parameter (CSC order out of a
// row-major parameter, or back) cost one pass over HBM instead of a 16-fold over-fetch of partially used cache lines.
// ACC: out += (rows without entries are left untouched).
template <bool ACC>
__global__ void __launch_bounds__(512) k_parammap_lds(int rows, int cols, const int *__restrict__ indptr, const int *__restrict__ indices,
                                                      const double *__restrict__ vals, const double *__restrict__ P, long ldp,
                                                      double *__restrict__ out, long ldo) {
    extern __shared__ double pl[];
    constexpr int NT = 512, U = 4;
    const double *p = P + (size_t)blockIdx.x * ldp;
    double *o = out + (size_t)blockIdx.x * ldo;
    for (int c = threadIdx.x; c < cols; c += NT) pl[c] = p[c];
    __syncthreads();
    for (int r0 = threadIdx.x; r0 < rows; r0 += U * NT) {
        int t0[U], t1[U];
        bool single = true;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = r0 + u * NT;
            t0[u] = r < rows ? indptr[r] : 0;
            t1[u] = r < rows ? indptr[r + 1] : 0;
            single = single && (t1[u] - t0[u] <= 1);
        }
        if (single) {                                        // the common shape of a canonicalisation map: one entry per row
            double v[U]; int c[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const bool on = t1[u] > t0[u]; v[u] = on ? vals[t0[u]] : 0.0; c[u] = on ? indices[t0[u]] : 0; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int r = r0 + u * NT;
                const double a = v[u] * pl[c[u]];
                if (r < rows) { if (!ACC) o[r] = a; else if (t1[u] > t0[u]) o[r] += a; }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int r = r0 + u * NT;
                if (r >= rows || (ACC && t0[u] == t1[u])) continue;
                double a = 0.0;
                for (int t = t0[u]; t < t1[u]; t++) a = fma(vals[t], pl[indices[t]], a);
                if (ACC) o[r] += a; else o[r] = a;
            }
        }
    }
}

template <int NB, bool ACC>
__global__ void __launch_bounds__(256) k_parammap(int rows, int B, const int *__restrict__ indptr, const int *__restrict__ indices,
                                                  const double *__restrict__ vals, const double *__restrict__ P, long ldp,
                                                  double *__restrict__ out, long ldo) {
    // one thread = one map row for NB consecutive instances: the row's (index, value) pairs are fetched once for NB gathers
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int b0 = blockIdx.y * NB;
    const int nb = min(NB, B - b0);
    const double *p = P + (size_t)b0 * ldp;
    const int t0 = indptr[r], t1 = indptr[r + 1];
    double a[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) a[u] = 0.0;
    if (nb == NB) {
        for (int t = t0; t < t1; t++) {
            const double v = vals[t]; const int c = indices[t];
#pragma unroll
            for (int u = 0; u < NB; u++) a[u] = fma(v, p[(size_t)u * ldp + c], a[u]);
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            double *o = out + (size_t)(b0 + u) * ldo + r;
            if (!ACC) *o = a[u]; else if (t1 > t0) *o += a[u];
        }
    } else {
        for (int u = 0; u < nb; u++) {
            double acc = 0.0;
            for (int t = t0; t < t1; t++) acc = fma(vals[t], p[(size_t)u * ldp + indices[t]], acc);
            double *o = out + (size_t)(b0 + u) * ldo + r;
            if (!ACC) *o = acc; else if (t1 > t0) *o += acc;
        }
    }
}
