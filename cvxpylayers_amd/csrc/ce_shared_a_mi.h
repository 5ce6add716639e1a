// ce_shared_a_mi.h -- SHARED-A adjoint (diffcp's LSQR on the full system, ce_shared_a.h) with SEVERAL instances per workgroup.
//
// k_sa_lsqr<RP> gives every instance its own workgroup, and every workgroup streams the shared dense rows A_d^T (256 KB at BASELINE config 5) from L2 twice per
// LSQR iteration: three co-resident workgroups pull 1.5 MB per iteration through one CU's L1, and that fill rate (~31 B/clk) is what an iteration costs
// (57 k cycles, of which ~50 k are the stream).  Here ONE workgroup of NI x 256 threads owns NI instances: the two streaming passes of an iteration run over
// all NI x 256 threads, and every 64-byte piece of a row that a lane loads is used for the NI instances' dot products and transposed products before it is
// dropped -- the stream is paid once per NI instances.  Everything else of the iteration (cone derivative, reductions, vector updates) is done by the 256-thread
// sub-group that owns the instance, exactly as in k_sa_lsqr; the sub-groups share the workgroup's barriers, so an instance that has stopped keeps walking through
// the iteration with its solution frozen until the last instance of the workgroup stops (the passes skip its arithmetic).
//
// Restricted to what config-5-like templates need: zero / nonnegative / second-order cones, the singleton / dense-row split (RP > 0), n, m <= 3 x 256 (the solution
// r lives in registers of the thread that owns the entry: six doubles), no re-solve list.  Everything else takes k_sa_lsqr (cone_engine.hip vjp_lsqr_launch).
// System, recurrences, stopping rule and outputs are k_sa_lsqr's (oracle/cone_oracle.c lsqr_MT / adjoint_one; reference call site diffcp_if.py:86).
#pragma once
#ifdef CE_TIMING   // debug build: shader cycles per phase of the iteration (first thread of the workgroup), written over the first entries of its instance's dA row
#define MI_T(k) do { const long long t1_ = __builtin_readcyclecounter(); mi_tacc[k] += t1_ - mi_t0; mi_t0 = t1_; } while (0)
#else
#define MI_T(k) do { } while (0)
#endif

constexpr int SAMI_EL = 3;          // entries of x (and of y) per thread of the owning sub-group
constexpr int SAMI_RED = 48;        // three reduction buffers of 4 waves x 4 values

// per-instance LDS doubles / the whole workgroup's
__host__ __device__ inline size_t sa_lsqr_mi_per_doubles(int n, int m, int nq, int RP) {
    const size_t me = m + (m & 1), ne = n + (n & 1), sq = 5 * (size_t)(nq > 0 ? nq : 1);
    return (size_t)RP + SAMI_RED + 6 * me + 3 * ne + sq + (sq & 1);
}
__host__ __device__ inline size_t sa_lsqr_mi_lds_doubles(int n, int m, int nq, int RP, int NI) {
    const int nwt = NI * 4;
    const size_t idx = (size_t)m + (3 * (size_t)m + nq + 1 + RP + 1) / 2 + 1;      // the template's row structure, shared by the instances (see the carve)
    return (size_t)NI * nwt * RP + (size_t)nwt * 32 + (size_t)NI * 8 + idx + (idx & 1) + (size_t)NI * sa_lsqr_mi_per_doubles(n, m, nq, RP);
}

template <int RP, int NI>
__global__ void __launch_bounds__(NI * 256, 1)
k_sa_lsqr_mi(DevT T, SaStruct S, SaSplit F, const double *__restrict__ Avals0, long sAb, const double *__restrict__ qg, long sqk, long sqb,
             const double *__restrict__ xg, const double *__restrict__ yg, const double *__restrict__ sg, const double *__restrict__ dxg, const double *__restrict__ dyg,
             double *__restrict__ dAo, double *__restrict__ dqo, long sdqk, long sdqb, int *__restrict__ adj_status, int *__restrict__ iters_o,
             double atol, double btol, double conlim, int itn_lim, int B) {
    // streaming pass: LPR lanes per row of A_d^T, each holding NL 16-byte pieces of it (piece i of lane k: doubles 2 (LPR i + k), + 1: the lanes of a row read whole
    // 128-byte lines), UR rows in flight per lane.  Three instances: 16 lanes per row halve the per-lane operand / accumulator sets (168 VGPRs at 12 waves per CU)
    constexpr int LPR = (NI >= 3 && RP >= 32) ? 16 : 8;
    constexpr int NTH = NI * 256, NWT = NTH / 64, NL = RP / (2 * LPR), RS = NTH / LPR, SNT = 256, SNW = 4, EL = SAMI_EL;
    constexpr int UR = 2;
    extern __shared__ __attribute__((aligned(16))) double sm[];
#ifdef CE_TIMING
    long long mi_tacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mi_t0 = 0;
#endif
    const int tid = threadIdx.x, lt = tid & 255, k8 = tid & (LPR - 1);
    // wave-uniform: the sub-group, the wave inside it and inside the workgroup live in scalar registers (and with them every per-instance base address)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), q = wv >> 2, lw = wv & 3;
    const int n = T.n, m = T.m, z = T.z, nl = T.l, nq = T.nq;
    const int me = m + (m & 1), ne = n + (n & 1);
    const size_t PER = sa_lsqr_mi_per_doubles(n, m, nq, RP);
    // workgroup-shared: partial sums of the transposed products [NI][NWT][RP], of the pass's norms [NWT][8][2], the instances' scalars [NI][8]
    double *const part = sm, *const pacc = part + (size_t)NI * NWT * RP, *const scal = pacc + NWT * 32;
    // the template's row structure in LDS (every phase of the iteration reads it; from global memory each lookup is an exposed L2 round trip, and the
    // phases between the barriers are chains of them): value and column of a single-entry row, slot of a dense row, cone of the row, cone offsets, dense rows
    double *const i_sv = scal + NI * 8;
    int *const i_col = reinterpret_cast<int *>(i_sv + m), *const i_slot = i_col + m, *const i_cone = i_slot + m, *const i_qoff = i_cone + m, *const i_drow = i_qoff + nq + 1;
    const size_t IDX = (size_t)m + (3 * (size_t)m + nq + 1 + RP + 1) / 2 + 1;
    double *const inst0 = i_sv + IDX + (IDX & 1);
    // per-instance vectors (offsets inside an instance's region; every one starts 16-byte aligned)
    const int O_WYD = 0, O_RED = RP, O_VV = RP + SAMI_RED, O_UY = O_VV + me, O_VY = O_UY + me, O_WY = O_VY + me, O_TY = O_WY + me, O_QV = O_TY + me,
              O_UX = O_QV + me, O_VX = O_UX + ne, O_WX = O_VX + ne, O_SOC = O_WX + ne;
    double *const my = inst0 + (size_t)q * PER;
    double *const wyd = my + O_WYD, *const red = my + O_RED, *const vv = my + O_VV, *const uy = my + O_UY, *const vy = my + O_VY, *const wy = my + O_WY,
           *const ty = my + O_TY, *const qv = my + O_QV, *const ux = my + O_UX, *const vx = my + O_VX, *const wx = my + O_WX, *const socs = my + O_SOC;

    const int inst_raw = blockIdx.x * NI + q;
    const bool valid = inst_raw < B;
    const int inst = valid ? inst_raw : B - 1;             // (a sub-group without an instance walks along on a copy of the last one and writes nothing)
    const double *x = xg + (size_t)inst * n, *y = yg + (size_t)inst * m, *s = sg + (size_t)inst * m;
    const bool TAU = qg != nullptr;
    const double *Ab = Avals0 + (size_t)inst * sAb;
    // the instance whose column results this lane finishes in the streaming passes (lane k of a row's eight lanes: instance k)
    const int kq = k8 < NI ? k8 : NI - 1;
    const int inst_k = min((int)blockIdx.x * NI + kq, B - 1);
    const double *cq_k = TAU ? qg + (size_t)inst_k * sqb : nullptr;
    double *const reg_k = inst0 + (size_t)kq * PER;

    for (int i = tid; i < m; i += NTH) { i_sv[i] = F.srow_val[i]; i_col[i] = F.srow_col[i]; i_slot[i] = F.rowslot[i]; i_cone[i] = nq > 0 ? T.rowcone[i] : -1; }
    for (int c = tid; c <= nq; c += NTH) i_qoff[c] = nq > 0 ? T.qoff[c] : m;
    for (int a = tid; a < RP; a += NTH) i_drow[a] = a < F.r ? F.drow[a] : -1;
    double bb[EL];                                         // b_i of the rows this thread owns (the tau row / column of the operator)
#pragma unroll
    for (int e = 0; e < EL; e++) { const int i = lt + e * SNT; const int pb = (TAU && i < m) ? S.bpos[i] : -1; bb[e] = pb >= 0 ? Ab[pb] : 0.0; }
    for (int i = lt; i < m; i += SNT) vv[i] = y[i] - s[i];
    __syncthreads();
    for (int c = lw; c < nq; c += SNW) {            // one wave per cone
        const int r0 = i_qoff[c], r1 = i_qoff[c + 1];
        const double t = vv[r0]; double nz = 0;
        for (int k = r0 + 1 + (tid & 63); k < r1; k += 64) nz = fma(vv[k], vv[k], nz);
        nz = sqrt(wave_reduce_dpp<false>(nz));
        if ((tid & 63) == 0) {
            socs[4 * c] = t; socs[4 * c + 1] = nz;
            socs[4 * c + 2] = (r1 - r0 == 1) ? (t >= 0 ? 0.0 : 1.0) : (nz <= t ? 0.0 : (nz <= -t ? 1.0 : 2.0));     // 0 identity, 1 zero, 2 boundary
        }
    }
    __syncthreads();

    // sink(i, DPi(hs * h)_i) for every row of THIS sub-group's instance (k_sa_lsqr's dproj without PSD blocks / triples).  One workgroup barrier when the
    // template has second-order cones: every sub-group calls it at the same points.
    auto dproj = [&](const double *h, double hs, auto &&sink) {
        if (nq > 0) {
            for (int c = lw; c < nq; c += SNW) {
                double a = 0;
                const int r0 = i_qoff[c], r1 = i_qoff[c + 1];
                for (int k = r0 + 1 + (tid & 63); k < r1; k += 64) a = fma(vv[k], h[k], a);
                a = wave_reduce_dpp<false>(a);
                if ((tid & 63) == 0) { socs[4 * c + 3] = a * hs; socs[4 * nq + c] = h[r0] * hs; }
            }
            __syncthreads();
        }
        for (int i = lt; i < m; i += SNT) {
            double o = h[i] * hs;
            if (i >= z && i < z + nl) o = (vv[i] > 0) ? o : 0.0;
            else {
                const int c = (i >= z + nl && nq > 0) ? i_cone[i] : -1;
                if (c >= 0) {
                    const double kase = socs[4 * c + 2];
                    if (kase == 1.0) o = 0.0;
                    else if (kase == 2.0) {
                        const int r0 = i_qoff[c];
                        const double t = socs[4 * c], nz = socs[4 * c + 1], zh = socs[4 * c + 3], h0 = socs[4 * nq + c];
                        const double nzs = fmax(nz, 1e-300);
                        if (i == r0) o = (nz * h0 + zh) / (2 * nzs);
                        else o = (vv[i] * h0 + (t + nz) * o - t * vv[i] * zh / (nzs * nzs)) / (2 * nzs);
                    }
                }
            }
            sink(i, o);
        }
    };
    // sums of K values over the sub-group: one barrier (each call site has its own buffer `rb`, next written an iteration later)
    auto sub_reduce = [&](auto &v, double *rb) {
        constexpr int K = sizeof(v) / sizeof(double);
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < K; k++) rb[lw * 4 + k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) {
            double t[SNW];
#pragma unroll
            for (int w = 0; w < SNW; w++) t[w] = rb[w * 4 + k];
            v[k] = uniform_d((t[0] + t[1]) + (t[2] + t[3]));      // (sub-group = whole waves: the sums and every scalar of the recurrences derived from them live in scalar registers)
        }
    };

    // One operator application for every instance of the workgroup: the two products share ONE stream over A_d^T.
    //   columns:  fx(j, (A^T yin)_j, c_j) for instance k of the lane (lane k of the row's eight), with that instance's vectors `rk`
    //   rows:     fy(i, (A xin)_i, b_i) by the owning sub-group
    // off_y / off_x: offsets of yin / xin inside an instance's region.  (acc, acct) collect fx's sums for the lane's instance; they are folded over the workgroup
    // into pacc and added to the owning sub-group's sums by its first NWT threads (pass_sums).  Ends synchronised.
    auto both_products = [&](int off_y, int off_x, unsigned livemask, auto &&fx, auto &&fy, int tb = 0) {
        for (int a = lt; a < RP; a += SNT) { const int dr = i_drow[a]; wyd[a] = dr >= 0 ? my[off_y + dr] : 0.0; }
        __syncthreads();
        MI_T(tb);
        {
            double2 vacc[NI][NL], wr[NI][NL];
            static_for<NI>([&](auto Q) {
                constexpr int qq = decltype(Q)::value;
                const double2 *w2 = reinterpret_cast<const double2 *>(inst0 + (size_t)qq * PER + O_WYD) + k8;
#pragma unroll
                for (int i = 0; i < NL; i++) { vacc[qq][i] = double2{0.0, 0.0}; wr[qq][i] = w2[LPR * i]; }
            });
            double pa = 0, pt = 0;
            const double *yk = reg_k + off_y;
            const double s0 = scal[kq * 8], s1 = scal[kq * 8 + 1];      // the lane's instance: (ut, beta) or (vt, alfa), published by its sub-group
            for (int j0 = tid / LPR; j0 < n; j0 += UR * RS) {
                double2 rv[UR][NL];
                int si[UR]; double sv[UR], ce[UR];
#pragma unroll
                for (int u = 0; u < UR; u++) {
                    const int j = j0 + u * RS, jc = j < n ? j : n - 1;
                    const double2 *row = reinterpret_cast<const double2 *>(F.AdT + (size_t)jc * RP) + k8;
#pragma unroll
                    for (int i = 0; i < NL; i++) rv[u][i] = row[LPR * i];
                    si[u] = F.sing_i[jc]; sv[u] = F.sing_v[jc];
                    ce[u] = cq_k ? cq_k[(size_t)jc * sqk] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UR; u++) {
                    const int j = j0 + u * RS;
                    const bool ok = j < n;
                    double mine = 0;
                    static_for<NI>([&](auto Q) {
                        constexpr int qq = decltype(Q)::value;
                        if ((livemask >> qq) & 1u) {
                            const double xv = ok ? (inst0 + (size_t)qq * PER + off_x)[j] : 0.0;
                            double a0 = 0, a1 = 0;
#pragma unroll
                            for (int i = 0; i < NL; i++) {
                                const double2 wv2 = wr[qq][i];
                                a0 = fma(rv[u][i].x, wv2.x, a0); a1 = fma(rv[u][i].y, wv2.y, a1);
                                vacc[qq][i].x = fma(rv[u][i].x, xv, vacc[qq][i].x); vacc[qq][i].y = fma(rv[u][i].y, xv, vacc[qq][i].y);
                            }
                            const double rsum = group_reduce<LPR, false>(a0 + a1);
                            if (k8 == qq) mine = rsum;
                        }
                    });
                    if (ok && k8 < NI && ((livemask >> k8) & 1u)) {
                        if (si[u] >= 0) mine = fma(sv[u], yk[si[u]], mine);
                        else if (si[u] == -2) { for (int k = F.scol_ptr[j]; k < F.scol_ptr[j + 1]; k++) { const int i = F.scol_row[k]; mine = fma(F.srow_val[i], yk[i], mine); } }
                        fx(j, mine, ce[u], s0, s1, pa, pt);
                    }
                }
            }
            static_for<NI>([&](auto Q) {
                constexpr int qq = decltype(Q)::value;
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    const double vx_ = colsum_rows<LPR>(vacc[qq][i].x), vy_ = colsum_rows<LPR>(vacc[qq][i].y);
                    if ((tid & 63) < LPR) reinterpret_cast<double2 *>(part + ((size_t)qq * NWT + wv) * RP)[LPR * i + k8] = double2{vx_, vy_};
                }
            });
            pa = colsum_rows<LPR>(pa); pt = colsum_rows<LPR>(pt);
            if ((tid & 63) < LPR) { pacc[(wv * 16 + k8) * 2] = pa; pacc[(wv * 16 + k8) * 2 + 1] = pt; }
        }
        MI_T(tb + 1);
        __syncthreads();
        MI_T(tb + 2);
#pragma unroll
        for (int e = 0; e < EL; e++) {
            const int i = lt + e * SNT;
            if (i < m) {
                const int c = i_col[i];
                if (c >= 0) fy(i, i_sv[i] * my[off_x + c], bb[e]);
                else {
                    const int aa = i_slot[i];
                    double s_ = 0;                               // slot of a dense row, -1: empty row
                    if (aa >= 0) {
                        double pv[NWT];
#pragma unroll
                        for (int w = 0; w < NWT; w++) pv[w] = part[((size_t)q * NWT + w) * RP + aa];
#pragma unroll
                        for (int w = 0; w < NWT; w++) s_ += pv[w];
                    }
                    fy(i, s_, bb[e]);
                }
            }
        }
        __syncthreads();
        MI_T(tb + 3);
    };
    // the column sums of the last pass that belong to this sub-group's instance (to be added before a sub_reduce)
    auto pass_sums = [&](double &a, double &t) { if (lt < NWT) { a += pacc[(lt * 16 + q) * 2]; t += pacc[(lt * 16 + q) * 2 + 1]; } };
    auto safe = [](double t) -> double { return t > 0 ? t : 1.0; };
    auto publish = [&](double s0, double s1, bool run) { if (lt == 0) { scal[q * 8] = s0; scal[q * 8 + 1] = s1; scal[q * 8 + 7] = run ? 1.0 : 0.0; } };
    auto livemask_of = [&]() -> unsigned { unsigned lm = 0; static_for<NI>([&](auto Q) { constexpr int qq = decltype(Q)::value; if (scal[qq * 8 + 7] != 0.0) lm |= 1u << qq; }); return (unsigned)__builtin_amdgcn_readfirstlane((int)lm); };

    // ---- LSQR (Paige & Saunders) on  N r = dz,  N = M^T  (ce_shared_a.h has the operator; the tau components are sub-group-uniform scalars)
    double rxr[EL], ryr[EL];
    double acc = 0, acct = 0;
#pragma unroll
    for (int e = 0; e < EL; e++) { const int j = lt + e * SNT; rxr[e] = 0; if (j < n) { const double v = dxg[(size_t)inst * n + j]; ux[j] = v; vx[j] = 0.0; acc = fma(v, v, acc); acct = fma(x[j], v, acct); } }
#pragma unroll
    for (int e = 0; e < EL; e++) { const int i = lt + e * SNT; ryr[e] = 0; if (i < m) { const double v = dyg[(size_t)inst * m + i]; ty[i] = v; vy[i] = 0.0; acct = fma(y[i], v, acct); } }
    __syncthreads();
    dproj(ty, 1.0, [&](int i, double o) { uy[i] = o; acc = fma(o, o, acc); });
    double r2[2] = {acc, acct};
    sub_reduce(r2, red);
    double ut = TAU ? -r2[1] : 0.0, vt = 0.0, wt = 0.0, rt = 0.0;          // dz_tau = -(x.dx + y.dy)
    const double bnorm = sqrt(fma(ut, ut, r2[0]));
    double beta = bnorm, ib = 1.0 / safe(beta);
    for (int j = lt; j < n; j += SNT) ux[j] *= ib;
    dproj(uy, ib, [&](int i, double o) { qv[i] = o; });
    for (int i = lt; i < m; i += SNT) uy[i] *= ib;
    ut *= ib;
    publish(ut, 0.0, true);
    __syncthreads();
    // v-hat = N^T u - beta v  (beta = 0, v = 0 at the start)
    auto fx_v = [&](int j, double a, double cj, double ut_k, double beta_k, double &pa, double &pt) {
        double *rk = reg_k;
        const double v = fma(cj, ut_k, a) - beta_k * rk[O_VX + j]; rk[O_VX + j] = v; pa = fma(v, v, pa); pt = fma(cj, rk[O_UX + j], pt);
    };
    acc = 0; acct = 0;
    both_products(O_QV, O_UX, (1u << NI) - 1u, fx_v,
                  [&](int i, double a, double bi) { const double qi = qv[i]; const double v = fma(bi, ut, -a - qi + uy[i]) - 0.0 * vy[i]; vy[i] = v; acc = fma(v, v, acc); acct = fma(bi, qi, acct); });
    pass_sums(acc, acct);
    r2[0] = acc; r2[1] = acct;
    sub_reduce(r2, red + 16);
    vt = TAU ? -r2[1] : 0.0;
    double alfa = sqrt(fma(vt, vt, r2[0]));
    double wsq = 0, ddnorm = 0;
    const double ctol = conlim > 0 ? 1.0 / conlim : 0.0;
    {
        const double ia = 1.0 / safe(alfa);
        for (int j = lt; j < n; j += SNT) { const double v = vx[j] * ia; vx[j] = v; wx[j] = v; wsq = fma(v, v, wsq); }
        for (int i = lt; i < m; i += SNT) { const double v = vy[i] * ia; vy[i] = v; wy[i] = v; wsq = fma(v, v, wsq); }
        vt *= ia; wt = vt;
    }
    double rhobar = alfa, phibar = beta, anorm = 0, xxnorm = 0, zz = 0, cs2 = -1, sn2 = 0;
    bool live = valid && bnorm > 0 && alfa * beta > 0;
    bool run = live && itn_lim > 0;
    int itn = 0;
    publish(vt, alfa, run);
    __syncthreads();
    unsigned lm = livemask_of();
    auto fx_u = [&](int j, double a, double cj, double vt_k, double alfa_k, double &pa, double &pt) {
        double *rk = reg_k;
        const double v = -a - cj * vt_k - alfa_k * rk[O_UX + j]; rk[O_UX + j] = v; pa = fma(v, v, pa); pt = fma(cj, rk[O_VX + j], pt);
    };
#ifdef CE_TIMING
    for (int k = 0; k < 13; k++) mi_tacc[k] = 0;
    mi_t0 = __builtin_readcyclecounter();
#endif
    while (lm != 0u) {
        if (run) itn++;
        // t = N v :  tx = -A^T vy - c vt ;  ty = DPi(A vx - b vt - vy) + vy ;  tt = c.vx + b.vy ;   u-hat = t - alfa u        (scal = (vt, alfa))
        acc = 0; acct = 0;
        both_products(O_VY, O_VX, lm, fx_u,
                      [&](int i, double a, double bi) { const double vyi = vy[i]; ty[i] = a - bi * vt - vyi; acct = fma(bi, vyi, acct); }, 0);
        pass_sums(acc, acct);
        dproj(ty, 1.0, [&](int i, double o) { const double v = o + vy[i] - alfa * uy[i]; uy[i] = v; acc = fma(v, v, acc); });
        MI_T(4);
        double r3[3] = {acc, acct, wsq};
        sub_reduce(r3, red + 32);
        MI_T(5);
        const double wsum = fma(wt, wt, r3[2]);              // |w_{k-1}|^2, tau component included
        ut = TAU ? r3[1] - alfa * ut : 0.0;
        beta = sqrt(fma(ut, ut, r3[0]));
        ib = 1.0 / safe(beta);
        anorm = sqrt(anorm * anorm + alfa * alfa + beta * beta);
        // u = u-hat / beta ;  q = DPi(uy) ;  (tx, ty) = N^T u ;  v-hat = t - beta v        (scal = (ut, beta))
        for (int j = lt; j < n; j += SNT) ux[j] *= ib;
        dproj(uy, ib, [&](int i, double o) { qv[i] = o; });
        for (int i = lt; i < m; i += SNT) uy[i] *= ib;
        ut *= ib;
        publish(ut, beta, run);
        __syncthreads();
        MI_T(6);
        acc = 0; acct = 0;
        both_products(O_QV, O_UX, lm, fx_v,
                      [&](int i, double a, double bi) { const double qi = qv[i]; const double v = fma(bi, ut, -a - qi + uy[i]) - beta * vy[i]; vy[i] = v; acc = fma(v, v, acc); acct = fma(bi, qi, acct); }, 7);
        pass_sums(acc, acct);
        r2[0] = acc; r2[1] = acct;
        sub_reduce(r2, red + 16);
        MI_T(11);
        vt = TAU ? -r2[1] - beta * vt : 0.0;
        alfa = sqrt(fma(vt, vt, r2[0]));
        const double rho = sqrt(rhobar * rhobar + beta * beta);
        const double cs_ = rhobar / safe(rho), sn = beta / safe(rho);
        const double theta = sn * alfa; rhobar = -cs_ * alfa; const double phi = cs_ * phibar; phibar = sn * phibar; const double tau = sn * phi;
        const double t1 = run ? phi / safe(rho) : 0.0, t2 = -theta / safe(rho), ia = 1.0 / safe(alfa);
        wsq = 0;
#pragma unroll
        for (int e = 0; e < EL; e++) {
            const int j = lt + e * SNT;
            if (j < n) { const double v = vx[j] * ia, w = wx[j], wn = v + t2 * w; vx[j] = v; if (run) rxr[e] = fma(t1, w, rxr[e]); wx[j] = wn; wsq = fma(wn, wn, wsq); }
        }
#pragma unroll
        for (int e = 0; e < EL; e++) {
            const int i = lt + e * SNT;
            if (i < m) { const double v = vy[i] * ia, w = wy[i], wn = v + t2 * w; vy[i] = v; if (run) ryr[e] = fma(t1, w, ryr[e]); wy[i] = wn; wsq = fma(wn, wn, wsq); }
        }
        ddnorm += wsum / (safe(rho) * safe(rho));
        { vt *= ia; if (run) rt = fma(t1, wt, rt); wt = fma(t2, wt, vt); }
        const double delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * zz, zbar = rhs / safe(fabs(gambar)) * (gambar > 0 ? 1.0 : (gambar < 0 ? -1.0 : 0.0));
        const double xnorm = sqrt(xxnorm + zbar * zbar);
        const double gamma = sqrt(gambar * gambar + theta * theta);
        cs2 = gambar / safe(gamma); sn2 = theta / safe(gamma); zz = rhs / safe(gamma); xxnorm += zz * zz;
        const double rnorm = phibar, arnorm = alfa * fabs(tau);
        const double test1 = rnorm / safe(bnorm), test2 = arnorm / (anorm * rnorm + 1e-300);
        const double rtol = btol + atol * anorm * xnorm / safe(bnorm);
        const double test3 = 1.0 / (anorm * sqrt(ddnorm) + 1e-300), tt1 = test1 / (1.0 + anorm * xnorm / safe(bnorm));
        if (run && (test1 <= rtol || test2 <= atol || test3 <= ctol || 1.0 + test3 <= 1.0 || 1.0 + test2 <= 1.0 || 1.0 + tt1 <= 1.0)) live = false;
        run = run && live && itn < itn_lim;
        alfa = uniform_d(alfa); beta = uniform_d(beta); ut = uniform_d(ut); vt = uniform_d(vt); wt = uniform_d(wt); rt = uniform_d(rt);
        rhobar = uniform_d(rhobar); phibar = uniform_d(phibar); anorm = uniform_d(anorm); xxnorm = uniform_d(xxnorm); zz = uniform_d(zz);
        cs2 = uniform_d(cs2); sn2 = uniform_d(sn2); ddnorm = uniform_d(ddnorm);
        publish(vt, alfa, run);
        __syncthreads();
        lm = livemask_of();
        MI_T(12);
    }
    // ---- outputs (k_sa_lsqr's): the solution leaves the registers through the dead LSQR vectors
#pragma unroll
    for (int e = 0; e < EL; e++) { const int j = lt + e * SNT; if (j < n) ux[j] = rxr[e]; const int i = lt + e * SNT; if (i < m) uy[i] = ryr[e]; }
    __syncthreads();
    if (valid) {
        double *dA = dAo + (size_t)inst * T.nnz_aug;
        for (int k = lt; k < T.nnz_aug; k += SNT) {
            const int r = T.rowidx[k], c = T.colidx[k];
            dA[k] = (c < n) ? -(x[c] * uy[r] - y[r] * ux[c]) : fma(y[r], rt, -uy[r]);
        }
        for (int j = lt; j <= n; j += SNT) dqo[j * sdqk + inst * sdqb] = (j < n) ? fma(x[j], rt, -ux[j]) : 0.0;
        if (lt == 0) { if (adj_status) adj_status[inst] = live ? 1 : 0; if (iters_o) iters_o[inst] = itn; }
    }
#ifdef CE_TIMING
    __syncthreads();
    if (tid == 0) { double *dA = dAo + (size_t)inst * T.nnz_aug; for (int k = 0; k < 13; k++) dA[k] = (double)mi_tacc[k]; }
#endif
}
