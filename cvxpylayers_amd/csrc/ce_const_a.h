// ce_const_a.h -- elementwise / cone / termination kernels of the CONSTANT-A path.
//
// When the template's A is batch-invariant (only b and c are parametrised -- control, portfolio, most "parameters in the
// right-hand side" layers; the reference's Moreau plugin has the same special case, moreau_if.py:234-256) the three matrix
// products of the ADMM iteration are shared by all instances and become fp64 GEMMs over the batch,
//     T = rho_x W_x - W_y A ,   P_x = ((T Q) * 1/(rho_x + scale_b Lambda)) Q^T ,   Q_y = P_x A^T ,
// (A^T D0 A = Q Lambda Q^T once per call, so every instance keeps ITS OWN adaptive scale), executed by rocBLAS on the MFMA
// pipe (cvxpylayers_amd/interfaces/const_a.py).  What remains per instance is elementwise: tau-tilde, u-tilde, the cone
// projections, the relaxed update, and every 25 iterations the termination / adaptive-scale logic.  These kernels do exactly
// that, one workgroup per instance, vectors (x | y | tau) of length l = n + m + 1 with row pitch lp; the arithmetic and the
// order of operations are those of ce_forward_v2.h / oracle/cone_oracle.c.
#pragma once

// K1: one iteration's elementwise part.  PX (B x n) = p_x, QY (B x m) = A p_x.
//   update_w  : 1 on ordinary iterations (w += alpha (u - ut)); 0 on check iterations (the check kernel finishes the iteration)
//   norm_after: 1 if the NEXT iteration is a check iteration (w is then rescaled to norm sqrt(l), as the per-instance kernels do
//               at the top of a check iteration)
__global__ void __launch_bounds__(NT)
k_ca_step(DevT T, int lp, double *__restrict__ Wg, double *__restrict__ UTg, double *__restrict__ Ug,
          const double *__restrict__ PX, long ldpx, const double *__restrict__ QY, long ldqy,
          const double *__restrict__ Gg, const double *__restrict__ PHIg, const double *__restrict__ scale_g,
          const double *__restrict__ invden_g, const int *__restrict__ active, int update_w, int norm_after, double alpha) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (!active[inst]) return;
    const int n = T.n, m = T.m, l = n + m + 1, z = T.z, nq = T.nq;
    double *zb = sm;                 // [l]  cone input, then u
    double *socc = zb + l;           // [2 nq]
    double *red = socc + 2 * (nq > 0 ? nq : 1);     // [NW * 8]
    double *W = Wg + (size_t)inst * lp, *UT = UTg + (size_t)inst * lp, *U = Ug + (size_t)inst * lp;
    const double *G = Gg + (size_t)inst * lp, *PHI = PHIg + (size_t)inst * lp;
    const double *px = PX + (size_t)inst * ldpx, *qy = QY + (size_t)inst * ldqy;
    const double scale = scale_g[inst], inv_den = invden_g[inst], rtau = TAU_FACTOR;
    // tau-tilde = (r_tau w_tau + phi . w) / (r_tau + h.g)
    double r[1] = {0};
    for (int e = tid; e < l - 1; e += NT) r[0] = fma(PHI[e], W[e], r[0]);
    block_reduce<1>(r, 0u, red);
    const double tau_t = (rtau * W[l - 1] + r[0]) * inv_den;
    for (int e = tid; e < l; e += NT) {
        double ute, ze;
        const double we = W[e];
        if (e < n) { ute = px[e] - tau_t * G[e]; ze = 2 * ute - we; }
        else if (e < l - 1) {
            const int i = e - n;
            const double dy = (i < z) ? ZERO_CONE_FACTOR * scale : scale;
            ute = we + dy * qy[i] - tau_t * G[e]; ze = 2 * ute - we;
            if (i >= z && i < z + T.l && ze < 0) ze = 0;                 // nonnegative rows; the zero cone's dual is free
        } else { ute = tau_t; ze = fmax(0.0, 2 * tau_t - we); }
        UT[e] = ute; zb[e] = ze;
    }
    __syncthreads();
    if (nq > 0) {   // SOC blocks: one leader per cone computes (c0, f), rows apply
        for (int c = tid; c < nq; c += NT) {
            const int r0 = n + T.qoff[c], r1 = n + T.qoff[c + 1];
            const double t0 = zb[r0]; double nz = 0;
            for (int k = r0 + 1; k < r1; k++) nz = fma(zb[k], zb[k], nz);
            nz = sqrt(nz);
            double c0, f;
            if (r1 - r0 == 1) { c0 = fmax(t0, 0.0); f = 0.0; }
            else if (nz <= t0) { c0 = t0; f = 1.0; }
            else if (nz <= -t0) { c0 = 0.0; f = 0.0; }
            else { c0 = 0.5 * (t0 + nz); f = c0 / nz; }
            socc[2 * c] = c0; socc[2 * c + 1] = f;
        }
        __syncthreads();
        for (int i = tid + z + T.l; i < m; i += NT) {
            const int c = T.rowcone[i];
            if (c >= 0) zb[n + i] = (i == T.qoff[c]) ? socc[2 * c] : socc[2 * c + 1] * zb[n + i];
        }
        __syncthreads();
    }
    double nrm[1] = {0};
    for (int e = tid; e < l; e += NT) {
        const double ue = zb[e];
        U[e] = ue;
        if (update_w) { const double we = W[e] + alpha * (ue - UT[e]); W[e] = we; nrm[0] = fma(we, we, nrm[0]); }
    }
    if (update_w && norm_after) {      // uniform
        block_reduce<1>(nrm, 0u, red);
        const double nw = sqrt(nrm[0]);
        if (nw > 0) { const double f = sqrt((double)l) / nw; for (int e = tid; e < l; e += NT) W[e] *= f; }
    }
}

// K2: termination test, infeasibility certificates and adaptive scale of a check iteration (AX = A-hat x-hat, ATY = A-hat^T y-hat
// from the batch GEMMs), then the iteration's relaxed update.  Per-instance state lives in global arrays.
__global__ void __launch_bounds__(NT)
k_ca_check(DevT T, ce_settings S, int lp, int iter, double *__restrict__ Wg, const double *__restrict__ UTg, const double *__restrict__ Ug,
           const double *__restrict__ AX, long ldax, const double *__restrict__ ATY, long lday,
           const double *__restrict__ Dv, const double *__restrict__ Ev, const double *__restrict__ BHg, const double *__restrict__ CHg,
           const double *__restrict__ sigma_g, const double *__restrict__ nb0_g, const double *__restrict__ nc0_g,
           double *__restrict__ scale_g, double *__restrict__ sumlog_g, int *__restrict__ nlog_g, int *__restrict__ lastsc_g,
           int *__restrict__ active, int *__restrict__ status_g, int *__restrict__ iters_g, double *__restrict__ resid_g,
           int *__restrict__ rescaled_g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (!active[inst]) return;
    const int n = T.n, m = T.m, l = n + m + 1, z = T.z;
    double *red = sm;
    double *W = Wg + (size_t)inst * lp;
    const double *UT = UTg + (size_t)inst * lp, *U = Ug + (size_t)inst * lp;
    const double *ax_ = AX + (size_t)inst * ldax, *aty_ = ATY + (size_t)inst * lday;
    const double *bh = BHg + (size_t)inst * m, *ch = CHg + (size_t)inst * n;
    const double scale = scale_g[inst], sigma = sigma_g[inst], isg = 1.0 / sigma;
    const double tau = fabs(U[l - 1]);
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rp, nax, ns, naxs, rd, naty (max) ; ctx, bty (sum)
    for (int i = tid; i < m; i += NT) {
        const double dy = (i < z) ? ZERO_CONE_FACTOR * scale : scale;
        const double sc_ = isg / Dv[i];
        const double ax = ax_[i] * sc_;
        const double uy = U[n + i];
        const double sh = (uy + W[n + i] - 2 * UT[n + i]) / dy * sc_;
        const double bt = bh[i] * tau * sc_;
        r[0] = fmax(r[0], fabs(ax + sh - bt)); r[1] = fmax(r[1], fabs(ax)); r[2] = fmax(r[2], fabs(sh)); r[3] = fmax(r[3], fabs(ax + sh));
        r[7] += bh[i] * uy * isg * isg;
    }
    for (int j = tid; j < n; j += NT) {
        const double sc_ = isg / Ev[j];
        const double aty = aty_[j] * sc_;
        r[4] = fmax(r[4], fabs(aty + ch[j] * tau * sc_)); r[5] = fmax(r[5], fabs(aty));
        r[6] += ch[j] * U[j] * isg * isg;
    }
    block_reduce<8>(r, 0x3Fu, red);
    const double rp = r[0], nax = r[1], ns = r[2], naxs = r[3], rd = r[4], naty = r[5], ctx = r[6], bty = r[7];
    const double nrm_b0 = nb0_g[inst], nrm_c0 = nc0_g[inst];
    int status = 0; bool stop = false, rescale = false;
    double new_scale = scale;
    if (tau > 0) {
        const double res_pri = rp / tau, res_dual = rd / tau, gap = fabs(ctx + bty) / tau;
        if (tid == 0) { resid_g[3 * inst] = res_pri; resid_g[3 * inst + 1] = res_dual; resid_g[3 * inst + 2] = gap; }
        const double prl = fmax(fmax(nrm_b0 * tau, ns), nax) / tau, drl = fmax(nrm_c0 * tau, naty) / tau;
        const double grl = fmax(fabs(ctx), fabs(bty)) / tau;
        if (res_pri <= S.eps_abs + S.eps_rel * prl && res_dual <= S.eps_abs + S.eps_rel * drl && gap <= S.eps_abs + S.eps_rel * grl) { status = 1; stop = true; }
    }
    if (!stop && bty < 0 && naty / (-bty) <= S.eps_infeas) { status = -2; stop = true; }
    if (!stop && ctx < 0 && naxs / (-ctx) <= S.eps_infeas) { status = -1; stop = true; }
    if (!stop && S.adaptive_scale && iter > 0) {
        const double dp = fmax(fmax(nax, ns), nrm_b0 * tau), dd = fmax(naty, nrm_c0 * tau);
        const double rel_p = rp / (dp > 0 ? dp : 1), rel_d = rd / (dd > 0 ? dd : 1);
        if (rel_p > 0 && rel_d > 0 && isfinite(rel_p) && isfinite(rel_d)) {
            const double sum_log = sumlog_g[inst] + log(rel_p) - log(rel_d);
            const int n_log = nlog_g[inst] + 1;
            const double factor = sqrt(exp(sum_log / n_log));
            double sl_out = sum_log; int nl_out = n_log;
            if (iter - lastsc_g[inst] >= RESCALING_MIN_ITERS) {
                const double ns2 = fmin(fmax(scale * factor, MIN_SCALE_VALUE), MAX_SCALE_VALUE);
                if (ns2 != scale && (factor > sqrt(10.0) || factor < 1.0 / sqrt(10.0))) { rescale = true; new_scale = ns2; sl_out = 0.0; nl_out = 0; }
            }
            __syncthreads();
            if (tid == 0) { sumlog_g[inst] = sl_out; nlog_g[inst] = nl_out; if (rescale) { lastsc_g[inst] = iter; scale_g[inst] = new_scale; rescaled_g[inst] = 1; } }
        }
    }
    if (stop) {
        if (tid == 0) { status_g[inst] = status; iters_g[inst] = iter; active[inst] = 0; }
        return;
    }
    if (iter + 1 >= S.max_iters) return;     // last iteration: w stays pre-update so that (s, kappa) match the last cone step
    // keep (s, kappa) across a rescale:  w_y+ = rsk_y / r_y+ + 2 ut_y - u_y ; then the relaxed update of this iteration
    const double dy_ratio = new_scale / scale;
    for (int e = tid; e < l; e += NT) {
        double we = W[e];
        const double ue = U[e], ute = UT[e];
        if (rescale && e >= n && e < l - 1) we = (ue + we - 2 * ute) * dy_ratio + 2 * ute - ue;
        W[e] = we + S.alpha * (ue - ute);
    }
}

// K3: classification of unfinished instances (SCS set_unfinished) and un-normalised write-back of (x, y, s).
__global__ void __launch_bounds__(NT)
k_ca_finish(DevT T, int lp, int max_iters, const double *__restrict__ Wg, const double *__restrict__ UTg, const double *__restrict__ Ug,
            const double *__restrict__ Dv, const double *__restrict__ Ev, const double *__restrict__ BHg, const double *__restrict__ CHg,
            const double *__restrict__ sigma_g, const double *__restrict__ scale_g, const int *__restrict__ active,
            int *__restrict__ status_g, int *__restrict__ iters_g, double *__restrict__ xo, double *__restrict__ yo, double *__restrict__ so) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    const int n = T.n, m = T.m, l = n + m + 1, z = T.z;
    double *red = sm;
    const double *W = Wg + (size_t)inst * lp, *UT = UTg + (size_t)inst * lp, *U = Ug + (size_t)inst * lp;
    const double *bh = BHg + (size_t)inst * m, *ch = CHg + (size_t)inst * n;
    const double sigma = sigma_g[inst], scale = scale_g[inst], rtau = TAU_FACTOR;
    const double tau = fabs(U[l - 1]);
    int status = status_g[inst];
    if (active[inst]) {    // ran out of iterations
        const double kap = fabs(rtau * (U[l - 1] + W[l - 1] - 2 * UT[l - 1]));
        double r[2] = {0, 0};
        const double isg = 1.0 / sigma;
        for (int j = tid; j < n; j += NT) r[0] += ch[j] * U[j] * isg * isg;
        for (int i = tid; i < m; i += NT) r[1] += bh[i] * U[n + i] * isg * isg;
        block_reduce<2>(r, 0u, red);
        if (tau > kap) status = 2; else if (r[1] < r[0]) status = -7; else status = -6;
        if (tid == 0) { status_g[inst] = status; iters_g[inst] = max_iters; }
    }
    const bool solved = (status == 1 || status == 2), infeas = (status == -2 || status == -7);
    const double it = solved ? 1.0 / (sigma * tau) : 1.0 / sigma;
    for (int j = tid; j < n; j += NT) xo[(size_t)inst * n + j] = infeas ? NAN : Ev[j] * U[j] * it;
    for (int i = tid; i < m; i += NT) {
        const double dy = (i < z) ? ZERO_CONE_FACTOR * scale : scale;
        const double uy = U[n + i], di = Dv[i];
        const double sh = (uy + W[n + i] - 2 * UT[n + i]) / dy;
        yo[(size_t)inst * m + i] = (solved || infeas) ? di * uy * it : NAN;
        so[(size_t)inst * m + i] = infeas ? NAN : sh / di * it;
    }
}

// K4: PSD blocks of the cone input (B, lp) projected in place; one workgroup per (instance, cone), Jacobi in LDS (psd_project).
__global__ void __launch_bounds__(NT)
k_ca_psd(DevT T, int lp, double *__restrict__ Ug, const int *__restrict__ active) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, c = blockIdx.y;
    if (!active[inst]) return;
    const int k = T.sord[c];
    double *Sm = sm, *Vm = Sm + T.maxs * T.maxs, *cs = Vm + T.maxs * T.maxs, *red = cs + 2 * T.maxs + 8;
    psd_project<NT>(Ug + (size_t)inst * lp + T.n + T.soff[c], k, Sm, Vm, cs, red);
}

// K4': the same projection by warm-started eigen-refinement on the matrix cores (ce_psd_mfma.h, psd_project_refine; Jacobi sweeps as the
// fall-back).  Vstate (B, ns, maxs * maxs): eigenvectors of every block from the previous call (row-major k x k); warm = 0: none yet.
__global__ void __launch_bounds__(NT)
k_ca_psd_mfma(DevT T, int lp, double *__restrict__ Ug, double *__restrict__ Vstate, int warm, const int *__restrict__ active) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, c = blockIdx.y;
    if (!active[inst]) return;
    const int k = T.sord[c], P = psd_refine_pitch(k);
    const int PM = T.maxs * psd_refine_pitch(T.maxs);
    double *Vl = sm, *Sm = Vl + PM, *Tm = Sm + PM, *Dm = Tm + PM, *Rm = Dm + PM, *cs = Rm + PM, *red = cs + 4 * T.maxs + 16;
    double *Vg = Vstate + ((size_t)inst * T.ns + c) * T.maxs * T.maxs;
    if (warm) for (int idx = threadIdx.x; idx < k * k; idx += NT) { const int i = idx / k, j = idx - i * k; Vl[i * P + j] = Vg[idx]; }
    __syncthreads();
    psd_project_refine<NT>(Ug + (size_t)inst * lp + T.n + T.soff[c], k, Vl, Sm, Tm, Dm, Rm, cs, red, warm, nullptr, (warm & 2) ? 0 : 1);      // (warm & 2: debug, Jacobi sweeps only)
    for (int idx = threadIdx.x; idx < k * k; idx += NT) { const int i = idx / k, j = idx - i * k; Vg[idx] = Vl[i * P + j]; }
}

// K4b: exponential / power cone triples of the cone input (B, lp) projected in place, one thread per (instance, cone); `roots`
// (B, nep + np) keeps each cone's root between iterations (the warm start of the bracketed Newton iteration, ce_expcone.h).
__global__ void __launch_bounds__(NT)
k_ca_triples(DevT T, int lp, int B, double *__restrict__ Ug, double *__restrict__ roots, const int *__restrict__ active) {
    const int ntri = T.nep + T.np;
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= B * ntri) return;
    const int inst = idx / ntri, c = idx - inst * ntri;
    if (!active[inst]) return;
    double *zc = Ug + (size_t)inst * lp + T.n + T.eoff + 3 * c;
    double *rs = roots + (size_t)inst * ntri + c;
    if (c < T.nep) exp_project_dual(zc, rs); else pow_project_dual_of_entry(zc, T.pw[c - T.nep], rs);
}
// K4c: J (B, nep + np, 9) = D Pi of the same projection at v = y - s (row-major 3x3 per cone), for the batched-LSQR adjoint.
__global__ void __launch_bounds__(NT)
k_ca_triple_jac(DevT T, int B, const double *__restrict__ vg, long ldv, double *__restrict__ Jg) {
    const int ntri = T.nep + T.np;
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= B * ntri) return;
    const int inst = idx / ntri, c = idx - inst * ntri;
    const double *v = vg + (size_t)inst * ldv + T.eoff + 3 * c;
    double w[3] = {-v[0], -v[1], -v[2]}, J[9];
    if (c < T.nep) { exp_dproject(w, J); for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i]; }
    else {
        const double a = T.pw[c - T.nep];
        if (a < 0) { const double vv[3] = {v[0], v[1], v[2]}; pow_dproject(vv, -a, J); }
        else { pow_dproject(w, a, J); for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i]; }
    }
    for (int i = 0; i < 9; i++) Jg[(size_t)idx * 9 + i] = J[i];
}

// K5: the relaxed update (and renormalisation) split off k_ca_step, for templates whose PSD blocks are projected in between.
__global__ void __launch_bounds__(NT)
k_ca_update(int l, int lp, double *__restrict__ Wg, const double *__restrict__ UTg, const double *__restrict__ Ug,
            const int *__restrict__ active, int norm_after, double alpha) {
    __shared__ double red[NW * 8];
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (!active[inst]) return;
    double *W = Wg + (size_t)inst * lp;
    const double *UT = UTg + (size_t)inst * lp, *U = Ug + (size_t)inst * lp;
    double nrm[1] = {0};
    for (int e = tid; e < l; e += NT) { const double we = W[e] + alpha * (U[e] - UT[e]); W[e] = we; nrm[0] = fma(we, we, nrm[0]); }
    if (norm_after) {
        block_reduce<1>(nrm, 0u, red);
        const double nw = sqrt(nrm[0]);
        if (nw > 0) { const double f = sqrt((double)l) / nw; for (int e = tid; e < l; e += NT) W[e] *= f; }
    }
}
