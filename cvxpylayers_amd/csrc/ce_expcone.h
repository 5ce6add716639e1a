// ce_expcone.h -- exponential cone  K_exp = cl{(x,y,z): y > 0, y e^(x/y) <= z}  (SCS / CVXPY row order inside a triple).
//
// Projection of v = (r,s,t) by the Moreau decomposition v = p - d, p in K, d in K*, p.d = 0: on the boundary
//     p = yy (rho, 1, e^rho),   d = mu (-1, rho-1, e^-rho)          (p.d = 0 identically)
// and p - d = v gives yy = (s + r(rho-1))/D, mu = (r - s rho)/D, D = rho^2 - rho + 1, with rho the root of
//     h(rho) = (s + r(rho-1)) e^rho - (r - s rho) e^-rho - t D
// on the interval where yy > 0 and mu > 0 (univariate root finding; Friberg 2023, the method SCS 3.2 adopted).  h is evaluated
// scaled by e^rho (rho <= 0) or e^-rho (rho > 0) so nothing overflows; the root is found by a bracketed Newton iteration that
// starts from the previous ADMM iteration's root of the same cone.  One thread per cone.
// The solver projects onto the DUAL cone: Pi_K*(v) = v + Pi_K(-v), D Pi_K*(v) = I - D Pi_K(-v).
#pragma once
#ifndef EXP_COUNT_ITER
#define EXP_COUNT_ITER
#endif

struct ExpInfo { int kase; double rho, Y, M; };   // kase 0 inside K, 1 inside -K*, 2 the (r<=0, s<=0) face, 3 boundary

__device__ __forceinline__ void exp_g(double rho, double r, double s, double t, double &g, double &dg) {
    const double ny = s + r * (rho - 1), nm = r - s * rho, D = rho * rho - rho + 1, dD = 2 * rho - 1;
    if (rho <= 0) {
        const double E = exp(rho), E2 = E * E;
        g = ny * E2 - nm - t * D * E;
        dg = (r + 2 * ny) * E2 + s - t * (dD + D) * E;
    } else {
        const double F = exp(-rho), F2 = F * F;
        g = ny - nm * F2 - t * D * F;
        dg = r + (s + 2 * nm) * F2 - t * (dD - D) * F;
    }
}

// (p0,p1,p2) <- Pi_K(p0,p1,p2) in place; rho0: warm start (ignored unless inside the bracket).  Everything stays in registers.
__device__ __forceinline__ void exp_project(double &p0, double &p1, double &p2, double rho0, ExpInfo &inf) {
    const double r = p0, s = p1, t = p2;
    inf.rho = rho0; inf.Y = 0; inf.M = 0;
    if ((s > 0 && s * exp(r / s) <= t) || (r <= 0 && s == 0 && t >= 0)) { inf.kase = 0; return; }
    if ((r > 0 && r * exp(s / r) <= -2.718281828459045235 * t) || (r == 0 && s <= 0 && t <= 0)) { inf.kase = 1; p0 = p1 = p2 = 0; return; }
    if (r <= 0 && s <= 0) { inf.kase = 2; p1 = 0; if (t < 0) p2 = 0; return; }
    // bracket: yy > 0 <=> s + r(rho-1) > 0 ; mu > 0 <=> r - s rho > 0.   g < 0 at lo, g > 0 at hi.
    double lo, hi; bool lo_inf = false, hi_inf = false;
    if (r > 0 && s > 0) { lo = 1 - s / r; hi = r / s; }
    else if (r > 0) { lo = 1 - s / r; hi = 0; hi_inf = true; }
    else { hi = r / s; lo = 0; lo_inf = true; }
    double rho = rho0;
    if (!(rho > lo || lo_inf) || !(rho < hi || hi_inf) || !(fabs(rho) <= 1000.0))     // (also rejects NaN / stale LDS contents)
        rho = hi_inf ? lo + 1 : (lo_inf ? hi - 1 : 0.5 * (lo + hi));
    double st = 1;
    for (int it = 0; it < 120; it++) {
        double g, dg; exp_g(rho, r, s, t, g, dg);
        EXP_COUNT_ITER
        if (g > 0) { hi = rho; hi_inf = false; } else if (g < 0) { lo = rho; lo_inf = false; } else break;
        double nr = rho - g / dg;
        // Newton converges quadratically: once a step is this small the error after taking it is ~ step^2
        if (dg > 0 && fabs(nr - rho) <= 3e-9 * (1 + fabs(rho))) { rho = nr; break; }
        const bool ok = (dg > 0) && (lo_inf || nr > lo) && (hi_inf || nr < hi) && fabs(nr - rho) < 64.0;
        if (!ok) {
            if (hi_inf) { st *= 2; nr = rho + st; }
            else if (lo_inf) { st *= 2; nr = rho - st; }
            else nr = 0.5 * (lo + hi);
        }
        rho = nr;
        if (!lo_inf && !hi_inf && hi - lo <= 2e-16 * (1 + fabs(rho))) break;
    }
    const double D = rho * rho - rho + 1;
    double ny = s + r * (rho - 1), nm = r - s * rho;
    if (ny < 0) ny = 0;
    if (nm < 0) nm = 0;
    inf.kase = 3; inf.rho = rho;
    // two algebraically equal forms; each is the accurate one on its side
    if (rho <= 0) {
        const double E = exp(rho), yy = ny / D;
        p0 = yy * rho; p1 = yy; p2 = yy * E;
        inf.Y = yy; inf.M = p2 - t;                     // mu e^-rho (third equation)
    } else {
        const double F = exp(-rho), mu = nm / D;
        p0 = r - mu; p1 = fmax(s + mu * (rho - 1), 0.0); p2 = t + mu * F;
        inf.M = mu; inf.Y = p2;                         // yy e^rho
    }
}

// y-block of the ADMM cone step: v <- Pi_K*(v) = v + Pi_K(-v).   *rho_state keeps the root between iterations.
__device__ __forceinline__ void exp_project_dual(double *v, double *rho_state) {
    const double v0 = v[0], v1 = v[1], v2 = v[2];
    double w0 = -v0, w1 = -v1, w2 = -v2;
    ExpInfo inf;
    exp_project(w0, w1, w2, *rho_state, inf);
    *rho_state = inf.rho;
    v[0] = v0 + w0; v[1] = v1 + w1; v[2] = v2 + w2;
}

// J (row-major 3x3) = D Pi_K(v).  Boundary: p = Y a(rho), p - v = M b(rho) with
//   rho <= 0: a = (rho, 1, E), b = (-E, (rho-1)E, 1) ;  rho > 0: a = (rho F, F, 1), b = (-1, rho-1, F)
// G(Y, M, rho) = Y a - M b = v  =>  dG = [a | -b | Y a' - M b'],  dp = a dY + Y a' drho  =>  J = [a | 0 | Y a'] dG^-1.
__device__ __noinline__ void exp_dproject(const double *v, double *J) {
    double w0 = v[0], w1 = v[1], w2 = v[2];
    ExpInfo inf;
    exp_project(w0, w1, w2, 0.0, inf);
    for (int i = 0; i < 9; i++) J[i] = 0;
    if (inf.kase == 0) { J[0] = J[4] = J[8] = 1; return; }
    if (inf.kase == 1) return;
    if (inf.kase == 2) { J[0] = 1; J[8] = v[2] > 0 ? 1.0 : 0.0; return; }
    const double rho = inf.rho, Y = inf.Y, M = inf.M;
    if (rho < -690) { J[0] = J[4] = 1; return; }       // e^rho underflows: p = (r, s, 0)
    if (rho > 690) { J[8] = 1; return; }               // e^-rho underflows: p = (0, 0, t)
    double a[3], b[3], da[3], db[3];
    if (rho <= 0) { const double E = exp(rho); a[0] = rho; a[1] = 1; a[2] = E; da[0] = 1; da[1] = 0; da[2] = E;
                    b[0] = -E; b[1] = (rho - 1) * E; b[2] = 1; db[0] = -E; db[1] = rho * E; db[2] = 0; }
    else { const double F = exp(-rho); a[0] = rho * F; a[1] = F; a[2] = 1; da[0] = (1 - rho) * F; da[1] = -F; da[2] = 0;
           b[0] = -1; b[1] = rho - 1; b[2] = F; db[0] = 0; db[1] = 1; db[2] = -F; }
    double G[9];
    for (int i = 0; i < 3; i++) { G[i * 3] = a[i]; G[i * 3 + 1] = -b[i]; G[i * 3 + 2] = Y * da[i] - M * db[i]; }
    const double c00 = G[4] * G[8] - G[5] * G[7], c01 = G[5] * G[6] - G[3] * G[8], c02 = G[3] * G[7] - G[4] * G[6];
    const double idet = 1.0 / (G[0] * c00 + G[1] * c01 + G[2] * c02);
    const double i0[3] = {c00 * idet, (G[2] * G[7] - G[1] * G[8]) * idet, (G[1] * G[5] - G[2] * G[4]) * idet};      // row 0 of G^-1
    const double i2[3] = {c02 * idet, (G[1] * G[6] - G[0] * G[7]) * idet, (G[0] * G[4] - G[1] * G[3]) * idet};      // row 2 of G^-1
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[i * 3 + j] = a[i] * i0[j] + Y * da[i] * i2[j];
}

// Eigendecomposition of a symmetric 3x3 (row-major S9; only the symmetric part is used):  S = W diag(th) W^T, cyclic Jacobi.
__device__ __noinline__ void sym3_eig(const double *S9, double *W, double *th) {
    double S[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S[i][j] = 0.5 * (S9[i * 3 + j] + S9[j * 3 + i]);
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; sweep++) {
        const double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
        if (off <= 1e-300) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = S[p][q];
            if (fabs(apq) <= 1e-18 * (fabs(S[p][p]) + fabs(S[q][q]))) { S[p][q] = S[q][p] = 0; continue; }
            const double tau = (S[q][q] - S[p][p]) / (2 * apq);
            const double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
            const double c = 1.0 / sqrt(1 + tt * tt), sn = tt * c;
#pragma unroll
            for (int k = 0; k < 3; k++) { const double skp = S[k][p], skq = S[k][q]; S[k][p] = c * skp - sn * skq; S[k][q] = sn * skp + c * skq; }
#pragma unroll
            for (int k = 0; k < 3; k++) { const double spk = S[p][k], sqk = S[q][k]; S[p][k] = c * spk - sn * sqk; S[q][k] = sn * spk + c * sqk; }
#pragma unroll
            for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
        }
    }
    for (int i = 0; i < 3; i++) { th[i] = S[i][i]; for (int j = 0; j < 3; j++) W[i * 3 + j] = V[i][j]; }
}

// S = D Pi_K*(v) = I - D Pi_K(-v) of the exponential cone, diagonalised (columns of W).
__device__ __noinline__ void exp_dual_eig(const double *v, double *W, double *th) {
    double w[3] = {-v[0], -v[1], -v[2]}, J[9];
    exp_dproject(w, J);
    for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i];
    sym3_eig(J, W, th);
}

// ------------------------------------------------------------------------------------------------
// 3-d power cone  K_a = {(x,y,z): x^a y^(1-a) >= |z|, x,y >= 0},  K_a^* = {(u,v,w): (u/a)^a (v/(1-a))^(1-a) >= |w|, u,v >= 0}.
// Outside K and -K*, with z0 != 0, the projection is (x(r), y(r), sign(z0) r),
//     x(r) = (x0 + sqrt(x0^2 + 4 a r (|z0| - r)))/2,  y(r) = (y0 + sqrt(y0^2 + 4 (1-a) r (|z0| - r)))/2,
// r in (0, |z0|) the root of Phi(r) = x(r)^a y(r)^(1-a) - r  (Hien 2015; the formulation SCS uses).  Bracketed Newton, warm-started
// from the previous iteration's r.  A template entry a < 0 denotes the dual cone K_|a|^* (SCS convention).
__device__ __forceinline__ double pow_branch(double t0, double q) {       // (t0 + sqrt(t0^2 + 4q))/2 without cancellation for t0 < 0
    const double sq = sqrt(t0 * t0 + 4 * q);
    return t0 >= 0 ? 0.5 * (t0 + sq) : 2 * q / (sq - t0);
}
// returns the case (0 inside K, 1 inside -K*, 2 the z0 == 0 face, 3 boundary); r_io: warm start in, root out
__device__ __forceinline__ int pow_project(double &p0, double &p1, double &p2, double a, double &r_io) {
    const double x0 = p0, y0 = p1, z0 = p2, az = fabs(z0);
    if (x0 >= 0 && y0 >= 0 && pow(x0, a) * pow(y0, 1 - a) >= az) return 0;
    if (x0 <= 0 && y0 <= 0 && pow(-x0 / a, a) * pow(-y0 / (1 - a), 1 - a) >= az) { p0 = p1 = p2 = 0; return 1; }
    if (az == 0) { p0 = x0 > 0 ? x0 : 0; p1 = y0 > 0 ? y0 : 0; return 2; }
    double lo = 0, hi = az, r = r_io;
    if (!(r > lo && r < hi)) r = 0.5 * az;
    double x = 0, y = 0;
    for (int it = 0; it < 120; it++) {
        const double qq = r * (az - r);
        const double sx = sqrt(x0 * x0 + 4 * a * qq), sy = sqrt(y0 * y0 + 4 * (1 - a) * qq);
        x = x0 >= 0 ? 0.5 * (x0 + sx) : 2 * a * qq / (sx - x0);
        y = y0 >= 0 ? 0.5 * (y0 + sy) : 2 * (1 - a) * qq / (sy - y0);
        EXP_COUNT_ITER
        const double f = pow(x, a) * pow(y, 1 - a), g = f - r;
        if (g > 0) lo = r; else if (g < 0) hi = r; else break;
        const double dq = az - 2 * r;
        const double dg = f * (a * a * dq / (sx * x) + (1 - a) * (1 - a) * dq / (sy * y)) - 1;      // Phi'(r)
        double nr = r - g / dg;
        if (dg < 0 && nr > 0 && nr < az && fabs(nr - r) <= 3e-9 * fmin(nr, az - nr)) { r = nr; break; }   // quadratic convergence: error ~ step^2
        if (!(dg < 0) || !(nr > lo) || !(nr < hi)) nr = 0.5 * (lo + hi);
        const double step = fabs(nr - r);
        r = nr;
        if (step <= 4e-16 * az) break;
        if (hi - lo <= 2e-16 * az) break;
    }
    const double qq = r * (az - r);
    p0 = pow_branch(x0, a * qq); p1 = pow_branch(y0, (1 - a) * qq); p2 = z0 > 0 ? r : -r;
    r_io = r;
    return 3;
}
// projection of the solver's cone step for a template entry `a`: onto the DUAL of the entry's cone (a > 0: K_a^*, a < 0: K_|a|)
__device__ __forceinline__ void pow_project_dual_of_entry(double *v, double a, double *r_state) {
    const double al = fabs(a);
    double r = *r_state;
    if (a < 0) { double w0 = v[0], w1 = v[1], w2 = v[2]; pow_project(w0, w1, w2, al, r); v[0] = w0; v[1] = w1; v[2] = w2; }
    else {
        const double v0 = v[0], v1 = v[1], v2 = v[2];
        double w0 = -v0, w1 = -v1, w2 = -v2;
        pow_project(w0, w1, w2, al, r);
        v[0] = v0 + w0; v[1] = v1 + w1; v[2] = v2 + w2;
    }
    *r_state = r;
}
// J = D Pi_{K_a}(v): on the boundary p - v = lam grad g(p), g = x^a y^(1-a) - |z| = 0, lam = |z0| - r; implicit function theorem:
//   [[I - lam H, -grad g], [-grad g^T, 0]] [dp; dlam] = [dv; 0]       (H = Hessian of g), solved by 4x4 elimination with pivoting.
__device__ __noinline__ void pow_dproject(const double *v, double a, double *J) {
    double p0 = v[0], p1 = v[1], p2 = v[2], r = -1.0;
    const int kase = pow_project(p0, p1, p2, a, r);
    for (int i = 0; i < 9; i++) J[i] = 0;
    if (kase == 0) { J[0] = J[4] = J[8] = 1; return; }
    if (kase == 1) return;
    if (kase == 2) { J[0] = v[0] > 0 ? 1.0 : 0.0; J[4] = v[1] > 0 ? 1.0 : 0.0; return; }
    const double x = fmax(p0, 1e-100), y = fmax(p1, 1e-100), sg = v[2] > 0 ? 1.0 : -1.0, lam = fmax(fabs(v[2]) - fabs(p2), 0.0);
    const double f = pow(x, a) * pow(y, 1 - a);
    const double g[3] = {a * f / x, (1 - a) * f / y, -sg};
    const double hxy = a * (1 - a) * f / (x * y);
    double G[4][4] = {{1 - lam * a * (a - 1) * f / (x * x), -lam * hxy, 0, -g[0]},
                      {-lam * hxy, 1 + lam * a * (1 - a) * f / (y * y), 0, -g[1]},
                      {0, 0, 1, -g[2]},
                      {-g[0], -g[1], -g[2], 0}};
    double R[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int pv = c;
#pragma unroll
        for (int i = 0; i < 4; i++) if (i > c && fabs(G[i][c]) > fabs(G[pv][c])) pv = i;
#pragma unroll
        for (int i = 0; i < 4; i++) if (i == pv && pv != c) {
#pragma unroll
            for (int j = 0; j < 4; j++) { const double t = G[c][j]; G[c][j] = G[i][j]; G[i][j] = t; }
#pragma unroll
            for (int j = 0; j < 3; j++) { const double t = R[c][j]; R[c][j] = R[i][j]; R[i][j] = t; }
        }
        const double d = 1.0 / G[c][c];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i == c) continue;
            const double fct = G[i][c] * d;
#pragma unroll
            for (int j = 0; j < 4; j++) G[i][j] -= fct * G[c][j];
#pragma unroll
            for (int j = 0; j < 3; j++) R[i][j] -= fct * R[c][j];
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[i * 3 + j] = R[i][j] / G[i][i];
}
// S = D Pi onto the dual of the entry's cone, diagonalised
__device__ __noinline__ void pow_dual_eig(const double *v, double a, double *W, double *th) {
    double J[9];
    if (a < 0) pow_dproject(v, -a, J);
    else {
        double w[3] = {-v[0], -v[1], -v[2]};
        pow_dproject(w, a, J);
        for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i];
    }
    sym3_eig(J, W, th);
}
