/*
 * oracle/cone_oracle.c  --  TEST INFRASTRUCTURE ONLY (never shipped, never on the product path)
 *
 * CPU restatement, in plain C, of the arithmetic behind the reference hot path
 *     cvxpylayers.torch.CvxpyLayer.forward/backward
 *       -> cvxpylayers/interfaces/diffcp_if.py:329-403  (_CvxpyLayer.forward / backward)
 *       -> diffcp.solve_and_derivative_batch / adj_batch   (diffcp_if.py:365-371, :86)
 *       -> SCS (direct linear-system variant)
 * diffcp (pin 1.1.4, uv.lock:590-592) and SCS (pin 3.2.9, uv.lock:2352-2354) are NOT vendored in
 * /root/reference and are not installable here, so this file restates their PUBLISHED algorithms:
 *   forward : O'Donoghue, "Operator splitting for a homogeneous embedding of the linear
 *             complementarity problem" (SCS 3): Ruiz equilibration, Douglas-Rachford on the
 *             homogeneous self-dual embedding with the diagonal metric R = diag(rho_x I, 1/scale I, 10),
 *             direct factorisation of the reduced KKT system, over-relaxation alpha, adaptive
 *             scale, inf-norm termination + infeasibility certificates.
 *   backward: Agrawal et al., "Differentiating through a cone program" (diffcp):
 *             z=(x, y-s, 1), M = (Q-I) DPi(z) + I, solve M^T r = dz (LSQR, or dense elimination),
 *             dA = r_y x^T - y r_x^T (antisymmetrised outer products), db, dc.
 * PARITY PIN (round 3): the oracle IS pinned on outputs of the real cvxpylayers -> diffcp -> SCS stack: the numbers stored in the cell outputs of the
 * reference's example notebooks (/root/reference/examples/torch/{optimal_transport, lqr, tutorial, supply_chain}.ipynb), parsed into
 * tests/golden/ref_notebook_*.npz by tests/golden/make_notebook_golden.py and reproduced in tests/test_notebook_golden.py: forward values of an
 * exponential-cone layer and diffcp's gradients through it (to the printed 4 decimals), an SDP solved by SCS (to SCS's own 6e-6), a least-squares
 * layer, and eight epochs of a training trace whose every point depends on the adjoints of 100 solves (to the printed 5 digits).  Also pinned
 * (tests/test_oracle_known_answers.py, tests/test_independent_checks.py): every closed-form known-answer problem the reference's own tests assert,
 * central finite differences of the forward solve, scipy's HiGHS on LPs, conic optimality certificates.  NOT pinned: SCS's iterate sequence
 * (iteration counts are this restatement's; no reference output records them) -- e.g. the one-pair Anderson acceleration and its give-up rule.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 *
 * Conventions (reference: diffcp_if.py:59-67): solver form  min c^T x  s.t.  A x + s = b, s in K,
 * K = zero(z) x nonneg(l) x SOC(q_1) x ... x PSD(s_1) x ... x EXP^nep x POW(a_1) x ...  (SCS row order z,l,q,s,ep,p),
 * EXP = cl{(x,y,z): y > 0, y exp(x/y) <= z} (SCS / CVXPY row order inside a triple),
 * SOC = (t, x) with ||x|| <= t; PSD = lower-triangular column-major svec with sqrt(2) off-diagonals
 * (cvxpylayers/torch/cvxpylayer.py:201-222).  A is dense row-major (m x n) per instance here; the
 * python wrapper densifies the CSC template.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int z, l, nq, ns;
    const int *q;
    const int *s;
    int nep;               /* primal exponential cones (3 rows each), after the PSD blocks (SCS row order z,l,q,s,ep) */
    int np;                /* 3-d power cones  x^a y^(1-a) >= |z|, x, y >= 0  (after the exponential cones; SCS "p"), a in (0,1);  */
    const double *pw;      /* a negative entry -a means the DUAL power cone of exponent a (SCS convention)                        */
} oc_cones;

typedef struct {
    double eps_abs, eps_rel, eps_infeas, alpha, rho_x, scale;
    int max_iters, normalize, adaptive_scale;
    /* backward */
    int adj_mode;          /* 0 = lsqr (diffcp default), 1 = dense elimination */
    double lsqr_atol, lsqr_btol, lsqr_conlim;
    int lsqr_iter_lim;     /* <=0: 2*N */
    int warm_start;        /* != 0: x, y, s hold an initial point on entry (SCS warm start u = (x, y, 1), v = (0, s, 0)) */
    int aa_mem;            /* Anderson acceleration memory (SCS acceleration_lookback; 0 = off) */
    int aa_interval;       /* applied every aa_interval iterations (SCS acceleration_interval, default 10) */
} oc_opts;

enum { OC_SOLVED = 1, OC_SOLVED_INACCURATE = 2, OC_UNBOUNDED = -1, OC_INFEASIBLE = -2,
       OC_UNBOUNDED_INACCURATE = -6, OC_INFEASIBLE_INACCURATE = -7, OC_FAILED = -4 };

#define MIN_SCALE 1e-4
#define MAX_SCALE 1e4
#define NUM_RUIZ_PASSES 25
#define NUM_L2_PASSES 1
#define TAU_FACTOR 10.0
#define CONVERGED_INTERVAL 25
#define AA_MAX_REJECT 10      /* safeguard rejections after which Anderson acceleration is switched off for the instance */
#define RESCALING_MIN_ITERS 100
#define MIN_SCALE_VALUE 1e-6
#define MAX_SCALE_VALUE 1e6
#define ZERO_CONE_FACTOR 1000.0

void oc_default_opts(oc_opts *o) {
    o->eps_abs = 1e-4; o->eps_rel = 1e-4; o->eps_infeas = 1e-7; o->alpha = 1.5;
    o->rho_x = 1e-6; o->scale = 0.1; o->max_iters = 100000; o->normalize = 1;
    o->adaptive_scale = 1; o->adj_mode = 0; o->lsqr_atol = 1e-8; o->lsqr_btol = 1e-8;
    o->lsqr_conlim = 1e8; o->lsqr_iter_lim = -1; o->warm_start = 0; o->aa_mem = 0; o->aa_interval = 10;
}

static int cone_rows(const oc_cones *k) {
    int m = k->z + k->l;
    for (int i = 0; i < k->nq; i++) m += k->q[i];
    for (int i = 0; i < k->ns; i++) m += k->s[i] * (k->s[i] + 1) / 2;
    m += 3 * k->nep + 3 * k->np;
    return m;
}

/* ------------------------------------------------------------------ small dense helpers */
static double dot(const double *a, const double *b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }
static double norm2(const double *a, int n) { return sqrt(dot(a, a, n)); }
static double norm_inf(const double *a, int n) { double s = 0; for (int i = 0; i < n; i++) { double v = fabs(a[i]); if (v > s) s = v; } return s; }
/* y = A x (A m x n row-major) */
static void matvec(const double *A, const double *x, double *y, int m, int n) {
    for (int i = 0; i < m; i++) y[i] = dot(A + (size_t)i * n, x, n);
}
/* x = A^T y */
static void matvec_t(const double *A, const double *y, double *x, int m, int n) {
    memset(x, 0, sizeof(double) * n);
    for (int i = 0; i < m; i++) { const double *r = A + (size_t)i * n; double yi = y[i]; for (int j = 0; j < n; j++) x[j] += r[j] * yi; }
}

/* symmetric eigendecomposition, cyclic Jacobi.  S (k x k, row-major, destroyed) -> w (eigvals), V (columns = eigvecs) */
static void jacobi_eig(double *S, int k, double *w, double *V) {
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) V[i * k + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < k; i++) { diag += S[i * k + i] * S[i * k + i]; for (int j = i + 1; j < k; j++) off += S[i * k + j] * S[i * k + j]; }
        if (off <= 1e-32 * (diag + off) || off == 0) break;
        for (int p = 0; p < k - 1; p++) for (int q = p + 1; q < k; q++) {
            double apq = S[p * k + q];
            if (apq == 0) continue;
            double theta = (S[q * k + q] - S[p * k + p]) / (2 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
            double c = 1 / sqrt(t * t + 1), s = t * c;
            for (int r = 0; r < k; r++) { double a = S[r * k + p], b = S[r * k + q]; S[r * k + p] = c * a - s * b; S[r * k + q] = s * a + c * b; }
            for (int r = 0; r < k; r++) { double a = S[p * k + r], b = S[q * k + r]; S[p * k + r] = c * a - s * b; S[q * k + r] = s * a + c * b; }
            for (int r = 0; r < k; r++) { double a = V[r * k + p], b = V[r * k + q]; V[r * k + p] = c * a - s * b; V[r * k + q] = s * a + c * b; }
        }
    }
    for (int i = 0; i < k; i++) w[i] = S[i * k + i];
}
static void svec_to_mat(const double *v, int k, double *S) {
    int idx = 0; const double r = 1 / sqrt(2.0);
    for (int j = 0; j < k; j++) for (int i = j; i < k; i++) { double val = v[idx++]; if (i != j) val *= r; S[i * k + j] = val; S[j * k + i] = val; }
}
static void mat_to_svec(const double *S, int k, double *v) {
    int idx = 0; const double r = sqrt(2.0);
    for (int j = 0; j < k; j++) for (int i = j; i < k; i++) v[idx++] = (i == j) ? S[i * k + j] : r * 0.5 * (S[i * k + j] + S[j * k + i]);
}

/* ------------------------------------------------------------------ cone projections (onto the DUAL cone K*) */
static void proj_soc(double *v, int d) {
    if (d == 0) return;
    if (d == 1) { if (v[0] < 0) v[0] = 0; return; }
    double t = v[0], nz = norm2(v + 1, d - 1);
    if (nz <= t) return;
    if (nz <= -t) { memset(v, 0, sizeof(double) * d); return; }
    double a = 0.5 * (t + nz);
    v[0] = a; double f = a / nz;
    for (int i = 1; i < d; i++) v[i] *= f;
}
static void proj_psd(double *v, int k) {
    if (k == 0) return;
    double *S = malloc(sizeof(double) * k * k * 3 + sizeof(double) * k), *V = S + k * k, *T = V + k * k, *w = T + k * k;
    svec_to_mat(v, k, S); jacobi_eig(S, k, w, V);
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) { double a = 0; for (int r = 0; r < k; r++) if (w[r] > 0) a += V[i * k + r] * w[r] * V[j * k + r]; T[i * k + j] = a; }
    mat_to_svec(T, k, v); free(S);
}

/* ------------------------------------------------------------------ exponential cone
 * K_exp = cl{(x,y,z): y>0, y e^{x/y} <= z},  K_exp^* = cl{(u,v,w): u<0, -u e^{v/u} <= e w}.
 * Projection of v=(r,s,t) (Moreau decomposition v = p - d, p in K, d in K*, p.d = 0; cf. Friberg, "Projection onto the
 * exponential cone: a univariate root-finding problem", 2023, which SCS 3.2 follows): on the boundary
 *     p = yy (rho, 1, e^rho),   d = mu (-1, rho-1, e^-rho)        (p.d = 0 identically)
 * and  p - d = v  gives  yy = (s + r(rho-1))/D, mu = (r - s rho)/D, D = rho^2 - rho + 1, with rho the root of
 *     h(rho) = (s + r(rho-1)) e^rho - (r - s rho) e^-rho - t D
 * on the interval where yy > 0 and mu > 0 (h is increasing there).  Returns the case: 0 inside K, 1 inside -K*, 2 the
 * (r<=0, s<=0) face, 3 boundary (rho, yy, mu returned). */
typedef struct { int kase; double rho, yy, mu; } exp_info;   /* boundary case: rho <= 0: (yy, mu e^-rho) ; rho > 0: (yy e^rho, mu) */

/* h scaled by e^rho (rho <= 0) or e^-rho (rho > 0): same sign, same root, no overflow anywhere */
static double exp_h(double rho, double r, double s, double t) {
    double ny = s + r * (rho - 1), nm = r - s * rho, D = rho * rho - rho + 1;
    if (rho <= 0) { double E = exp(rho); return ny * E * E - nm - t * D * E; }
    double F = exp(-rho); return ny - nm * F * F - t * D * F;
}
static void proj_exp(double *v, exp_info *inf) {
    double r = v[0], s = v[1], t = v[2];
    exp_info loc; if (!inf) inf = &loc;
    inf->rho = inf->yy = inf->mu = 0;
    if ((s > 0 && s * exp(r / s) <= t) || (r <= 0 && s == 0 && t >= 0)) { inf->kase = 0; return; }
    if ((r > 0 && r * exp(s / r) <= -2.718281828459045235 * t) || (r == 0 && s <= 0 && t <= 0)) { inf->kase = 1; v[0] = v[1] = v[2] = 0; return; }
    if (r <= 0 && s <= 0) { inf->kase = 2; v[1] = 0; if (t < 0) v[2] = 0; return; }
    /* bracket: yy > 0 <=> s + r(rho-1) > 0, mu > 0 <=> r - s rho > 0 */
    double lo, hi; int lo_inf = 0, hi_inf = 0;
    if (r > 0 && s > 0) { lo = 1 - s / r; hi = r / s; }
    else if (r > 0) { lo = 1 - s / r; hi = 0; hi_inf = 1; }              /* s <= 0 */
    else { hi = r / s; lo = 0; lo_inf = 1; }                             /* r <= 0, s > 0 */
    if (hi_inf) { double st = 1; hi = lo + st; while (exp_h(hi, r, s, t) < 0 && st < 1e15) { st *= 2; hi = lo + st; } }
    if (lo_inf) { double st = 1; lo = hi - st; while (exp_h(lo, r, s, t) > 0 && st < 1e15) { st *= 2; lo = hi - st; } }
    double rho = 0.5 * (lo + hi);
    for (int it = 0; it < 300; it++) {
        rho = 0.5 * (lo + hi);
        if (exp_h(rho, r, s, t) > 0) hi = rho; else lo = rho;
        if (hi - lo <= 1e-16 * (1 + fabs(rho))) break;
    }
    double D = rho * rho - rho + 1, ny = s + r * (rho - 1), nm = r - s * rho;
    if (ny < 0) ny = 0; if (nm < 0) nm = 0;
    inf->kase = 3; inf->rho = rho;
    /* two algebraically equal forms; each is the accurate one on its side (the other multiplies a cancelled numerator by a huge exponential) */
    if (rho <= 0) {
        double E = exp(rho), yy = ny / D;
        v[0] = yy * rho; v[1] = yy; v[2] = yy * E;
        inf->yy = yy; inf->mu = v[2] - t;                      /* mu e^-rho, from the third equation */
    } else {
        double F = exp(-rho), mu = nm / D;
        v[0] = r - mu; v[1] = s + mu * (rho - 1); v[2] = t + mu * F; if (v[1] < 0) v[1] = 0;
        inf->mu = mu; inf->yy = v[2];                          /* yy e^rho */
    }
}
static void inv3(const double *G, double *inv) {
    double c00 = G[4] * G[8] - G[5] * G[7], c01 = G[5] * G[6] - G[3] * G[8], c02 = G[3] * G[7] - G[4] * G[6];
    double det = G[0] * c00 + G[1] * c01 + G[2] * c02;
    double t[9] = { c00, G[2] * G[7] - G[1] * G[8], G[1] * G[5] - G[2] * G[4],
                    c01, G[0] * G[8] - G[2] * G[6], G[2] * G[3] - G[0] * G[5],
                    c02, G[1] * G[6] - G[0] * G[7], G[0] * G[4] - G[1] * G[3] };
    for (int i = 0; i < 9; i++) inv[i] = t[i] / det;
}
/* J (3x3 row-major) = D Pi_{K_exp}(v).  Boundary case: the projection is p = Y a(rho), p - v = d = M b(rho), where
 *   rho <= 0:  a = (rho, 1, E),  b = (-E, (rho-1) E, 1),  E = e^rho    (Y = yy,  M = mu / E)
 *   rho  > 0:  a = (rho F, F, 1), b = (-1, rho-1, F),     F = e^-rho   (Y = yy / F, M = mu)
 * so G(Y, M, rho) = Y a - M b = v,  dG = [a | -b | Y a' - M b'],  dp = a dY + Y a' drho  =>  J = [a | 0 | Y a'] dG^-1. */
static void dproj_exp(const double *v, double *J) {
    double w[3] = { v[0], v[1], v[2] }; exp_info inf;
    proj_exp(w, &inf);
    memset(J, 0, 9 * sizeof(double));
    if (inf.kase == 0) { J[0] = J[4] = J[8] = 1; return; }
    if (inf.kase == 1) return;
    if (inf.kase == 2) { J[0] = 1; J[8] = v[2] > 0 ? 1 : 0; return; }
    double rho = inf.rho, Y = inf.yy, M = inf.mu, a[3], b[3], da[3], db[3];
    if (rho < -690) { J[0] = J[4] = 1; return; }      /* e^rho underflows: p = (r, s, 0), the flat face */
    if (rho > 690) { J[8] = 1; return; }              /* e^-rho underflows: p = (0, 0, t) */
    if (rho <= 0) { double E = exp(rho); a[0] = rho; a[1] = 1; a[2] = E; da[0] = 1; da[1] = 0; da[2] = E;
                    b[0] = -E; b[1] = (rho - 1) * E; b[2] = 1; db[0] = -E; db[1] = rho * E; db[2] = 0; }
    else { double F = exp(-rho); a[0] = rho * F; a[1] = F; a[2] = 1; da[0] = (1 - rho) * F; da[1] = -F; da[2] = 0;
           b[0] = -1; b[1] = rho - 1; b[2] = F; db[0] = 0; db[1] = 1; db[2] = -F; }
    double G[9], inv[9];
    for (int i = 0; i < 3; i++) { G[i * 3] = a[i]; G[i * 3 + 1] = -b[i]; G[i * 3 + 2] = Y * da[i] - M * db[i]; }
    inv3(G, inv);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[i * 3 + j] = a[i] * inv[j] + Y * da[i] * inv[6 + j];
}
/* dual cone by Moreau:  Pi_{K*}(v) = v + Pi_K(-v),   D Pi_{K*}(v) = I - D Pi_K(-v) */
static void proj_exp_dual(double *v) {
    double w[3] = { -v[0], -v[1], -v[2] };
    proj_exp(w, NULL);
    for (int i = 0; i < 3; i++) v[i] += w[i];
}
static void dproj_exp_dual(const double *v, double *J) {
    double w[3] = { -v[0], -v[1], -v[2] };
    dproj_exp(w, J);
    for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i];
}
/* test hooks (ctypes): which = 0 primal, 1 dual */
void oc_proj_exp(double *v, int which) { if (which) proj_exp_dual(v); else proj_exp(v, NULL); }
void oc_dproj_exp(const double *v, int which, double *J) { if (which) dproj_exp_dual(v, J); else dproj_exp(v, J); }


/* ------------------------------------------------------------------ 3-d power cone
 * K_a = {(x,y,z): x^a y^(1-a) >= |z|, x,y >= 0},  K_a^* = {(u,v,w): (u/a)^a (v/(1-a))^(1-a) >= |w|, u,v >= 0}.
 * Projection (Hien, "Differential properties of Euclidean projection onto power cone", 2015; the formulation SCS uses):
 * outside K and -K*, with z0 != 0, the projection is (x(r), y(r), sign(z0) r) with
 *     x(r) = (x0 + sqrt(x0^2 + 4 a r (|z0| - r)))/2,  y(r) = (y0 + sqrt(y0^2 + 4 (1-a) r (|z0| - r)))/2
 * and r in (0, |z0|) the root of Phi(r) = x(r)^a y(r)^(1-a) - r  (Phi(0) >= 0 > Phi(|z0|)).   Bisection here. */
/* (t0 + sqrt(t0^2 + 4 q))/2 for q >= 0 without cancellation when t0 < 0 */
static double pow_branch(double t0, double q) {
    double sq = sqrt(t0 * t0 + 4 * q);
    return t0 >= 0 ? 0.5 * (t0 + sq) : 2 * q / (sq - t0);
}
static int proj_pow(double *v, double a) {     /* returns the case: 0 inside, 1 inside -K*, 2 z0 == 0 face, 3 boundary */
    double x0 = v[0], y0 = v[1], z0 = v[2], az = fabs(z0);
    if (x0 >= 0 && y0 >= 0 && pow(x0, a) * pow(y0, 1 - a) >= az) return 0;
    if (x0 <= 0 && y0 <= 0 && pow(-x0 / a, a) * pow(-y0 / (1 - a), 1 - a) >= az) { v[0] = v[1] = v[2] = 0; return 1; }
    if (az == 0) { v[0] = x0 > 0 ? x0 : 0; v[1] = y0 > 0 ? y0 : 0; return 2; }
    double lo = 0, hi = az, r = 0.5 * az, x = 0, y = 0;
    for (int it = 0; it < 200; it++) {
        r = 0.5 * (lo + hi);
        x = pow_branch(x0, a * r * (az - r)); y = pow_branch(y0, (1 - a) * r * (az - r));
        if (pow(x, a) * pow(y, 1 - a) - r > 0) lo = r; else hi = r;
        if (hi - lo <= 1e-16 * az) break;
    }
    v[0] = x; v[1] = y; v[2] = z0 > 0 ? r : -r;
    return 3;
}
/* solve the 4x4 system G X = RHS (3 right-hand sides) by Gaussian elimination with partial pivoting */
static void solve4(double G[4][4], double R[4][3]) {
    for (int c = 0; c < 4; c++) {
        int p = c; for (int i = c + 1; i < 4; i++) if (fabs(G[i][c]) > fabs(G[p][c])) p = i;
        if (p != c) { for (int j = 0; j < 4; j++) { double t = G[c][j]; G[c][j] = G[p][j]; G[p][j] = t; } for (int j = 0; j < 3; j++) { double t = R[c][j]; R[c][j] = R[p][j]; R[p][j] = t; } }
        double d = G[c][c];
        for (int i = 0; i < 4; i++) { if (i == c) continue; double f = G[i][c] / d; for (int j = 0; j < 4; j++) G[i][j] -= f * G[c][j]; for (int j = 0; j < 3; j++) R[i][j] -= f * R[c][j]; }
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) R[i][j] /= G[i][i];
}
/* J = D Pi_{K_a}(v): on the boundary p - v = lam grad g(p), g = x^a y^(1-a) - |z| = 0, lam = |z0| - r; implicit function:
 *   [[I - lam H, -grad g], [-grad g^T, 0]] [dp; dlam] = [dv; 0]      (H = Hessian of g) */
static void dproj_pow(const double *v, double a, double *J) {
    double p[3] = { v[0], v[1], v[2] };
    int kase = proj_pow(p, a);
    memset(J, 0, 9 * sizeof(double));
    if (kase == 0) { J[0] = J[4] = J[8] = 1; return; }
    if (kase == 1) return;
    if (kase == 2) { J[0] = v[0] > 0; J[4] = v[1] > 0; return; }
    double x = p[0], y = p[1], sg = v[2] > 0 ? 1.0 : -1.0, r = fabs(p[2]), lam = fabs(v[2]) - r;
    if (x < 1e-100) x = 1e-100;       /* keeps the curvature terms finite on the nearly flat parts of the boundary (a near 0 or 1) */
    if (y < 1e-100) y = 1e-100;
    if (lam < 0) lam = 0;
    double f = pow(x, a) * pow(y, 1 - a);
    double g[3] = { a * f / x, (1 - a) * f / y, -sg };
    double H[3][3] = { { a * (a - 1) * f / (x * x), a * (1 - a) * f / (x * y), 0 }, { a * (1 - a) * f / (x * y), -a * (1 - a) * f / (y * y), 0 }, { 0, 0, 0 } };
    double G[4][4], R[4][3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) { G[i][j] = (i == j ? 1.0 : 0.0) - lam * H[i][j]; R[i][j] = (i == j) ? 1.0 : 0.0; } G[i][3] = -g[i]; G[3][i] = -g[i]; }
    G[3][3] = 0; R[3][0] = R[3][1] = R[3][2] = 0;
    solve4(G, R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[i * 3 + j] = R[i][j];
}
/* entry a > 0: primal cone K_a; a < 0: the dual cone K_|a|^*.   which = 0: project onto that cone, 1: onto ITS dual. */
static void proj_pow_entry(double *v, double a, int onto_dual) {
    int dualcone = (a < 0) != (onto_dual != 0);
    double al = fabs(a);
    if (!dualcone) { proj_pow(v, al); return; }
    double w[3] = { -v[0], -v[1], -v[2] };
    proj_pow(w, al);
    for (int i = 0; i < 3; i++) v[i] += w[i];
}
static void dproj_pow_entry(const double *v, double a, int onto_dual, double *J) {
    int dualcone = (a < 0) != (onto_dual != 0);
    double al = fabs(a);
    if (!dualcone) { dproj_pow(v, al, J); return; }
    double w[3] = { -v[0], -v[1], -v[2] };
    dproj_pow(w, al, J);
    for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) - J[i];
}
void oc_proj_pow(double *v, double a, int onto_dual) { proj_pow_entry(v, a, onto_dual); }
void oc_dproj_pow(const double *v, double a, int onto_dual, double *J) { dproj_pow_entry(v, a, onto_dual, J); }

/* y <- Pi_{K*}(y): zero cone K={0} has K* = R^z (free) */
static void proj_dual_cone(double *y, const oc_cones *k) {
    int off = k->z;
    for (int i = 0; i < k->l; i++) if (y[off + i] < 0) y[off + i] = 0;
    off += k->l;
    for (int c = 0; c < k->nq; c++) { proj_soc(y + off, k->q[c]); off += k->q[c]; }
    for (int c = 0; c < k->ns; c++) { proj_psd(y + off, k->s[c]); off += k->s[c] * (k->s[c] + 1) / 2; }
    for (int c = 0; c < k->nep; c++) { proj_exp_dual(y + off); off += 3; }
    for (int c = 0; c < k->np; c++) { proj_pow_entry(y + off, k->pw[c], 1); off += 3; }
}

/* ------------------------------------------------------------------ Cholesky of SPD n x n (row-major, lower) */
static int chol_factor(double *S, int n) {
    for (int j = 0; j < n; j++) {
        double d = S[j * n + j];
        for (int k = 0; k < j; k++) d -= S[j * n + k] * S[j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d); S[j * n + j] = d;
        for (int i = j + 1; i < n; i++) { double v = S[i * n + j]; for (int k = 0; k < j; k++) v -= S[i * n + k] * S[j * n + k]; S[i * n + j] = v / d; }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *x) {
    for (int i = 0; i < n; i++) { double v = x[i]; for (int k = 0; k < i; k++) v -= L[i * n + k] * x[k]; x[i] = v / L[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= L[k * n + i] * x[k]; x[i] = v / L[i * n + i]; }
}

/* ------------------------------------------------------------------ forward solve, one instance */
typedef struct { int iters, status; double pobj, dobj, res_pri, res_dual, gap, scale; int n_rescale; } oc_info;

/* average the row scalings inside each SOC / PSD block so the scaled cone is still the cone */
static void block_average(double *D, const oc_cones *k) {
    int off = k->z + k->l;
    for (int c = 0; c < k->nq; c++) { int d = k->q[c]; if (d > 0) { double s = 0; for (int i = 0; i < d; i++) s += D[off + i]; s /= d; for (int i = 0; i < d; i++) D[off + i] = s; } off += d; }
    for (int c = 0; c < k->ns; c++) { int d = k->s[c] * (k->s[c] + 1) / 2; if (d > 0) { double s = 0; for (int i = 0; i < d; i++) s += D[off + i]; s /= d; for (int i = 0; i < d; i++) D[off + i] = s; } off += d; }
    for (int c = 0; c < k->nep + k->np; c++) { double s = (D[off] + D[off + 1] + D[off + 2]) / 3; D[off] = D[off + 1] = D[off + 2] = s; off += 3; }
}
static double clamp_scale(double v) { if (v < MIN_SCALE) return 1.0; if (v > MAX_SCALE) return MAX_SCALE; return v; }

static void set_ry(double *ry, int m, const oc_cones *k, double scale) {
    for (int i = 0; i < m; i++) ry[i] = (i < k->z) ? 1.0 / (ZERO_CONE_FACTOR * scale) : 1.0 / scale;
}

/* factor S = rho_x I + P + A^T diag(1/ry) A  (P may be NULL); returns 0 ok */
static int factor_kkt(const double *A, const double *Pm, const double *ry, double rho_x, int m, int n, double *L) {
    memset(L, 0, sizeof(double) * n * n);
    if (Pm) for (int a = 0; a < n; a++) for (int b2 = 0; b2 <= a; b2++) L[a * n + b2] = 0.5 * (Pm[a * n + b2] + Pm[b2 * n + a]);
    for (int i = 0; i < m; i++) { const double *r = A + (size_t)i * n; double w = 1.0 / ry[i];
        for (int a = 0; a < n; a++) { double ra = r[a] * w; if (ra == 0) continue; for (int b2 = 0; b2 <= a; b2++) L[a * n + b2] += ra * r[b2]; } }
    for (int a = 0; a < n; a++) { L[a * n + a] += rho_x; for (int b2 = a + 1; b2 < n; b2++) L[a * n + b2] = L[b2 * n + a]; }
    return chol_factor(L, n);
}
/* solve [[rho_x I, A^T],[A, -R_y]] (x,y) = (a, bb):  x = S^{-1}(a + A^T (bb/ry)),  y = (A x - bb)/ry  */
static void kkt_solve(const double *A, const double *L, const double *ry, int m, int n, const double *a, const double *bb, double *x, double *y, double *tmp_m) {
    for (int i = 0; i < m; i++) tmp_m[i] = bb[i] / ry[i];
    matvec_t(A, tmp_m, x, m, n);
    for (int j = 0; j < n; j++) x[j] += a[j];
    chol_solve(L, n, x);
    matvec(A, x, y, m, n);
    for (int i = 0; i < m; i++) y[i] = (y[i] - bb[i]) / ry[i];
}

/* P0: optional quadratic objective 1/2 x^T P x (dense n x n, symmetric PSD; SCS 3's QP extension of the embedding:
 * Q(x, y, tau) = (P x + A^T y + c tau, -A x + b tau, -(1/tau) x^T P x - c^T x - b^T y)) */
static int solve_one(int n, int m, const double *A0, const double *b0, const double *c0, const double *P0, const oc_cones *K, const oc_opts *o,
                     double *xo, double *yo, double *so, oc_info *info) {
    int l = n + m + 1;
    size_t szA = (size_t)m * n;
    double *buf = calloc(szA + 2 * (size_t)n * n + 16 * (size_t)l + 6 * (size_t)(m + n) + 64, sizeof(double));
    if (!buf) return OC_FAILED;
    double *A = buf, *L = A + szA, *Pbuf = L + (size_t)n * n, *p = Pbuf + (size_t)n * n;
    double *Pm = P0 ? Pbuf : NULL;
    if (P0) memcpy(Pm, P0, sizeof(double) * n * n);
    double *b = p; p += m; double *c = p; p += n; double *D = p; p += m; double *E = p; p += n;
    double *ry = p; p += m; double *g = p; p += l; double *w = p; p += l; double *ut = p; p += l; double *u = p; p += l;
    double *rsk = p; p += l; double *pp = p; p += l; double *t1 = p; p += l; double *t2 = p; p += l; double *t3 = p; p += l;
    double *Dt = p; p += m; double *Et = p; p += n; double *xs = p; p += n; double *ys = p; p += m; double *ss = p; p += m;
    double *Pg = p; p += n; double *Pv = p; p += n;
    memcpy(A, A0, sizeof(double) * szA); memcpy(b, b0, sizeof(double) * m); memcpy(c, c0, sizeof(double) * n);
    for (int i = 0; i < m; i++) D[i] = 1; for (int j = 0; j < n; j++) E[j] = 1;
    double sigma = 1.0;
    /* ---- equilibration (SCS normalize: NUM_RUIZ_PASSES inf-norm passes + NUM_L2_PASSES 2-norm pass) */
    if (o->normalize) {
        for (int pass = 0; pass < NUM_RUIZ_PASSES + NUM_L2_PASSES; pass++) {
            int l2 = pass >= NUM_RUIZ_PASSES;
            for (int i = 0; i < m; i++) { const double *r = A + (size_t)i * n; Dt[i] = l2 ? norm2(r, n) : norm_inf(r, n); }
            for (int j = 0; j < n; j++) Et[j] = 0;
            for (int i = 0; i < m; i++) { const double *r = A + (size_t)i * n; for (int j = 0; j < n; j++) { if (l2) Et[j] += r[j] * r[j]; else { double v = fabs(r[j]); if (v > Et[j]) Et[j] = v; } } }
            if (Pm) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {       /* columns of [P; A] */
                double v = Pm[i * n + j]; if (l2) Et[j] += v * v; else if (fabs(v) > Et[j]) Et[j] = fabs(v); }
            if (l2) for (int j = 0; j < n; j++) Et[j] = sqrt(Et[j]);
            block_average(Dt, K);
            for (int i = 0; i < m; i++) Dt[i] = 1.0 / sqrt(clamp_scale(Dt[i]));
            for (int j = 0; j < n; j++) Et[j] = 1.0 / sqrt(clamp_scale(Et[j]));
            for (int i = 0; i < m; i++) { double *r = A + (size_t)i * n; for (int j = 0; j < n; j++) r[j] *= Dt[i] * Et[j]; D[i] *= Dt[i]; }
            if (Pm) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Pm[i * n + j] *= Et[i] * Et[j];
            for (int j = 0; j < n; j++) E[j] *= Et[j];
        }
        for (int i = 0; i < m; i++) b[i] *= D[i];
        for (int j = 0; j < n; j++) c[j] *= E[j];
        double nb = norm_inf(b, m), nc = norm_inf(c, n), mx = nb > nc ? nb : nc;
        sigma = 1.0 / clamp_scale(mx);
        for (int i = 0; i < m; i++) b[i] *= sigma;
        for (int j = 0; j < n; j++) c[j] *= sigma;
    }
    double nrm_b0 = norm_inf(b0, m), nrm_c0 = norm_inf(c0, n);
    double scale = o->scale, rtau = TAU_FACTOR, rho_x = o->rho_x;
    set_ry(ry, m, K, scale);
    if (factor_kkt(A, Pm, ry, rho_x, m, n, L)) { free(buf); info->status = OC_FAILED; return OC_FAILED; }
    /* g = (R_z + M_zz)^{-1} h, h=(c,b):  kkt rhs (c, -b) */
    for (int i = 0; i < m; i++) t1[i] = -b[i];
    kkt_solve(A, L, ry, m, n, c, t1, g, g + n, t2);
    double hg = dot(c, g, n) + dot(b, g + n, m), gPg = 0;
    if (Pm) { matvec(Pm, g, Pg, n, n); gPg = dot(g, Pg, n); }
    /* cold start */
    memset(w, 0, sizeof(double) * l); w[l - 1] = 1.0;
    if (o->warm_start) {   /* the fixed point of the iteration map has w = u + R^-1 v;  x^ = sigma x / E, y^ = sigma y / D, s^ = sigma D s */
        int ok = 1;
        for (int j = 0; j < n; j++) if (!(fabs(xo[j]) < 1e300)) ok = 0;
        for (int i = 0; i < m; i++) if (!(fabs(yo[i]) < 1e300) || !(fabs(so[i]) < 1e300)) ok = 0;
        if (ok) {
            for (int j = 0; j < n; j++) w[j] = sigma * xo[j] / E[j];
            for (int i = 0; i < m; i++) w[n + i] = sigma * yo[i] / D[i] + sigma * D[i] * so[i] / ry[i];
        }
    }
    int status = 0, iter, last_scale_iter = 0, n_rescale = 0; double sum_log = 0; int n_log = 0;
    double res_pri = NAN, res_dual = NAN, gap = NAN, pobj = NAN, dobj = NAN;
    /* ---- Anderson acceleration of the fixed-point map w -> F(w) (type I, as in SCS 3: Zhang, O'Donoghue, Boyd, "Globally
     * convergent type-I Anderson acceleration for nonsmooth fixed-point iterations"): every aa_interval iterations the pair
     * (x = input of the last iteration, f = its output) extends the secant history S = [dx], Y = [dg], D = [df], g = x - f;
     * gamma = (S^T Y + r I)^-1 S^T g, r = 1e-8 |S|_F |Y|_F, and the iterate is replaced by f - D gamma.  Safeguard: if the residual
     * of the map at the accelerated point exceeds the residual before, the step is undone and the history dropped.  The history
     * is also dropped on a rescale (the map changes) and scaled with w when w is renormalised (the map is homogeneous).
     * Robustness rule (not in SCS): after AA_MAX_REJECT safeguard rejections the acceleration is switched off for the instance -- on slowly
     * converging LPs most steps of a short history are rejected and the accepted ones do harm (random LP n=50 m=100: 1716 iterations plain,
     * 5496 and 12 % unsolved at 20000 with memory 1 and no cap, 1805 with the cap; configurations where AA helps never reach the cap). */
    const int aa_mem = o->aa_mem > 0 ? o->aa_mem : 0, aa_int = o->aa_interval > 0 ? o->aa_interval : 10;
    double *aab = aa_mem > 0 ? calloc((size_t)(3 * aa_mem + 7) * l + (size_t)aa_mem * (aa_mem + 2), sizeof(double)) : NULL;
    double *aS = aab, *aY = aS ? aS + (size_t)aa_mem * l : NULL, *aD = aS ? aY + (size_t)aa_mem * l : NULL;
    double *aXp = aS ? aD + (size_t)aa_mem * l : NULL, *aFp = aS ? aXp + l : NULL, *aGp = aS ? aFp + l : NULL, *wprev = aS ? aGp + l : NULL,
           *aFsave = aS ? wprev + l : NULL, *aXsave = aS ? aFsave + l : NULL, *aG = aS ? aXsave + l : NULL, *aM = aS ? aG + l : NULL;
    int aa_iter = 0, aa_pending = 0, aa_rej = 0, aa_live = 1, aa_stale = 0; double aa_normg = 0;
    for (iter = 0; iter < o->max_iters; iter++) {
        int check = (iter % CONVERGED_INTERVAL) == 0;
        if (aa_mem > 0 && aa_pending) {      /* safeguard: residual of the map at the accelerated point */
            double dn = 0; for (int i = 0; i < l; i++) { double d = wprev[i] - w[i]; dn += d * d; } dn = sqrt(dn);
            if (!(dn <= aa_normg)) { memcpy(w, aFsave, sizeof(double) * l); memcpy(wprev, aXsave, sizeof(double) * l); aa_iter = 0; if (++aa_rej >= AA_MAX_REJECT) aa_live = 0; }
            aa_pending = 0;
        }
        if (aa_mem > 0 && aa_live && iter > 0 && iter % aa_int == 0 && !aa_stale) {      /* aa_stale: wprev predates a rescale (possible only with intervals that put a step right behind a check iteration) */
            for (int i = 0; i < l; i++) aG[i] = wprev[i] - w[i];          /* x = wprev, f = w */
            if (aa_iter > 0) {
                int idx = (aa_iter - 1) % aa_mem;
                for (int i = 0; i < l; i++) { aS[(size_t)idx * l + i] = wprev[i] - aXp[i]; aY[(size_t)idx * l + i] = aG[i] - aGp[i]; aD[(size_t)idx * l + i] = w[i] - aFp[i]; }
            }
            memcpy(aXp, wprev, sizeof(double) * l); memcpy(aFp, w, sizeof(double) * l); memcpy(aGp, aG, sizeof(double) * l);
            if (aa_iter > 0) {
                int len = aa_iter < aa_mem ? aa_iter : aa_mem;
                double *M = aM, *rhs = aM + (size_t)aa_mem * aa_mem, *gam = rhs + aa_mem;
                double ns = 0, ny = 0;
                for (int a = 0; a < len; a++) { ns += dot(aS + (size_t)a * l, aS + (size_t)a * l, l); ny += dot(aY + (size_t)a * l, aY + (size_t)a * l, l); }
                for (int a = 0; a < len; a++) { for (int b2 = 0; b2 < len; b2++) M[a * len + b2] = dot(aS + (size_t)a * l, aY + (size_t)b2 * l, l); rhs[a] = dot(aS + (size_t)a * l, aG, l); }
                double reg = 1e-8 * sqrt(ns) * sqrt(ny);
                for (int a = 0; a < len; a++) M[a * len + a] += reg;
                int ok = 1;
                for (int c2 = 0; c2 < len && ok; c2++) {      /* Gaussian elimination, partial pivoting */
                    int pv = c2; for (int r = c2 + 1; r < len; r++) if (fabs(M[r * len + c2]) > fabs(M[pv * len + c2])) pv = r;
                    if (!(fabs(M[pv * len + c2]) > 1e-300)) { ok = 0; break; }
                    if (pv != c2) { for (int j = 0; j < len; j++) { double t = M[c2 * len + j]; M[c2 * len + j] = M[pv * len + j]; M[pv * len + j] = t; } double t = rhs[c2]; rhs[c2] = rhs[pv]; rhs[pv] = t; }
                    for (int r = c2 + 1; r < len; r++) { double f = M[r * len + c2] / M[c2 * len + c2]; for (int j = c2; j < len; j++) M[r * len + j] -= f * M[c2 * len + j]; rhs[r] -= f * rhs[c2]; }
                }
                double gn = 0;
                if (ok) for (int r = len - 1; r >= 0; r--) { double v = rhs[r]; for (int j = r + 1; j < len; j++) v -= M[r * len + j] * gam[j]; gam[r] = v / M[r * len + r]; gn += gam[r] * gam[r]; }
                if (!ok || !(sqrt(gn) < 1e10)) aa_iter = 0;
                else {
                    memcpy(aFsave, w, sizeof(double) * l); memcpy(aXsave, wprev, sizeof(double) * l);
                    aa_normg = norm2(aG, l);
                    for (int a = 0; a < len; a++) for (int i = 0; i < l; i++) w[i] -= gam[a] * aD[(size_t)a * l + i];
                    aa_pending = 1;
                }
            }
            aa_iter++;
        }
        if (check && iter > 0) { /* keep the homogeneous iterate in range (iteration map is positively homogeneous) */
            double nw = norm2(w, l); if (nw > 0) { double f = sqrt((double)l) / nw; for (int i = 0; i < l; i++) w[i] *= f;
                if (aa_mem > 0) { for (size_t i = 0; i < (size_t)(3 * aa_mem + 7) * l; i++) aab[i] *= f; aa_normg *= f; } }
        }
        if (aa_mem > 0) { memcpy(wprev, w, sizeof(double) * l); aa_stale = 0; }
        /* (1) linear-system step: p = (R_z+M_zz)^{-1} R_z w_z : kkt rhs (rho_x w_x, -r_y w_y) */
        for (int j = 0; j < n; j++) t1[j] = rho_x * w[j];
        for (int i = 0; i < m; i++) t2[i] = -ry[i] * w[n + i];
        kkt_solve(A, L, ry, m, n, t1, t2, pp, pp + n, t3);
        double tau_t;
        if (Pm) {   /* positive root of (r_tau + h.g - g^T P g) t^2 + (-(r_tau w_tau + h.p) + 2 p^T P g) t - p^T P p = 0 */
            matvec(Pm, pp, Pv, n, n);
            double qa = rtau + hg - gPg, qb = -(rtau * w[l - 1] + dot(c, pp, n) + dot(b, pp + n, m)) + 2 * dot(pp, Pg, n), qc = -dot(pp, Pv, n);
            tau_t = (-qb + sqrt(fmax(qb * qb - 4 * qa * qc, 0.0))) / (2 * qa);
        } else tau_t = (rtau * w[l - 1] + dot(c, pp, n) + dot(b, pp + n, m)) / (rtau + hg);
        for (int i = 0; i < l - 1; i++) ut[i] = pp[i] - tau_t * g[i];
        ut[l - 1] = tau_t;
        /* (2) cone step */
        for (int i = 0; i < l; i++) u[i] = 2 * ut[i] - w[i];
        proj_dual_cone(u + n, K);
        if (u[l - 1] < 0) u[l - 1] = 0;
        /* (3) (s, kappa) = R (u + w - 2 ut) */
        for (int j = 0; j < n; j++) rsk[j] = rho_x * (u[j] + w[j] - 2 * ut[j]);
        for (int i = 0; i < m; i++) rsk[n + i] = ry[i] * (u[n + i] + w[n + i] - 2 * ut[n + i]);
        rsk[l - 1] = rtau * (u[l - 1] + w[l - 1] - 2 * ut[l - 1]);
        /* (4) termination test on un-normalised residuals (inf-norms) */
        if (check) {
            double tau = fabs(u[l - 1]), kap = fabs(rsk[l - 1]);
            /* un-normalise: x = E xh / sigma, y = D yh / sigma, s = sh / (D sigma) */
            for (int j = 0; j < n; j++) xs[j] = E[j] * u[j] / sigma;
            for (int i = 0; i < m; i++) { ys[i] = D[i] * u[n + i] / sigma; ss[i] = rsk[n + i] / (D[i] * sigma); }
            matvec(A0, xs, t1, m, n);           /* Ax */
            matvec_t(A0, ys, t2, m, n);         /* A^T y */
            double nax = norm_inf(t1, m), ns = norm_inf(ss, m), naty = norm_inf(t2, n);
            double rp = 0, rd = 0, naxs = 0;
            for (int i = 0; i < m; i++) { double v = t1[i] + ss[i]; if (fabs(v) > naxs) naxs = fabs(v); v -= b0[i] * tau; if (fabs(v) > rp) rp = fabs(v); }
            for (int j = 0; j < n; j++) { double v = t2[j] + c0[j] * tau; if (fabs(v) > rd) rd = fabs(v); }
            double ctx = dot(c0, xs, n), bty = dot(b0, ys, m), xPx = 0, nPx = 0;
            if (P0) {   /* dual residual P x + A^T y + c tau, gap x^T P x + c^T x + b^T y  (x, y homogeneous: x / tau is the point) */
                matvec(P0, xs, Pv, n, n);
                xPx = dot(xs, Pv, n); nPx = norm_inf(Pv, n);
                rd = 0; for (int j = 0; j < n; j++) { double v = Pv[j] + t2[j] + c0[j] * tau; if (fabs(v) > rd) rd = fabs(v); }
            }
            (void)kap;
            if (tau > 0) {
                double xPxt = xPx / tau;
                res_pri = rp / tau; res_dual = rd / tau; gap = fabs(xPxt + ctx + bty) / tau; pobj = (0.5 * xPxt + ctx) / tau; dobj = (-0.5 * xPxt - bty) / tau;
                double prl = fmax(fmax(nrm_b0 * tau, ns), nax) / tau, drl = fmax(fmax(nrm_c0 * tau, naty), nPx) / tau, grl = fmax(fmax(fabs(ctx), fabs(bty)), fabs(xPxt)) / tau;
                if (res_pri <= o->eps_abs + o->eps_rel * prl && res_dual <= o->eps_abs + o->eps_rel * drl && gap <= o->eps_abs + o->eps_rel * grl) { status = OC_SOLVED; break; }
            }
            if (bty < 0 && naty / (-bty) <= o->eps_infeas) { status = OC_INFEASIBLE; break; }
            if (ctx < 0 && fmax(naxs, nPx) / (-ctx) <= o->eps_infeas) { status = OC_UNBOUNDED; break; }
            /* (5) adaptive scale (SCS 3 heuristic: running geometric mean of relative residual ratio) */
            if (o->adaptive_scale && iter > 0) {
                double dp = fmax(fmax(nax, ns), nrm_b0 * tau), dd = fmax(fmax(naty, nrm_c0 * tau), nPx);
                double rel_p = rp / (dp > 0 ? dp : 1), rel_d = rd / (dd > 0 ? dd : 1);
                if (rel_p > 0 && rel_d > 0 && isfinite(rel_p) && isfinite(rel_d)) {
                    sum_log += log(rel_p) - log(rel_d); n_log++;
                    double factor = sqrt(exp(sum_log / n_log));
                    if (iter - last_scale_iter >= RESCALING_MIN_ITERS) {
                        double ns2 = fmin(fmax(scale * factor, MIN_SCALE_VALUE), MAX_SCALE_VALUE);
                        if (ns2 != scale && (factor > sqrt(10.0) || factor < 1.0 / sqrt(10.0))) {
                            sum_log = 0; n_log = 0; last_scale_iter = iter; scale = ns2; n_rescale++; aa_iter = 0; aa_pending = 0; aa_stale = 1;
                            set_ry(ry, m, K, scale);
                            if (factor_kkt(A, Pm, ry, rho_x, m, n, L)) { status = OC_FAILED; break; }
                            for (int i = 0; i < m; i++) t1[i] = -b[i];
                            kkt_solve(A, L, ry, m, n, c, t1, g, g + n, t2);
                            hg = dot(c, g, n) + dot(b, g + n, m);
                            if (Pm) { matvec(Pm, g, Pg, n, n); gPg = dot(g, Pg, n); }
                            /* keep (s,kappa): R+ (w+ + u - 2ut) = rsk */
                            for (int i = 0; i < m; i++) w[n + i] = rsk[n + i] / ry[i] + 2 * ut[n + i] - u[n + i];
                        }
                    }
                }
            }
        }
        /* (6) relaxed dual update */
        for (int i = 0; i < l; i++) w[i] += o->alpha * (u[i] - ut[i]);
    }
    double tau = fabs(u[l - 1]), kap = fabs(rsk[l - 1]);
    if (status == 0) { /* ran out of iterations: SCS set_unfinished */
        for (int j = 0; j < n; j++) xs[j] = E[j] * u[j] / sigma;
        for (int i = 0; i < m; i++) ys[i] = D[i] * u[n + i] / sigma;
        double ctx = dot(c0, xs, n), bty = dot(b0, ys, m);
        if (tau > kap) status = OC_SOLVED_INACCURATE; else if (bty < ctx) status = OC_INFEASIBLE_INACCURATE; else status = OC_UNBOUNDED_INACCURATE;
    }
    if (status == OC_SOLVED || status == OC_SOLVED_INACCURATE) {
        for (int j = 0; j < n; j++) xo[j] = E[j] * u[j] / (sigma * tau);
        for (int i = 0; i < m; i++) { yo[i] = D[i] * u[n + i] / (sigma * tau); so[i] = rsk[n + i] / (D[i] * sigma * tau); }
    } else if (status == OC_INFEASIBLE || status == OC_INFEASIBLE_INACCURATE) {
        for (int j = 0; j < n; j++) xo[j] = NAN;
        for (int i = 0; i < m; i++) { yo[i] = D[i] * u[n + i] / sigma; so[i] = NAN; }
    } else {
        for (int j = 0; j < n; j++) xo[j] = E[j] * u[j] / sigma;
        for (int i = 0; i < m; i++) { yo[i] = NAN; so[i] = rsk[n + i] / (D[i] * sigma); }
    }
    free(aab);
    info->iters = iter; info->status = status; info->pobj = pobj; info->dobj = dobj; info->res_pri = res_pri;
    info->res_dual = res_dual; info->gap = gap; info->scale = scale; info->n_rescale = n_rescale;
    free(buf);
    return status;
}

/* ------------------------------------------------------------------ derivative of the dual-cone projection */
/* out = DPi_{K*}(v) h   (block diagonal, symmetric for every cone here).  psd_ws: per-PSD-cone eigen cache or NULL */
typedef struct { double *V, *w; } psd_cache;

static void dproj_soc(const double *v, int d, const double *h, double *out) {
    if (d == 0) return;
    if (d == 1) { out[0] = v[0] >= 0 ? h[0] : 0; return; }
    double t = v[0], nz = norm2(v + 1, d - 1);
    if (nz <= t) { memcpy(out, h, sizeof(double) * d); return; }
    if (nz <= -t) { memset(out, 0, sizeof(double) * d); return; }
    /* DPi = 1/(2 nz) [[nz, z^T],[z, (t+nz) I - t z z^T / nz^2]] */
    double zh = dot(v + 1, h + 1, d - 1);
    out[0] = (nz * h[0] + zh) / (2 * nz);
    for (int i = 1; i < d; i++) out[i] = (v[i] * h[0] + (t + nz) * h[i] - t * v[i] * zh / (nz * nz)) / (2 * nz);
}
static void dproj_psd(const psd_cache *pc, int k, const double *h, double *out) {
    double *H = malloc(sizeof(double) * k * k * 3), *T = H + k * k, *T2 = T + k * k;
    const double *V = pc->V, *w = pc->w;
    svec_to_mat(h, k, H);
    /* T = V^T H V */
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) { double a = 0; for (int r = 0; r < k; r++) a += H[i * k + r] * V[r * k + j]; T2[i * k + j] = a; }
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) { double a = 0; for (int r = 0; r < k; r++) a += V[r * k + i] * T2[r * k + j]; T[i * k + j] = a; }
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) {
        double wi = w[i], wj = w[j], bij;
        double pi = wi > 0 ? wi : 0, pj = wj > 0 ? wj : 0;
        if (wi > 0 && wj > 0) bij = 1; else if (wi <= 0 && wj <= 0) bij = 0; else bij = (pi - pj) / (wi - wj);
        T[i * k + j] *= bij;
    }
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) { double a = 0; for (int r = 0; r < k; r++) a += V[i * k + r] * T[r * k + j]; T2[i * k + j] = a; }
    for (int i = 0; i < k; i++) for (int j = 0; j < k; j++) { double a = 0; for (int r = 0; r < k; r++) a += T2[i * k + r] * V[j * k + r]; H[i * k + j] = a; }
    mat_to_svec(H, k, out); free(H);
}
static void dproj_dual_cone(const double *v, const oc_cones *K, const psd_cache *pcs, const double *h, double *out) {
    int off = 0;
    for (int i = 0; i < K->z; i++) out[i] = h[i];                      /* dual of zero cone is free: identity */
    off = K->z;
    for (int i = 0; i < K->l; i++) out[off + i] = v[off + i] > 0 ? h[off + i] : (v[off + i] < 0 ? 0 : 0.5 * h[off + i]);
    off += K->l;
    for (int c = 0; c < K->nq; c++) { dproj_soc(v + off, K->q[c], h + off, out + off); off += K->q[c]; }
    for (int c = 0; c < K->ns; c++) { dproj_psd(&pcs[c], K->s[c], h + off, out + off); off += K->s[c] * (K->s[c] + 1) / 2; }
    for (int c = 0; c < K->nep; c++) {
        double J[9]; dproj_exp_dual(v + off, J);
        for (int i = 0; i < 3; i++) out[off + i] = J[i * 3] * h[off] + J[i * 3 + 1] * h[off + 1] + J[i * 3 + 2] * h[off + 2];
        off += 3;
    }
    for (int c = 0; c < K->np; c++) {
        double J[9]; dproj_pow_entry(v + off, K->pw[c], 1, J);
        for (int i = 0; i < 3; i++) out[off + i] = J[i * 3] * h[off] + J[i * 3 + 1] * h[off + 1] + J[i * 3 + 2] * h[off + 2];
        off += 3;
    }
}

/* P (optional) and Px, xPx: quadratic objective; DQ(z) = [[P, A^T, c], [-A, 0, b], [-(c + 2 P x)^T, -b^T, x^T P x]] at tau = 1 */
typedef struct { int n, m; const double *A, *b, *c, *v; const oc_cones *K; const psd_cache *pcs; double *t1, *t2;
                 const double *P, *Px; double xPx; double *tn; } adj_op;

/* out = M^T r,  M = (Q - I) DPi(z) + I,  Q = [[0,A^T,c],[-A,0,b],[-c^T,-b^T,0]]
 *   M^T r = DPi^T ( -Q r - r ) + r                                                   */
static void apply_MT(const adj_op *op, const double *r, double *out) {
    int n = op->n, m = op->m; const double *rx = r, *ry = r + n; double rt = r[n + m];
    /* x block: -A^T ry - c rt */
    matvec_t(op->A, ry, out, m, n);
    for (int j = 0; j < n; j++) out[j] = -out[j] - op->c[j] * rt;
    if (op->P) { matvec_t(op->P, rx, op->tn, n, n); for (int j = 0; j < n; j++) out[j] += op->tn[j] - 2 * op->Px[j] * rt; }
    /* y block: D (A rx - b rt - ry) + ry */
    matvec(op->A, rx, op->t1, m, n);
    for (int i = 0; i < m; i++) op->t1[i] = op->t1[i] - op->b[i] * rt - ry[i];
    dproj_dual_cone(op->v, op->K, op->pcs, op->t1, op->t2);
    for (int i = 0; i < m; i++) out[n + i] = op->t2[i] + ry[i];
    /* tau block (DPi = 1): c^T rx + b^T ry (+ x^T P x r_tau) */
    out[n + m] = dot(op->c, rx, n) + dot(op->b, ry, m) + (op->P ? op->xPx * rt : 0.0);
}
/* out = M p */
static void apply_M(const adj_op *op, const double *p, double *out) {
    int n = op->n, m = op->m;
    /* q = DPi p */
    double *qy = op->t1; dproj_dual_cone(op->v, op->K, op->pcs, p + n, qy);
    const double *qx = p; double qt = p[n + m];
    /* (Q - I) q + p */
    matvec_t(op->A, qy, out, m, n);
    for (int j = 0; j < n; j++) out[j] = out[j] + op->c[j] * qt - qx[j] + p[j];
    if (op->P) { matvec(op->P, qx, op->tn, n, n); for (int j = 0; j < n; j++) out[j] += op->tn[j]; }
    matvec(op->A, qx, op->t2, m, n);
    for (int i = 0; i < m; i++) out[n + i] = -op->t2[i] + op->b[i] * qt - qy[i] + p[n + i];
    out[n + m] = -dot(op->c, qx, n) - dot(op->b, qy, m) - qt + p[n + m];
    if (op->P) out[n + m] += -2 * dot(op->Px, qx, n) + op->xPx * qt;
}

/* LSQR (Paige & Saunders 1982), solving min || MT r - dz ||, zero start -> minimum-norm solution */
static int lsqr_MT(const adj_op *op, const double *bvec, int N, double *x, const oc_opts *o, double *work) {
    double *u = work, *v = u + N, *wv = v + N, *tmp = wv + N;
    int iter_lim = o->lsqr_iter_lim > 0 ? o->lsqr_iter_lim : 2 * N;
    double atol = o->lsqr_atol, btol = o->lsqr_btol, ctol = o->lsqr_conlim > 0 ? 1.0 / o->lsqr_conlim : 0;
    memset(x, 0, sizeof(double) * N);
    memcpy(u, bvec, sizeof(double) * N);
    double bnorm = norm2(bvec, N), beta = bnorm, alfa = 0;
    if (beta > 0) { for (int i = 0; i < N; i++) u[i] /= beta; apply_M(op, u, v); alfa = norm2(v, N); } else { memset(v, 0, sizeof(double) * N); }
    if (alfa > 0) for (int i = 0; i < N; i++) v[i] /= alfa;
    memcpy(wv, v, sizeof(double) * N);
    double rhobar = alfa, phibar = beta, anorm = 0, acond = 0, ddnorm = 0, xnorm = 0, xxnorm = 0, z = 0, cs2 = -1, sn2 = 0, res2 = 0;
    double arnorm = alfa * beta;
    if (arnorm == 0) return 0;
    int itn = 0;
    while (itn < iter_lim) {
        itn++;
        /* u = MT v - alfa u */
        apply_MT(op, v, tmp);
        for (int i = 0; i < N; i++) u[i] = tmp[i] - alfa * u[i];
        beta = norm2(u, N);
        if (beta > 0) {
            for (int i = 0; i < N; i++) u[i] /= beta;
            anorm = sqrt(anorm * anorm + alfa * alfa + beta * beta);
            apply_M(op, u, tmp);
            for (int i = 0; i < N; i++) v[i] = tmp[i] - beta * v[i];
            alfa = norm2(v, N);
            if (alfa > 0) for (int i = 0; i < N; i++) v[i] /= alfa;
        }
        double rho = sqrt(rhobar * rhobar + beta * beta), cs = rhobar / rho, sn = beta / rho;
        double theta = sn * alfa; rhobar = -cs * alfa; double phi = cs * phibar; phibar = sn * phibar; double tau = sn * phi;
        double t1 = phi / rho, t2 = -theta / rho;
        double dd = 0;
        for (int i = 0; i < N; i++) { double dk = wv[i] / rho; dd += dk * dk; x[i] += t1 * wv[i]; wv[i] = v[i] + t2 * wv[i]; }
        ddnorm += dd;
        double delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * z, zbar = rhs / gambar;
        xnorm = sqrt(xxnorm + zbar * zbar);
        double gamma = sqrt(gambar * gambar + theta * theta); cs2 = gambar / gamma; sn2 = theta / gamma; z = rhs / gamma; xxnorm += z * z;
        acond = anorm * sqrt(ddnorm);
        double res1 = phibar * phibar; double rnorm = sqrt(res1 + res2);
        arnorm = alfa * fabs(tau);
        double test1 = rnorm / bnorm, test2 = arnorm / (anorm * rnorm + 1e-300), test3 = 1 / (acond + 1e-300);
        double tt1 = test1 / (1 + anorm * xnorm / bnorm), rtol = btol + atol * anorm * xnorm / bnorm;
        if (1 + test3 <= 1 || 1 + test2 <= 1 || 1 + tt1 <= 1) break;
        if (test3 <= ctol || test2 <= atol || test1 <= rtol) break;
    }
    return itn;
}

/* LSMR (Fong & Saunders 2011, "LSMR: an iterative algorithm for sparse least-squares problems"; the recurrences and stopping tests as scipy.sparse.linalg.lsmr states
 * them, damp = 0, zero start): diffcp's mode="lsmr".  Generic over the operator so that tests can pin the recurrence on an explicit matrix against scipy
 * (oc_lsmr_dense below); solve_and_derivative's adjoint runs it on MT like lsqr_MT.  work: 2 * mrows + 3 * ncols doubles, mrows >= ncols.  Returns the iteration count. */
typedef void (*lsmr_apply)(const void *ctx, const double *in, double *out);
static void sym_ortho(double a, double b, double *c, double *s, double *r) {
    if (b == 0) { *c = a == 0 ? 1.0 : (a > 0 ? 1.0 : -1.0); *s = 0; *r = fabs(a); }
    else if (a == 0) { *c = 0; *s = b > 0 ? 1.0 : -1.0; *r = fabs(b); }
    else if (fabs(b) > fabs(a)) { double tau = a / b; *s = (b > 0 ? 1.0 : -1.0) / sqrt(1 + tau * tau); *c = *s * tau; *r = b / *s; }
    else { double tau = b / a; *c = (a > 0 ? 1.0 : -1.0) / sqrt(1 + tau * tau); *s = *c * tau; *r = a / *c; }
}
static int lsmr_core(lsmr_apply A_mul, lsmr_apply AT_mul, const void *ctx, int mrows, int ncols, const double *bvec, double *x,
                     double atol, double btol, double conlim, int maxiter, double *work) {
    double *u = work, *tmpm = u + mrows, *v = tmpm + mrows, *h = v + ncols, *hbar = h + ncols;      /* (tmpm is the scratch of both products: mrows >= ncols) */
    memset(x, 0, sizeof(double) * ncols);
    memcpy(u, bvec, sizeof(double) * mrows);
    double normb = norm2(bvec, mrows), beta = normb, alpha = 0;
    if (beta > 0) { for (int i = 0; i < mrows; i++) u[i] /= beta; AT_mul(ctx, u, v); alpha = norm2(v, ncols); } else memset(v, 0, sizeof(double) * ncols);
    if (alpha > 0) for (int j = 0; j < ncols; j++) v[j] /= alpha;
    double zetabar = alpha * beta, alphabar = alpha, rho = 1, rhobar = 1, cbar = 1, sbar = 0;
    memcpy(h, v, sizeof(double) * ncols); memset(hbar, 0, sizeof(double) * ncols);
    double betadd = beta, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, d = 0;
    double normA2 = alpha * alpha, maxrbar = 0, minrbar = 1e100;
    const double ctol = conlim > 0 ? 1.0 / conlim : 0;
    if (alpha * beta == 0) return 0;
    int itn = 0;
    while (itn < maxiter) {
        itn++;
        A_mul(ctx, v, tmpm);
        for (int i = 0; i < mrows; i++) u[i] = tmpm[i] - alpha * u[i];
        beta = norm2(u, mrows);
        if (beta > 0) {
            for (int i = 0; i < mrows; i++) u[i] /= beta;
            AT_mul(ctx, u, tmpm);
            for (int j = 0; j < ncols; j++) v[j] = tmpm[j] - beta * v[j];
            alpha = norm2(v, ncols);
            if (alpha > 0) for (int j = 0; j < ncols; j++) v[j] /= alpha;
        }
        /* damp = 0: the first rotation is trivial */
        double chat, shat, alphahat; sym_ortho(alphabar, 0.0, &chat, &shat, &alphahat);
        const double rhoold = rho; double c, s_; sym_ortho(alphahat, beta, &c, &s_, &rho);
        const double thetanew = s_ * alpha; alphabar = c * alpha;
        const double rhobarold = rhobar, zetaold = zeta, thetabar = sbar * rho, rhotemp = cbar * rho;
        sym_ortho(cbar * rho, thetanew, &cbar, &sbar, &rhobar);
        zeta = cbar * zetabar; zetabar = -sbar * zetabar;
        const double f1 = -(thetabar * rho / (rhoold * rhobarold)), f2 = zeta / (rho * rhobar), f3 = -(thetanew / rho);
        double xx = 0;
        for (int j = 0; j < ncols; j++) { const double hb = hbar[j] * f1 + h[j]; hbar[j] = hb; const double xv = x[j] + f2 * hb; x[j] = xv; xx += xv * xv; h[j] = h[j] * f3 + v[j]; }
        const double betaacute = chat * betadd, betacheck = -shat * betadd;
        const double betahat = c * betaacute; betadd = -s_ * betaacute;
        const double thetatildeold = thetatilde; double ctildeold, stildeold, rhotildeold; sym_ortho(rhodold, thetabar, &ctildeold, &stildeold, &rhotildeold);
        thetatilde = stildeold * rhobar; rhodold = ctildeold * rhobar; betad = -stildeold * betad + ctildeold * betahat;
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
        const double taud = (zeta - thetatilde * tautildeold) / rhodold;
        d += betacheck * betacheck;
        const double normr = sqrt(d + (betad - taud) * (betad - taud) + betadd * betadd);
        normA2 += beta * beta; const double normA = sqrt(normA2); normA2 += alpha * alpha;
        if (rhobarold > maxrbar) maxrbar = rhobarold;
        if (itn > 1 && rhobarold < minrbar) minrbar = rhobarold;
        const double condA = (maxrbar > rhotemp ? maxrbar : rhotemp) / (minrbar < rhotemp ? minrbar : rhotemp);
        const double normar = fabs(zetabar), normx = sqrt(xx);
        const double test1 = normr / normb, test2 = (normA * normr) != 0 ? normar / (normA * normr) : INFINITY, test3 = 1 / condA;
        const double t1 = test1 / (1 + normA * normx / normb), rtol = btol + atol * normA * normx / normb;
        if (1 + test3 <= 1 || 1 + test2 <= 1 || 1 + t1 <= 1) break;
        if (test3 <= ctol || test2 <= atol || test1 <= rtol) break;
    }
    return itn;
}
static void lsmr_op_MT(const void *ctx, const double *in, double *out) { apply_MT((const adj_op *)ctx, in, out); }
static void lsmr_op_M(const void *ctx, const double *in, double *out) { apply_M((const adj_op *)ctx, in, out); }
static int lsmr_MT(const adj_op *op, const double *bvec, int N, double *x, const oc_opts *o, double *work) {
    return lsmr_core(lsmr_op_MT, lsmr_op_M, op, N, N, bvec, x, o->lsqr_atol, o->lsqr_btol, o->lsqr_conlim, o->lsqr_iter_lim > 0 ? o->lsqr_iter_lim : 2 * N, work);
}
/* the same recurrence on an explicit row-major matrix A (mrows x ncols): test entry point (tests/test_oracle_known_answers.py pins it on scipy.sparse.linalg.lsmr) */
typedef struct { int mrows, ncols; const double *A; } lsmr_dense_ctx;
static void lsmr_dense_A(const void *ctx, const double *in, double *out) { const lsmr_dense_ctx *c = ctx; for (int i = 0; i < c->mrows; i++) out[i] = dot(c->A + (size_t)i * c->ncols, in, c->ncols); }
static void lsmr_dense_AT(const void *ctx, const double *in, double *out) {
    const lsmr_dense_ctx *c = ctx; memset(out, 0, sizeof(double) * c->ncols);
    for (int i = 0; i < c->mrows; i++) for (int j = 0; j < c->ncols; j++) out[j] += c->A[(size_t)i * c->ncols + j] * in[i];
}
int oc_lsmr_dense(int mrows, int ncols, const double *A, const double *b, double atol, double btol, double conlim, int maxiter, double *x) {
    if (mrows < ncols) return -1;          /* (lsmr_core's scratch vector serves both sides: mrows >= ncols; the adjoint's operator is square) */
    lsmr_dense_ctx c = { mrows, ncols, A };
    double *work = calloc((size_t)2 * mrows + 3 * (size_t)ncols, sizeof(double));
    const int it = lsmr_core(lsmr_dense_A, lsmr_dense_AT, &c, mrows, ncols, b, x, atol, btol, conlim, maxiter, work);
    free(work);
    return it;
}

/* dense: build MT column by column, Gaussian elimination with complete pivoting, rank-revealing (the
 * embedding makes M singular along z; the system is consistent; free variables set to zero). */
static void dense_solve_MT(const adj_op *op, const double *bvec, int N, double *x) {
    double *Mt = malloc(sizeof(double) * ((size_t)N * N + 3 * N)); double *e = Mt + (size_t)N * N, *col = e + N, *rhs = col + N;
    int *cp = malloc(sizeof(int) * N);
    for (int k = 0; k < N; k++) { memset(e, 0, sizeof(double) * N); e[k] = 1; apply_MT(op, e, col); for (int i = 0; i < N; i++) Mt[(size_t)i * N + k] = col[i]; }
    memcpy(rhs, bvec, sizeof(double) * N);
    for (int i = 0; i < N; i++) cp[i] = i;
    double amax0 = 0; for (size_t i = 0; i < (size_t)N * N; i++) if (fabs(Mt[i]) > amax0) amax0 = fabs(Mt[i]);
    int rank = N;
    for (int k = 0; k < N; k++) {
        int pi = k, pj = k; double best = 0;
        for (int i = k; i < N; i++) for (int j = k; j < N; j++) { double v = fabs(Mt[(size_t)i * N + j]); if (v > best) { best = v; pi = i; pj = j; } }
        if (best <= 1e-11 * amax0) { rank = k; break; }
        if (pi != k) { for (int j = 0; j < N; j++) { double t = Mt[(size_t)k * N + j]; Mt[(size_t)k * N + j] = Mt[(size_t)pi * N + j]; Mt[(size_t)pi * N + j] = t; } double t = rhs[k]; rhs[k] = rhs[pi]; rhs[pi] = t; }
        if (pj != k) { for (int i = 0; i < N; i++) { double t = Mt[(size_t)i * N + k]; Mt[(size_t)i * N + k] = Mt[(size_t)i * N + pj]; Mt[(size_t)i * N + pj] = t; } int t = cp[k]; cp[k] = cp[pj]; cp[pj] = t; }
        double piv = Mt[(size_t)k * N + k];
        for (int i = k + 1; i < N; i++) { double f = Mt[(size_t)i * N + k] / piv; if (f == 0) continue; for (int j = k; j < N; j++) Mt[(size_t)i * N + j] -= f * Mt[(size_t)k * N + j]; rhs[i] -= f * rhs[k]; }
    }
    double *xp = col; memset(xp, 0, sizeof(double) * N);
    for (int k = rank - 1; k >= 0; k--) { double v = rhs[k]; for (int j = k + 1; j < rank; j++) v -= Mt[(size_t)k * N + j] * xp[j]; xp[k] = v / Mt[(size_t)k * N + k]; }
    for (int k = 0; k < N; k++) x[cp[k]] = xp[k];
    free(Mt); free(cp);
}

static int adjoint_one(int n, int m, const double *A, const double *b, const double *c, const double *Pq, const oc_cones *K, const oc_opts *o,
                       const double *x, const double *y, const double *s, const double *dx, const double *dy, const double *ds,
                       double *dA, double *db, double *dc, double *dP) {
    int N = n + m + 1;
    double *buf = calloc((size_t)12 * N + 4 * m + 2 * n, sizeof(double));
    double *v = buf, *dz = v + m, *r = dz + N, *t1 = r + N, *t2 = t1 + m, *t3 = t2 + m, *work = t3 + m;
    for (int i = 0; i < m; i++) v[i] = y[i] - s[i];
    psd_cache *pcs = NULL;
    if (K->ns > 0) {
        pcs = malloc(sizeof(psd_cache) * K->ns);
        int off = K->z + K->l; for (int q = 0; q < K->nq; q++) off += K->q[q];
        for (int cidx = 0; cidx < K->ns; cidx++) { int k = K->s[cidx]; double *S = malloc(sizeof(double) * k * k);
            pcs[cidx].V = malloc(sizeof(double) * k * k); pcs[cidx].w = malloc(sizeof(double) * k);
            svec_to_mat(v + off, k, S); jacobi_eig(S, k, pcs[cidx].w, pcs[cidx].V); free(S); off += k * (k + 1) / 2; }
    }
    double *Pxv = buf + (size_t)12 * N + 4 * m, *tnv = Pxv + n, xPx = 0;
    if (Pq) { matvec(Pq, x, Pxv, n, n); xPx = dot(x, Pxv, n); }
    adj_op op = { n, m, A, b, c, v, K, pcs, t1, t2, Pq, Pxv, xPx, tnv };
    /* dz = (dx, DPi^T (dy + ds) - ds, -(x.dx + y.dy + s.ds)) */
    memcpy(dz, dx, sizeof(double) * n);
    for (int i = 0; i < m; i++) t3[i] = dy[i] + (ds ? ds[i] : 0);
    dproj_dual_cone(v, K, pcs, t3, dz + n);
    double dw = -(dot(x, dx, n) + dot(y, dy, m));
    if (ds) { for (int i = 0; i < m; i++) dz[n + i] -= ds[i]; dw -= dot(s, ds, m); }
    dz[N - 1] = dw;
    int itn = 0;
    if (norm_inf(dz, N) == 0) memset(r, 0, sizeof(double) * N);
    else if (o->adj_mode == 1) dense_solve_MT(&op, dz, N, r);
    else if (o->adj_mode == 2) itn = lsmr_MT(&op, dz, N, r, o, work);
    else itn = lsqr_MT(&op, dz, N, r, o, work);
    /* dQ = r Pi(z)^T antisymmetrised, Pi(z) = (x, y, 1) */
    const double *rx = r, *ry = r + n; double rt = r[N - 1];
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) dA[(size_t)i * n + j] = x[j] * ry[i] - y[i] * rx[j];
    for (int i = 0; i < m; i++) db[i] = y[i] * rt - ry[i];
    for (int j = 0; j < n; j++) dc[j] = x[j] * rt - rx[j];
    if (Pq && dP)      /* Q_xx = P, Q_tau,x = -(c + (1/tau) P x)^T  =>  dP = sym(-r_x x^T) + r_tau x x^T */
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) dP[(size_t)i * n + j] = -0.5 * (rx[i] * x[j] + rx[j] * x[i]) + rt * x[i] * x[j];
    if (pcs) { for (int cidx = 0; cidx < K->ns; cidx++) { free(pcs[cidx].V); free(pcs[cidx].w); } free(pcs); }
    free(buf);
    return itn;
}

/* ------------------------------------------------------------------ batch entry points (ctypes) */
int oc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* The per-instance routines allocate their workspaces with calloc / malloc.  Blocks above glibc's mmap threshold (128 KB: the dense (n+m+1)^2 system of the
 * adjoint, the Anderson history) would be mmap'ed and unmapped once per instance, and with many OpenMP threads those calls serialise on the process's
 * address-space lock -- the timed CPU baseline then measures the kernel's mm lock, not the solver.  Keep them in the per-thread arenas instead. */
static void tune_malloc(void) {
    static int done = 0;
    if (!done) { mallopt(M_MMAP_THRESHOLD, 256 << 20); mallopt(M_TRIM_THRESHOLD, 512 << 20); mallopt(M_TOP_PAD, 16 << 20); done = 1; }
}

/* A: [B][m][n] row-major dense, b: [B][m], c: [B][n]; outputs x [B][n], y,s [B][m], iters/status [B], resid [B][3] */
int oc_solve_batch_qp(int B, int n, int m, const double *A, const double *b, const double *c, const double *Pq,
                      int z, int l, int nq, const int *q, int ns, const int *s, int nep, int np, const double *pw, const oc_opts *o,
                      double *x, double *y, double *sv, int *iters, int *status, double *resid, int nthreads) {
    oc_cones K = { z, l, nq, ns, q, s, nep, np, pw };
    if (cone_rows(&K) != m) return -1;
    tune_malloc();
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    #pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < B; i++) {
        oc_info info; memset(&info, 0, sizeof(info));
        solve_one(n, m, A + (size_t)i * m * n, b + (size_t)i * m, c + (size_t)i * n, Pq ? Pq + (size_t)i * n * n : NULL, &K, o, x + (size_t)i * n, y + (size_t)i * m, sv + (size_t)i * m, &info);
        iters[i] = info.iters; status[i] = info.status;
        if (resid) { resid[3 * i] = info.res_pri; resid[3 * i + 1] = info.res_dual; resid[3 * i + 2] = info.gap; }
    }
    return 0;
}

int oc_solve_batch(int B, int n, int m, const double *A, const double *b, const double *c,
                   int z, int l, int nq, const int *q, int ns, const int *s, int nep, int np, const double *pw, const oc_opts *o,
                   double *x, double *y, double *sv, int *iters, int *status, double *resid, int nthreads) {
    return oc_solve_batch_qp(B, n, m, A, b, c, NULL, z, l, nq, q, ns, s, nep, np, pw, o, x, y, sv, iters, status, resid, nthreads);
}

/* ds may be NULL (the layer passes ds = 0, diffcp_if.py:84).  dA: [B][m][n] dense. lsqr_iters may be NULL.  Pq / dP: [B][n][n] or NULL. */
int oc_adjoint_batch_qp(int B, int n, int m, const double *A, const double *b, const double *c, const double *Pq,
                        int z, int l, int nq, const int *q, int ns, const int *s, int nep, int np, const double *pw, const oc_opts *o,
                        const double *x, const double *y, const double *sv, const double *dx, const double *dy, const double *ds,
                        double *dA, double *db, double *dc, double *dP, int *lsqr_iters, int nthreads) {
    oc_cones K = { z, l, nq, ns, q, s, nep, np, pw };
    if (cone_rows(&K) != m) return -1;
    tune_malloc();
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    #pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < B; i++) {
        int it = adjoint_one(n, m, A + (size_t)i * m * n, b + (size_t)i * m, c + (size_t)i * n, Pq ? Pq + (size_t)i * n * n : NULL, &K, o,
                             x + (size_t)i * n, y + (size_t)i * m, sv + (size_t)i * m, dx + (size_t)i * n, dy + (size_t)i * m,
                             ds ? ds + (size_t)i * m : NULL, dA + (size_t)i * m * n, db + (size_t)i * m, dc + (size_t)i * n,
                             dP ? dP + (size_t)i * n * n : NULL);
        if (lsqr_iters) lsqr_iters[i] = it;
    }
    return 0;
}

int oc_adjoint_batch(int B, int n, int m, const double *A, const double *b, const double *c,
                     int z, int l, int nq, const int *q, int ns, const int *s, int nep, int np, const double *pw, const oc_opts *o,
                     const double *x, const double *y, const double *sv, const double *dx, const double *dy, const double *ds,
                     double *dA, double *db, double *dc, int *lsqr_iters, int nthreads) {
    return oc_adjoint_batch_qp(B, n, m, A, b, c, NULL, z, l, nq, q, ns, s, nep, np, pw, o, x, y, sv, dx, dy, ds, dA, db, dc, NULL, lsqr_iters, nthreads);
}
