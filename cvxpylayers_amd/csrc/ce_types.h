// ce_types.h -- types shared by the translation units of libcone_engine.so and the launcher entry points each kernel
// translation unit exports to the host dispatch (cone_engine.hip).  The kernels are split over several .hip files so that
// they compile in parallel and a change to one kernel family rebuilds one object (csrc/Makefile).
#pragma once
#include <hip/hip_runtime.h>

#include "cone_engine.h"

struct DevT {
    int n, m, nnz_aug, nnzA, z, l, nq, lda, ldg, maxq;
    const int *rowidx;    // [nnz_aug] row of structural entry k
    const int *colidx;    // [nnz_aug] column (n == the b column)
    const int *rowcone;   // [m] -1 for zero / nonneg rows, else SOC index
    const int *qoff;      // [nq+1] first row of SOC c
    int ns, maxs;         // PSD cones, largest order
    const int *soff;      // [ns+1] first row of PSD cone c (svec blocks follow the SOCs, SCS row order z,l,q,s)
    const int *sord;      // [ns] order k of PSD cone c
    int nep, eoff;        // exponential cones (3 rows each) and their first row (after the PSD blocks: SCS row order z,l,q,s,ep,p)
    int np;               // 3-d power cones, after the exponential cones
    const double *pw;     // [np] exponent a of x^a y^(1-a) >= |z|; a < 0: the dual cone of exponent |a| (SCS convention)
    int f2_neumann;       // k_fwd2: a rescale updates G by a Neumann series instead of refactoring (CE_F2_NEUMANN=0 disables)
    int gen_blocked_f, gen_blocked_b;   // size-generic kernels with G / K in global memory: LDS holds the panels of the blocked eliminations (else: unblocked)
};

// arguments of one forward launch (all kernels of the forward family take a subset)
struct CeFwdArgs {
    DevT T; ce_settings S;
    const double *Abm; const double *q; long sqk, sqb;
    const int *idx_at, *idx_ar, *idx_b;
    double *x, *y, *s; int *iters, *status; double *resid;
    const double *P; int nnz_p; const int *idx_p;
    const int *row_perm;        // k_fwd2 WL variants: kernel row -> template row (NULL: rows in template order)
    double *gA, *gG;            // global residency workspaces of the size-generic kernel
    int *iters2;                // k_fwd2: second copy of the iteration counts (engine-owned; NULL: not wanted)
    const int *order;           // k_fwd2: workgroup -> instance (NULL: identity); longest-first dispatch from the previous call's iteration counts
    double *aa_ws;              // size-generic kernel: Anderson-acceleration history, [B][4][lp] doubles of global memory (NULL: plain iteration)
};
struct CeBwdArgs {
    DevT T; int nkcap, ldk;
    const double *Abm, *x, *y, *s, *dx, *dy;
    double *dA, *dq; long sdqk, sdqb; int *adj;
    const double *P; int nnz_p; const int *pmap, *prow, *pcol; int p_tri; double *dP;
    double *gA, *gK;
    int retry;                  // k_backward_rt: 1 = recompute only the instances an earlier launch flagged (adj == 2)
    int *nk_max;                // k_backward_rt: device maximum of the systems' order NK over the batch (NULL: not wanted)
    int *fix;                   // fix[0]: counter, fix[1 ...]: instances whose adjoint system the elimination found rank deficient (or too large for the tile), appended
                                // by the kernels for the LSQR re-solve behind them (cone_engine.hip ce_vjp_qp); NULL: not wanted
    int nonfinal;               // k_backward_rt: 1 = first launch of a two-tile plan (an instance this tile does not hold is not listed: the retry launch serves it)
};

// launchers (one per kernel object file): 0 on success, -1 unknown variant
int ce_launch_fwd2_plain(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a);   // zero / nonneg / SOC
int ce_launch_fwd2_psd(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a);     // + PSD / exponential / power cones
int ce_launch_fwd2_qp(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a);      // quadratic objective inside the kernel
int ce_launch_fwd_rt(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a);
int ce_launch_fwd_generic(int mode, int B, size_t lds, hipStream_t st, const CeFwdArgs &a);
int ce_launch_bwd_rt_plain(int variant, int B, size_t lds, hipStream_t st, const CeBwdArgs &a);
int ce_launch_bwd_ns(int variant, int B, size_t lds, hipStream_t st, const CeBwdArgs &a);       // search-free null-space adjoint (plain cones), variants {2|256, 4|256, 7|512}
size_t ce_bwd_ns_lds_bytes(int n, int m, int nq, int variant);
hipError_t ce_setattr_bwd_ns(int bytes);
int ce_launch_bwd_rt_psd(int variant, int B, size_t lds, hipStream_t st, const CeBwdArgs &a);
int ce_launch_bwd_generic(int mode, int B, size_t lds, hipStream_t st, const CeBwdArgs &a);
// raise the dynamic-LDS limit of every kernel of the family
hipError_t ce_setattr_fwd2_plain(int bytes);
hipError_t ce_setattr_fwd2_psd(int bytes);
hipError_t ce_setattr_fwd2_qp(int bytes);
hipError_t ce_setattr_fwd_rt(int bytes);
hipError_t ce_setattr_fwd_generic(int bytes);
hipError_t ce_setattr_bwd_rt_plain(int bytes);
hipError_t ce_setattr_bwd_rt_psd(int bytes);
hipError_t ce_setattr_bwd_generic(int bytes);
