// translation unit: register-tiled structured adjoint (k_backward_rt)
//   -DCE_BRT_PSD=0 plain cones, 1 PSD / exponential / power cones (one object file each: csrc/Makefile)
#include "ce_tu_prologue.h"
namespace {
#include "ce_common.h"
#include "ce_expcone.h"
#include "ce_forward_rt.h"
#include "ce_forward_v2.h"
#include "ce_global_mv.h"
#include "ce_backward.h"
#include "ce_backward_rt.h"
#if CE_BRT_PSD == 0
#include "ce_backward_ns.h"
#endif
}  // namespace

#ifndef CE_BRT_PSD
#error "compile with -DCE_BRT_PSD=0|1"
#endif
#define BRT_ARGS a.T, a.Abm, a.x, a.y, a.s, a.dx, a.dy, a.dA, a.dq, a.sdqk, a.sdqb, a.adj, a.P, a.nnz_p, a.pmap, a.prow, a.pcol, a.p_tri, a.dP, a.retry, a.nk_max, a.fix, a.nonfinal
#define LAUNCH_BRT(NTHREADS, ...) hipLaunchKernelGGL((k_backward_rt<__VA_ARGS__>), dim3(B), dim3(NTHREADS), lds, st, BRT_ARGS)
#define SETATTR(...) do { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_backward_rt<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e_ != hipSuccess) return e_; } while (0)

#if CE_BRT_PSD == 0
int ce_launch_bwd_rt_plain(int variant, int B, size_t lds, hipStream_t st, const CeBwdArgs &a) {
    switch (variant) {
    case 0: LAUNCH_BRT(256, 4, 4, 4); break;
    case 1: LAUNCH_BRT(256, 5, 5, 4); break;
    case 2: LAUNCH_BRT(256, 6, 6, 4); break;
    case 3: LAUNCH_BRT(256, 7, 7, 4); break;
    case 4: LAUNCH_BRT(256, 7, 7, 7); break;
    case 5: LAUNCH_BRT(512, 5, 9, 7, false, 32); break;
    case 6: LAUNCH_BRT(512, 7, 13, 7, false, 32); break;
    default: return -1;
    }
    return 0;
}
// search-free null-space adjoint (ce_backward_ns.h): variant -> {tiles of 16 reduced columns, threads}
#define NS_ARGS a.T, a.Abm, a.x, a.y, a.s, a.dx, a.dy, a.dA, a.dq, a.sdqk, a.sdqb, a.adj, a.fix
int ce_launch_bwd_ns(int variant, int B, size_t lds, hipStream_t st, const CeBwdArgs &a) {
    switch (variant) {
    case 0: hipLaunchKernelGGL((k_backward_ns<2, 256>), dim3(B), dim3(256), lds, st, NS_ARGS); break;
    case 1: hipLaunchKernelGGL((k_backward_ns<4, 256>), dim3(B), dim3(256), lds, st, NS_ARGS); break;
    case 2: hipLaunchKernelGGL((k_backward_ns<7, 512>), dim3(B), dim3(512), lds, st, NS_ARGS); break;
    default: return -1;
    }
    return 0;
}
hipError_t ce_setattr_bwd_ns(int bytes) {
#define SETATTR_NS(...) do { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_backward_ns<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e_ != hipSuccess) return e_; } while (0)
    SETATTR_NS(2, 256); SETATTR_NS(4, 256); SETATTR_NS(7, 512);
#undef SETATTR_NS
    return hipSuccess;
}
size_t ce_bwd_ns_lds_bytes(int n, int m, int nq, int variant) {
    static const int V[3][2] = {{2, 256}, {4, 256}, {7, 512}};
    return bwd_ns_lds_bytes_of(n, m, nq, V[variant][0], V[variant][1]);
}
hipError_t ce_setattr_bwd_rt_plain(int bytes) {
    SETATTR(4, 4, 4); SETATTR(5, 5, 4); SETATTR(6, 6, 4); SETATTR(7, 7, 4); SETATTR(7, 7, 7); SETATTR(5, 9, 7, false, 32); SETATTR(7, 13, 7, false, 32);
    return hipSuccess;
}
#else
int ce_launch_bwd_rt_psd(int variant, int B, size_t lds, hipStream_t st, const CeBwdArgs &a) {
    switch (variant) {
    case 0: LAUNCH_BRT(256, 4, 4, 4, true); break;
    case 3: LAUNCH_BRT(256, 7, 7, 4, true); break;
    case 4: LAUNCH_BRT(256, 7, 7, 7, true); break;
    case 6: LAUNCH_BRT(512, 7, 13, 7, true, 32); break;
    default: return -1;
    }
    return 0;
}
hipError_t ce_setattr_bwd_rt_psd(int bytes) {
    SETATTR(4, 4, 4, true); SETATTR(7, 7, 4, true); SETATTR(7, 7, 7, true); SETATTR(7, 13, 7, true, 32);
    return hipSuccess;
}
#endif
