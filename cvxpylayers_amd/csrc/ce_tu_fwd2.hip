// translation unit: second-generation forward kernel (k_fwd2), all instantiations of one `kind`
//   -DCE_F2_KIND=0 plain cones, 1 PSD / exponential / power cones, 2 quadratic objective   (one object file per kind: csrc/Makefile)
#include "ce_tu_prologue.h"
namespace {
#include "ce_common.h"
#include "ce_expcone.h"
#include "ce_forward_rt.h"
#include "ce_forward_v2.h"
}  // namespace

#ifndef CE_F2_KIND
#error "compile with -DCE_F2_KIND=0|1|2"
#endif

#define F2_ARGS a.T, a.S, a.Abm, a.q, a.sqk, a.sqb, a.idx_at, a.idx_ar, a.idx_b, a.x, a.y, a.s, a.iters, a.status, a.resid, a.P, a.nnz_p, a.idx_p, a.row_perm, a.order, a.iters2
#define LAUNCH_F2(NTHREADS, ...) hipLaunchKernelGGL((k_fwd2<__VA_ARGS__>), dim3(B), dim3(NTHREADS), lds, st, F2_ARGS)
#define SETATTR(...) do { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd2<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e_ != hipSuccess) return e_; } while (0)

#if CE_F2_KIND == 0
int ce_launch_fwd2_plain(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a) {
    if (a.row_perm) {      // rows packed so that every cone is wave-local (WL instantiations)
        switch (variant) {
        case 0: LAUNCH_F2(256, 16, 2, 8, 2, 16, 2, false, 256, false, true); break;
        case 1: LAUNCH_F2(256, 8, 8, 4, 8, 8, 4, false, 256, false, true); break;
        case 2: LAUNCH_F2(256, 4, 26, 2, 26, 4, 14, false, 256, false, true); break;
        case 3: LAUNCH_F2(512, 8, 20, 2, 32, 8, 8, false, 512, false, true); break;
        case 4: LAUNCH_F2(512, 4, 30, 4, 26, 4, 26, false, 512, false, true); break;
        default: return -1;
        }
        return 0;
    }
    switch (variant) {
    case 0: LAUNCH_F2(256, 16, 2, 8, 2, 16, 2); break;
    case 1: LAUNCH_F2(256, 8, 8, 4, 8, 8, 4); break;
    case 2: LAUNCH_F2(256, 4, 26, 2, 26, 4, 14); break;
    case 3: LAUNCH_F2(512, 8, 20, 2, 32, 8, 8, false, 512); break;
    case 4: LAUNCH_F2(512, 4, 30, 4, 26, 4, 26, false, 512); break;
    default: return -1;
    }
    return 0;
}
hipError_t ce_setattr_fwd2_plain(int bytes) {
    SETATTR(16, 2, 8, 2, 16, 2); SETATTR(8, 8, 4, 8, 8, 4); SETATTR(4, 26, 2, 26, 4, 14);
    SETATTR(8, 20, 2, 32, 8, 8, false, 512); SETATTR(4, 30, 4, 26, 4, 26, false, 512);
    SETATTR(16, 2, 8, 2, 16, 2, false, 256, false, true); SETATTR(8, 8, 4, 8, 8, 4, false, 256, false, true); SETATTR(4, 26, 2, 26, 4, 14, false, 256, false, true);
    SETATTR(8, 20, 2, 32, 8, 8, false, 512, false, true); SETATTR(4, 30, 4, 26, 4, 26, false, 512, false, true);
    return hipSuccess;
}
#elif CE_F2_KIND == 1
int ce_launch_fwd2_psd(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a) {
    switch (variant) {
    case 0: LAUNCH_F2(256, 16, 2, 8, 2, 16, 2, true); break;
    case 1: LAUNCH_F2(256, 8, 8, 4, 8, 8, 4, true); break;
    case 2: LAUNCH_F2(256, 4, 26, 2, 26, 4, 14, true); break;
    case 3: LAUNCH_F2(512, 8, 20, 2, 32, 8, 8, true, 512); break;
    case 4: LAUNCH_F2(512, 4, 30, 4, 26, 4, 26, true, 512); break;
    default: return -1;
    }
    return 0;
}
hipError_t ce_setattr_fwd2_psd(int bytes) {
    SETATTR(16, 2, 8, 2, 16, 2, true); SETATTR(8, 8, 4, 8, 8, 4, true); SETATTR(4, 26, 2, 26, 4, 14, true);
    SETATTR(8, 20, 2, 32, 8, 8, true, 512); SETATTR(4, 30, 4, 26, 4, 26, true, 512);
    return hipSuccess;
}
#else
int ce_launch_fwd2_qp(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a) {
    switch (variant) {
    case 2: LAUNCH_F2(256, 4, 26, 2, 26, 4, 14, false, 256, true); break;
    case 3: LAUNCH_F2(512, 8, 20, 2, 32, 8, 8, false, 512, true); break;
    case 4: LAUNCH_F2(512, 4, 30, 4, 26, 4, 26, false, 512, true); break;
    default: return -1;
    }
    return 0;
}
hipError_t ce_setattr_fwd2_qp(int bytes) {
    SETATTR(4, 26, 2, 26, 4, 14, false, 256, true); SETATTR(8, 20, 2, 32, 8, 8, false, 512, true); SETATTR(4, 30, 4, 26, 4, 26, false, 512, true);
    return hipSuccess;
}
#endif
