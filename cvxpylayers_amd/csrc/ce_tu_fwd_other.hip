// translation unit: first-generation register-tiled forward kernel (k_forward_rt) and the size-generic forward kernel (k_forward)
#include "ce_tu_prologue.h"
namespace {
#include "ce_common.h"
#include "ce_expcone.h"
#include "ce_forward_rt.h"        // (first: group_reduce / DPP helpers used by the size-generic kernel's global-memory products)
#include "ce_global_mv.h"
#include "ce_forward_v2.h"        // (psd_project: the workgroup-parallel Jacobi projection shared with k_fwd2<PSD>)
#include "ce_forward_generic.h"
}  // namespace

int ce_launch_fwd_rt(int variant, int B, size_t lds, hipStream_t st, const CeFwdArgs &a) {
#define LAUNCH_RT(...) hipLaunchKernelGGL((k_forward_rt<__VA_ARGS__>), dim3(B), dim3(NT2), lds, st, a.T, a.S, a.Abm, a.q, a.sqk, a.sqb, a.x, a.y, a.s, a.iters, a.status, a.resid, a.aa_ws)
    switch (variant) {
    case 0: LAUNCH_RT(8, 13, 7, 4, 13, 160, 4); break;
    case 1: LAUNCH_RT(8, 16, 8, 4, 16, 208, 4); break;
    case 2: LAUNCH_RT(4, 32, 32, 4, 32, 272, 2); break;
    default: return -1;
    }
#undef LAUNCH_RT
    return 0;
}
int ce_launch_fwd_generic(int mode, int B, size_t lds, hipStream_t st, const CeFwdArgs &a) {
#define LAUNCH_F(AL, GL) hipLaunchKernelGGL((k_forward<AL, GL>), dim3(B), dim3(NT), lds, st, a.T, a.S, a.Abm, a.q, a.sqk, a.sqb, a.x, a.y, a.s, a.iters, a.status, a.resid, a.gA, a.gG, a.aa_ws)
    switch (mode) {
    case 0: LAUNCH_F(true, true); break;
    case 1: LAUNCH_F(true, false); break;
    case 2: LAUNCH_F(false, false); break;
    default: return -1;
    }
#undef LAUNCH_F
    return 0;
}
#define SETATTR(kern) do { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e_ != hipSuccess) return e_; } while (0)
hipError_t ce_setattr_fwd_rt(int bytes) {
    SETATTR((k_forward_rt<8, 13, 7, 4, 13, 160, 4>)); SETATTR((k_forward_rt<8, 16, 8, 4, 16, 208, 4>)); SETATTR((k_forward_rt<4, 32, 32, 4, 32, 272, 2>));
    return hipSuccess;
}
hipError_t ce_setattr_fwd_generic(int bytes) {
    SETATTR((k_forward<true, true>)); SETATTR((k_forward<true, false>)); SETATTR((k_forward<false, false>));
    return hipSuccess;
}
