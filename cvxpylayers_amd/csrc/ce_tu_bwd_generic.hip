// translation unit: size-generic adjoint kernel (k_backward)
#include "ce_tu_prologue.h"
namespace {
#include "ce_common.h"
#include "ce_expcone.h"
#include "ce_forward_rt.h"        // group_reduce / DPP helpers
#include "ce_global_mv.h"
#include "ce_forward_v2.h"        // (psd_jacobi: the workgroup-parallel Jacobi eigensolver shared with k_fwd2<PSD> / k_backward_rt<PSD>)
#include "ce_backward.h"
}  // namespace

int ce_launch_bwd_generic(int mode, int B, size_t lds, hipStream_t st, const CeBwdArgs &a) {
#define LAUNCH_B(AL, KL) hipLaunchKernelGGL((k_backward<AL, KL>), dim3(B), dim3(NT), lds, st, a.T, a.nkcap, a.ldk, a.Abm, a.x, a.y, a.s, a.dx, a.dy, a.dA, a.dq, a.sdqk, a.sdqb, a.adj, a.gA, a.gK, a.fix)
    switch (mode) {
    case 0: LAUNCH_B(true, true); break;
    case 1: LAUNCH_B(true, false); break;
    case 2: LAUNCH_B(false, false); break;
    default: return -1;
    }
#undef LAUNCH_B
    return 0;
}
hipError_t ce_setattr_bwd_generic(int bytes) {
#define SETATTR(kern) do { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e_ != hipSuccess) return e_; } while (0)
    SETATTR((k_backward<true, true>)); SETATTR((k_backward<true, false>)); SETATTR((k_backward<false, false>));
    return hipSuccess;
}
