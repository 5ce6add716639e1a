// translation unit: third-generation forward kernel (k_fwd3: products with lane-broadcast operands, ce_forward_v3.h)
#include "ce_tu_prologue.h"
namespace {
#include "ce_common.h"
#include "ce_expcone.h"
#include "ce_forward_rt.h"
#include "ce_forward_v2.h"
#include "ce_forward_v3.h"
}  // namespace

// one instantiation: the set-up layouts of k_fwd2's variant 2 ({4,26,2,26,4,14}: n <= 50, 104 y slots)
int ce_launch_fwd3(int B, size_t lds, hipStream_t st, const CeFwdArgs &a) {
    hipLaunchKernelGGL((k_fwd3<4, 26, 2, 26, 4, 14>), dim3(B), dim3(256), lds, st, a.T, a.S, a.Abm, a.q, a.sqk, a.sqb, a.idx_at, a.idx_ar, a.idx_b,
                       a.idx_at3, a.idx_ar3, a.slot_soc, a.x, a.y, a.s, a.iters, a.status, a.resid, a.row_perm);
    return 0;
}
hipError_t ce_setattr_fwd3(int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd3<4, 26, 2, 26, 4, 14>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
