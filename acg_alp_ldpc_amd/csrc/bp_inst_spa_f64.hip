// Instantiations of the fused BP kernel for T = double, ALGO = 0 (sum-product, fp64).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace acg {
#include "bp_core.inc"

template <int MAXD, int L>
static const void *kptr(bool mc, bool idxlds) {
    if (mc) return idxlds ? (const void *) bp_fused_kernel<double, MAXD, L, 0, true, true>
                          : (const void *) bp_fused_kernel<double, MAXD, L, 0, true, false>;
    return idxlds ? (const void *) bp_fused_kernel<double, MAXD, L, 0, false, true>
                  : (const void *) bp_fused_kernel<double, MAXD, L, 0, false, false>;
}

template <int MAXD>
static const void *kptr_l(int L, bool mc, bool idxlds) {
    switch (L) {
        case 64: return kptr<MAXD, 64>(mc, idxlds);
        case 32: return kptr<MAXD, 32>(mc, idxlds);
        case 16: return kptr<MAXD, 16>(mc, idxlds);
        default: return nullptr;
    }
}

const void *bp_kernel_ptr_spa_f64(int maxd, int L, bool mc, bool idxlds) {
    if (maxd <= 8) return kptr_l<8>(L, mc, idxlds);
#ifndef ACG_FAST_BUILD
    if (maxd <= 16) return kptr_l<16>(L, mc, idxlds);
    if (maxd <= 32) return kptr_l<32>(L, mc, idxlds);
#endif
    return nullptr;
}

}  // namespace acg
