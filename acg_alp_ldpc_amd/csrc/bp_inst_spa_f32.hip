// Instantiations of the fused BP kernel for T = float, ALGO = 0 (sum-product, fp32).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace acg {
#include "bp_core.inc"

constexpr int BP_NVP = 12;  // register-resident LLRs for codes with <= 12 variable passes (n <= 12 * L)

// variant: 0 = index table read from global, LLR in LDS; 1 = index table in LDS, LLR in LDS;
//          2 = index table in LDS, LLR in registers (degree <= 8 kernels only)
template <int MAXD, int L>
static const void *kptr(bool mc, int variant) {
    if (MAXD <= 8 && variant == 2)
        return mc ? (const void *) bp_fused_kernel<float, MAXD, L, 0, true, true, (MAXD <= 8 ? BP_NVP : 0)>
                  : (const void *) bp_fused_kernel<float, MAXD, L, 0, false, true, (MAXD <= 8 ? BP_NVP : 0)>;
    if (variant >= 1)
        return mc ? (const void *) bp_fused_kernel<float, MAXD, L, 0, true, true, 0>
                  : (const void *) bp_fused_kernel<float, MAXD, L, 0, false, true, 0>;
    return mc ? (const void *) bp_fused_kernel<float, MAXD, L, 0, true, false, 0>
              : (const void *) bp_fused_kernel<float, MAXD, L, 0, false, false, 0>;
}

template <int MAXD>
static const void *kptr_l(int L, bool mc, int variant) {
    switch (L) {
        case 64: return kptr<MAXD, 64>(mc, variant);
        case 32: return kptr<MAXD, 32>(mc, variant);
        case 16: return kptr<MAXD, 16>(mc, variant);
        default: return nullptr;
    }
}

const void *bp_kernel_ptr_spa_f32(int maxd, int L, bool mc, int variant) {
    if (maxd <= 8) return kptr_l<8>(L, mc, variant);
#ifndef ACG_FAST_BUILD
    if (maxd <= 16) return kptr_l<16>(L, mc, variant);
    if (maxd <= 32) return kptr_l<32>(L, mc, variant);
#endif
    return nullptr;
}

// tests only (acg_ldpc_debug_bp_trace): the headline instance (degree <= 8, index table in LDS, LLRs in registers) with
// the message dump compiled in
const void *bp_kernel_ptr_spa_f32_dbg(int L) {
    if (L == 64) return (const void *) bp_fused_kernel<float, 8, 64, 0, false, true, BP_NVP, true>;
    if (L == 32) return (const void *) bp_fused_kernel<float, 8, 32, 0, false, true, BP_NVP, true>;
    return nullptr;
}

}  // namespace acg
