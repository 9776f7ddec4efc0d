// Fused QP-ADMM decoder kernel for gfx950 (MI355X).
//
// Replaces, for a whole batch of frames per launch, the reference's per-frame
//   ConstructADMMProblem   algo/qp_admm.h:13-102   (structure analysed once per H on the host: code.cpp)
//   DecodeQPADMM           algo/qp_admm.h:104-178  (the sweep below)
// and, in Monte-Carlo mode, transmit (utils/channel.h:18-26) + the classification of exp()
// (experiment.h:109-120, IsCodeword included because QP-ADMM always reports ok=true, qp_admm.h:177).
//
// Arithmetic notes that make the fp64 path reproduce the reference bit for bit:
//  * z_j = max(0, r_j - yl_j) and yl_j' = max(0, yl_j - r_j) (qp_admm.h:156-157) are the positive
//    and negative part of ONE number w_j = r_j - yl_j (a-b == -(b-a) exactly in IEEE-754), so a
//    single word of state per constraint row is kept and (z, yl) are re-derived from it exactly.
//  * the v-update adds its <=24 terms in the reference's construction order (qp_admm.h:134-138),
//    r_j subtracts its <=3 terms in ascending variable order (qp_admm.h:147-151), and this file is
//    compiled with -ffp-contract=off so no multiply-add is fused that x86-64 would not fuse.
//  * only the residual sum (qp_admm.h:158) is reduced in a different order (tree instead of
//    sequential); it is compared with eps_stop, never propagated.
//
// Mapping: L lanes of a wavefront own one frame; all state (w, v, q) is in LDS for the whole
// decode; a lane owns constraint groups (one per three-variable check) in the row phase and
// variables in the v phase.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/acg_ldpc.h"
#include "kernels.hpp"
#include "ldpc_internal.hpp"

namespace acg {

namespace {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                       uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1) {
    const float u1 = ((float) a + 0.5f) * 2.3283064365386963e-10f;
    const float u2 = (float) (b >> 8) * 5.9604644775390625e-08f;
    const float r = sqrtf(-2.0f * 0.693147180559945309f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

template <int L>
__device__ __forceinline__ bool group_any(bool pred, int g) {
    const unsigned long long b = __ballot(pred);
    if (L == 64) return b != 0ull;
    const unsigned long long mask = ((1ull << (L & 63)) - 1ull) << (g * L);
    return (b & mask) != 0ull;
}

template <int L, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace

struct AdmmDevTables {
    // row phase, per group slot (coalesced): three members in ASCENDING variable id, each packed as
    // id | wpos << 24 (0xFFFFFFFF = none); type per slot (0 = padding)
    const uint32_t *grp_mem;  // [3][G_pad]
    const uint8_t *grp_type;  // [G_pad]
    // v phase, per variable slot
    const int32_t *var_of_slot;  // [n_vpass*L] variable id or -1
    const int32_t *v_maxlist;    // [n_vpass]
    const int32_t *v_list_off;   // [n_vpass]
    const uint32_t *v_list;      // entries gslot | wpos << 20 | type << 22 ; padding -> zero slot, type 0
    const void *inv_coef;        // [n_vpass*L] T
    // workgroup-per-frame kernel (admm_block_kernel): the same structure pre-digested into LDS byte addresses
    const uint32_t *blk_mem;   // [3][G_pad] address of V[member k] | address of U[group][wpos k] << 16
    const uint32_t *blk_list;  // laid out like v_list: address of U[group][0] | "coefficient -1" flags (row r at bit 31-r)
    const int32_t *blk_mlw;    // [n_vpass][4] longest list among the variables of (pass, wavefront)
    const uint8_t *grp_type_slot;  // [G_pad] group type per slot of the workgroup-per-frame kernel (type-3 groups first)
    const uint8_t *blk_generic;    // [n_gpass][4] 1 = (pass, wavefront) holds one- or two-variable checks
    const int32_t *blk_cell;       // [n_vpass*L] V cell of the variable in that slot (-1 none)
    // syndrome (MC classification)
    const int32_t *row_ptr;
    const int32_t *edge_var;
    int32_t n, m, n_var, n_grp, n_gpass, n_vpass, G_pad, V_pad, zero_gslot, nwords;
    int32_t cell_is_slot;  // workgroup-per-frame kernel: 1 = the V cell of every variable is its thread slot (quasi-cyclic placement): no address table in registers
    int32_t U_slots;  // workgroup-per-frame kernel: U slots held in LDS (a multiple of 32, <= G_pad): thread slots beyond it are empty
    int32_t lds_bytes_per_frame;
};

struct AdmmDevice;
void admm_device_destroy(AdmmDevice *d);

struct AdmmDevice {
    AdmmDevTables t{};
    std::vector<void *> allocs;
    int L = 64, f32 = 0, block = 256, frames_per_block = 4, grid_cap = 256;
    bool reg = false;  // row state in registers (ADMM_NGP variant)
    bool blockmode = false;  // workgroup-per-frame kernel
    bool blk_lean = false;   // ... its instance without the general paths (see admm_block_kernel)
    const void *kernel[2] = {nullptr, nullptr};
    size_t lds_block = 0;
    bool guard = false;  // e_min*mu <= alpha  (qp_admm.h:108-114)
    bool budget0 = false;  // max_iter == 0: handled by admm_budget0_kernel
    double alpha = 0, mu = 0, eps = 0;
};

// NGP = 0: row state w in LDS (any code size).  NGP > 0 (requires n_gpass <= NGP): every lane keeps the w of its own
// constraint groups in registers and LDS holds u_j = yl_j + mu*(z_j - b_j) instead — exactly the term the v-update
// consumes (qp_admm.h:137), so its inner loop is one LDS read and one fma(+-1, u, B) per row (an fma with a +-1
// multiplier rounds exactly like the reference's `B += cf * (...)`).
template <typename T, int L, bool MC, int NGP>
__global__ void __launch_bounds__(256) admm_fused_kernel(const AdmmDevTables t, const DecodeArgs a, const T alpha,
                                                         const T mu, const T eps_stop) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FPW = 64 / L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane % L, g = lane / L;
    unsigned char *base = smem + (size_t) (wave * FPW + g) * t.lds_bytes_per_frame;
    T *W = reinterpret_cast<T *>(base);        // [4][G_pad]   w_j = r_j - yl_j
    T *V = W + 4 * t.G_pad;                     // [V_pad]      by variable id (+ one zero cell at n_var)
    T *Q = V + t.V_pad;                         // [n_vpass*L]  by variable slot
    uint32_t *OB = reinterpret_cast<uint32_t *>(Q + t.n_vpass * L);
    const T *inv_coef = reinterpret_cast<const T *>(t.inv_coef);

    // dynamic frame hand-out (see bp_core.inc): chunks of CHUNK frame indices from one global counter
    constexpr int CHUNK = 4 * FPW;
    int64_t wnext = 0, wend = 0;
    int64_t frame = 0;
    bool active = false, want = true, need_init = false;
    int it = 0;
    int ham = 0;
    unsigned int acc_correct = 0, acc_pseudo = 0, acc_total = 0;
    unsigned long long acc_ham = 0, acc_ham_ok = 0, acc_ham_wrong = 0, acc_iters = 0;
    bool converged = false;
    T wreg[NGP > 0 ? NGP : 1][4];
#pragma unroll
    for (int p = 0; p < (NGP > 0 ? NGP : 1); ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) wreg[p][r] = (T) 0;

    for (;;) {
        // ---- finish frames: residual below eps (qp_admm.h:161-163) or max_iter sweeps done ----
        const bool finish = active && !need_init && ((a.early_exit && converged) || it >= a.max_iter);
        if (__ballot(finish) != 0ull) {
            if (finish) {
                for (int w = l; w < t.nwords; w += L) OB[w] = 0u;
            }
            wave_sync();
            if (finish) {
                for (int v = l; v < t.n; v += L) {
                    const T val = V[v];
                    if (!(val <= (T) 0.5)) atomicOr(&OB[v >> 5], 1u << (v & 31));  // qp_admm.h:168-174
                }
            }
            wave_sync();
            if (finish) {
                if (a.out_bits)
                    for (int w = l; w < t.nwords; w += L) a.out_bits[(size_t) frame * t.nwords + w] = OB[w];
                if (l == 0) {
                    if (a.out_ok) a.out_ok[frame] = 1;  // qp_admm.h:165,177
                    if (a.out_iters) a.out_iters[frame] = it;
                }
                if (MC) {
                    // IsCodeword (experiment.h:111) then equality with the sent word (:112)
                    bool sbad = false;
                    for (int c = l; c < t.m; c += L) {
                        uint32_t s = 0;
                        for (int e = t.row_ptr[c]; e < t.row_ptr[c + 1]; ++e) {
                            const int v = t.edge_var[e];
                            s ^= (OB[v >> 5] >> (v & 31)) & 1u;
                        }
                        sbad |= (s != 0u);
                    }
                    bool neq = false;
                    const int64_t gf = a.first_frame + frame;
                    for (int w = l; w < t.nwords; w += L) {
                        const uint32_t cwv = a.cw_packed ? a.cw_packed[(size_t) (gf % a.n_cw) * t.nwords + w] : 0u;
                        neq |= (OB[w] != cwv);
                    }
                    const bool is_cw = !group_any<L>(sbad, g);
                    const bool differ = group_any<L>(neq, g);
                    const bool correct = is_cw && !differ;
                    acc_correct += correct;
                    acc_pseudo += (is_cw && differ);
                    acc_total += 1;
                    acc_ham += ham;
                    acc_ham_ok += correct ? ham : 0;
                    acc_ham_wrong += correct ? 0 : ham;
                    acc_iters += it;
                }
                active = false;
                want = true;
            }
        }
        unsigned long long wm = __ballot(want && l == 0);
        while (wm != 0ull) {
            const int leader = __builtin_ctzll(wm);
            wm &= wm - 1ull;
            if (wnext >= wend) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(a.work_counter, (unsigned long long) CHUNK);
                const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t) base);
                const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32));
                wnext = (int64_t) (((unsigned long long) bhi << 32) | blo);
                wend = wnext + CHUNK < a.frames ? wnext + CHUNK : a.frames;
            }
            const bool got = wnext < wend;
            const int64_t f = wnext;
            if (got) wnext += 1;
            if (lane / L == leader / L) {
                frame = f;
                active = got;
                need_init = got;
                want = false;
            }
        }
        if (__ballot(active) == 0ull) break;

        // ---- (re)start groups on a new frame ---------------------------------------------------
        if (__ballot(need_init) != 0ull) {
            wave_sync();
            const uint32_t *cw = nullptr;
            if (MC && need_init) {
                const int64_t gf = a.first_frame + frame;
                if (a.cw_packed) cw = a.cw_packed + (size_t) (gf % a.n_cw) * t.nwords;
                const int nq = (t.n + 3) >> 2;
                for (int q = l; q < nq; q += L) {
                    uint32_t r[4];
                    philox((uint32_t) gf, (uint32_t) (gf >> 32), (uint32_t) q, 0u, (uint32_t) a.seed,
                           (uint32_t) (a.seed >> 32), r);
                    float z[4];
                    box_muller(r[0], r[1], z[0], z[1]);
                    box_muller(r[2], r[3], z[2], z[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = 4 * q + e;
                        if (v < t.n) {
                            const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
                            V[v] = (T) __builtin_fmaf(a.sigma, z[e], bit ? -1.0f : 1.0f);  // explicit fma: same symbol in every TU (this one is built with -ffp-contract=off)
                        }
                    }
                }
            }
            wave_sync();
            int my_ham = 0;
            if (need_init) {
                for (int p = 0; p < t.n_vpass; ++p) {
                    const int slot = p * L + l;
                    const int i = t.var_of_slot[slot];
                    T q = (T) 0;  // auxiliaries: q = 0 (qp_admm.h:24)
                    if (i >= 0 && i < t.n) {
                        if (MC) {
                            const T yv = V[i];
                            const uint32_t bit = cw ? ((cw[i >> 5] >> (i & 31)) & 1u) : 0u;
                            my_ham += ((!bit && yv <= (T) 0) || (bit && yv > (T) 0)) ? 1 : 0;
                            q = (T) (2 * (double) yv / a.var);
                        } else if (a.y_is_f64) {
                            q = (T) (2 * reinterpret_cast<const double *>(a.y)[(size_t) frame * t.n + i] / a.var);
                        } else {
                            q = (T) (2 * (double) reinterpret_cast<const float *>(a.y)[(size_t) frame * t.n + i] / a.var);
                        }
                    }
                    Q[slot] = q;  // CalculateCoef, algo/algo.h:13-20
                }
            }
            wave_sync();
            if (MC) {
                const int hs = group_sum<L, int>(need_init ? my_ham : 0);
                if (need_init) ham = hs;
            }
            if (need_init) {
                if (NGP > 0) {
                    // z = yl = 0 (qp_admm.h:120-121): w = 0 in registers, u = 0 + mu*(0 - b) in LDS
#pragma unroll
                    for (int p = 0; p < (NGP > 0 ? NGP : 1); ++p)
                        if (p < t.n_gpass) {
                            const int gs = p * L + l;
                            const int ty = t.grp_type[gs];
#pragma unroll
                            for (int row = 0; row < 4; ++row) {
                                wreg[p][row] = (T) 0;
                                const T b = (ty == 3 && row == 3) ? (T) 2 : (T) 0;
                                W[row * t.G_pad + gs] = (T) 0 + mu * ((T) 0 - b);
                            }
                        }
                } else {
                    for (int w = l; w < 4 * t.G_pad; w += L) W[w] = (T) 0;  // z = yl = 0 (qp_admm.h:120-121)
                }
                for (int w = l; w < t.V_pad; w += L) V[w] = (T) 0;
                it = 0;
                converged = false;
                need_init = false;
            }
            wave_sync();
        }

        // ---- one ADMM sweep (qp_admm.h:130-164) -------------------------------------------------
        // v-update (qp_admm.h:132-142)
        for (int p = 0; p < t.n_vpass; ++p) {
            const int slot = p * L + l;
            const int ml = t.v_maxlist[p];
            const uint32_t *lp = t.v_list + t.v_list_off[p] + l;
            T B = Q[slot] + (alpha / 2);
            for (int k = 0; k < ml; ++k) {
                const uint32_t ent = lp[(size_t) k * L];
                const int gs = (int) (ent & 0xFFFFFu);
                const int wp = (int) ((ent >> 20) & 3u);
                const int ty = (int) (ent >> 22);
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    const bool plus = (ty == 3 && row == 3) || (row == wp);
                    if (NGP > 0) {
                        const T u = W[row * t.G_pad + gs];  // u_j = yl_j + mu*(z_j - b_j), stored by the row phase
                        B = __builtin_fma(plus ? (T) 1 : (T) -1, u, B);
                    } else {
                        const T w = W[row * t.G_pad + gs];
                        const T z = ((T) 0 < w) ? w : (T) 0;
                        const T nw = -w;
                        const T yl = ((T) 0 < nw) ? nw : (T) 0;
                        const T b = (ty == 3 && row == 3) ? (T) 2 : (T) 0;
                        const T term = yl + mu * (z - b);
                        B += plus ? term : -term;
                    }
                }
            }
            T v = B * inv_coef[slot];
            v = (v < (T) 0) ? (T) 0 : v;  // std::max(v, 0.0)
            v = ((T) 1 < v) ? (T) 1 : v;  // std::min(v, 1.0)
            const int i = t.var_of_slot[slot];
            if (active && i >= 0) V[i] = v;
        }
        wave_sync();
        // residual, multiplier and slack update (qp_admm.h:144-159), one lane per constraint group
        T sum2 = (T) 0;
        if (NGP > 0) {
#pragma unroll
            for (int p = 0; p < (NGP > 0 ? NGP : 1); ++p)
                if (p < t.n_gpass) {
            const int gs = p * L + l;
                const int ty = t.grp_type[gs];
                T vm[3];
                int wp[3];
                bool have[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t e = t.grp_mem[(size_t) k * t.G_pad + gs];
                    have[k] = (e != 0xFFFFFFFFu);
                    const int id = have[k] ? (int) (e & 0xFFFFFFu) : t.n_var;  // V[n_var] is a zero cell
                    wp[k] = (int) (e >> 24);
                    vm[k] = V[id];
                }
                const int rows = (ty == 3) ? 4 : ty;
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    T r = (ty == 3 && row == 3) ? (T) 2 : (T) 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const bool plus = (ty == 3 && row == 3) || (row == wp[k]);
                        const T prod = plus ? vm[k] : -vm[k];
                        r = have[k] ? (r - prod) : r;
                    }
                    const T wo = wreg[p][row];
                    const T nwo = -wo;
                    const T ylo = ((T) 0 < nwo) ? nwo : (T) 0;
                    const T wn = r - ylo;
                    const T z = ((T) 0 < wn) ? wn : (T) 0;
                    if (active && row < rows) {
                        wreg[p][row] = wn;
                            const T nwn = -wn;
                            const T yln = ((T) 0 < nwn) ? nwn : (T) 0;
                            const T bb = (ty == 3 && row == 3) ? (T) 2 : (T) 0;
                            W[row * t.G_pad + gs] = yln + mu * (z - bb);
                        const T d = z - r;
                        sum2 += d * d;
                    }
                }
                }
        } else {
            for (int p = 0; p < t.n_gpass; ++p) {
            const int gs = p * L + l;
                const int ty = t.grp_type[gs];
                T vm[3];
                int wp[3];
                bool have[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t e = t.grp_mem[(size_t) k * t.G_pad + gs];
                    have[k] = (e != 0xFFFFFFFFu);
                    const int id = have[k] ? (int) (e & 0xFFFFFFu) : t.n_var;  // V[n_var] is a zero cell
                    wp[k] = (int) (e >> 24);
                    vm[k] = V[id];
                }
                const int rows = (ty == 3) ? 4 : ty;
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    T r = (ty == 3 && row == 3) ? (T) 2 : (T) 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const bool plus = (ty == 3 && row == 3) || (row == wp[k]);
                        const T prod = plus ? vm[k] : -vm[k];
                        r = have[k] ? (r - prod) : r;
                    }
                    const T wo = W[row * t.G_pad + gs];
                    const T nwo = -wo;
                    const T ylo = ((T) 0 < nwo) ? nwo : (T) 0;
                    const T wn = r - ylo;
                    const T z = ((T) 0 < wn) ? wn : (T) 0;
                    if (active && row < rows) {
                        W[row * t.G_pad + gs] = wn;
                        const T d = z - r;
                        sum2 += d * d;
                    }
                }
            }
        }
        wave_sync();
        // (as in admm_block_kernel: a lane whose partial sum of squares already reaches eps settles "not converged" for its
        // frame; the fp64 reduction runs only when some frame of the wavefront has no such lane)
        const bool big = group_any<L>(sum2 >= eps_stop, g);
        it += 1;
        if (__ballot(!big) != 0ull) {
            sum2 = group_sum<L, T>(sum2);
            converged = (sum2 < eps_stop);
        } else {
            converged = false;
        }
    }

    if (MC && l == 0 && acc_total) {
        atomicAdd(&a.counters[MC_CORRECT], (unsigned long long) acc_correct);
        atomicAdd(&a.counters[MC_PSEUDO], (unsigned long long) acc_pseudo);
        atomicAdd(&a.counters[MC_TOTAL], (unsigned long long) acc_total);
        atomicAdd(&a.counters[MC_HAM], acc_ham);
        atomicAdd(&a.counters[MC_HAM_OK], acc_ham_ok);
        atomicAdd(&a.counters[MC_HAM_WRONG], acc_ham_wrong);
        atomicAdd(&a.counters[MC_ITERS], acc_iters);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Workgroup-per-frame variant: the 128/192/256 threads of a workgroup own ONE frame (requires <= 4 passes of
// constraint groups and variables).  Same arithmetic and the same order of every rounding step as the wavefront
// variant, organised so that nothing loop-invariant is recomputed inside a sweep:
//  * everything a thread needs about its groups and variables (LDS addresses, signs, list entries, channel term,
//    1/coefficient) is loaded ONCE per kernel / frame into registers; a sweep touches no global memory;
//  * U is stored [group][row], so the v-update fetches the four rows of a list entry with one 128-bit (fp32) or two
//    (fp64) LDS reads and one address; the +-1 coefficient of a row is rebuilt from a flag bit of the entry
//    (sign bit OR'ed onto the bit pattern of 1.0) and applied with an fma, which rounds like `B += cf * (...)`;
//  * the row phase works in "member order": slot k of a group is the row in which member k (ascending variable id)
//    has coefficient +1, i.e. row wpos[k] (qp_admm.h:48-70); the three residuals of those rows are
//        (-v0 + v1) + v2,  (v0 - v1) + v2,  (v0 + v1) - v2
//    — exactly what `r -= A_jk * v_k` in ascending variable order gives, with the common sub-expression v0 - v1
//    shared (-(a - b) == b - a exactly) — and row 3 is ((2 - v0) - v1) - v2.  The row state w lives in registers in
//    slot order; the result of slot k is stored to U[group][wpos[k]] through a precomputed address, so the v-update
//    still reads rows in construction order;
//  * yl + mu*(z - b) with b = 0 is max(mu*z, -w) (one of yl, z is zero, x + 0 is exact);
//  * the residual sum is only computed when the stopping rule is on (template EE), reduced with DPP lane
//    exchanges instead of LDS permutes.
constexpr int ADMM_BLK = 256;
constexpr int ADMM_BP = 4;  // most passes any instance handles
constexpr int ADMM_VK = 6;  // list entries per variable slot kept in registers (longer lists continue from global)
// ... of pass p: the host deals the variables out longest list first, so later passes hold the short lists (H05: 6/5/4/4,
// then 3/2/2/2, then 2/2/2/-) and need fewer registers; whatever is longer continues from global memory as well
__host__ __device__ constexpr int admm_vk(int p) { return p == 0 ? ADMM_VK : p == 1 ? 4 : 2; }
#ifndef ADMM_OCC
#define ADMM_OCC 8  // launch bound: ADMM_OCC - BP workgroups of 4 wavefronts per CU
#endif
#ifndef ADMM_OCC_F32
#define ADMM_OCC_F32 2  // extra workgroups per CU asked of the fp32 instances (fewer registers, half the LDS): measured best
#endif

template <typename T> struct AdmmVec;
template <> struct AdmmVec<double> {
    typedef double v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ double pm1(uint32_t x) {  // bit 31 of x set -> -1.0, else +1.0
        return __hiloint2double((int) ((x & 0x80000000u) | 0x3FF00000u), 0);
    }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ double max(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return __builtin_fmin(a, b); }
    static __device__ __forceinline__ double clamp01(double x) {
        double r;
        asm("v_max_f64 %0, %1, %1 clamp" : "=v"(r) : "v"(x));
        return r;
    }
};
template <> struct AdmmVec<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float pm1(uint32_t x) { return __uint_as_float((x & 0x80000000u) | 0x3F800000u); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float max(float a, float b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ float min(float a, float b) { return __builtin_fminf(a, b); }
    static __device__ __forceinline__ float clamp01(float x) {
        float r;
        asm("v_max_f32 %0, %1, %1 clamp" : "=v"(r) : "v"(x));
        return r;
    }
};

__device__ __forceinline__ uint32_t dpp_u32(uint32_t v, const int ctrl_tag) {
    // ctrl_tag: 0 quad_perm[1,0,3,2]  1 quad_perm[2,3,0,1]  2 row_half_mirror  3 row_mirror
    switch (ctrl_tag) {
        case 0: return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0xB1, 0xF, 0xF, true);
        case 1: return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x4E, 0xF, 0xF, true);
        case 2: return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x141, 0xF, 0xF, true);
        default: return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x140, 0xF, 0xF, true);
    }
}

// sum over the 64 lanes of a wavefront, identical in every lane (fixed order: butterfly inside rows of 16, then rows)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint32_t lo = dpp_u32((uint32_t) __double2loint(v), s), hi = dpp_u32((uint32_t) __double2hiint(v), s);
        v += __hiloint2double((int) hi, (int) lo);
    }
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k), __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
    return ((r[0] + r[1]) + r[2]) + r[3];
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 0; s < 4; ++s) v += __uint_as_float(dpp_u32(__float_as_uint(v), s));
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __uint_as_float((uint32_t) __builtin_amdgcn_readlane((int) __float_as_uint(v), 16 * k));
    return ((r[0] + r[1]) + r[2]) + r[3];
}

// The four rows of a U slot (32 words apart inside its tile).  fp64: four ds_read_b64, spelled out because the
// compiler would pair them into ds_read2_b64, which the LDS serves 16 lanes at a time at half the rate
// (MI355X_MICROARCH.md, LDS table) — and the static placement (placement_optimise) is tuned for the 32-lane groups of ds_read_b64.
__device__ __forceinline__ void admm_read_rows(unsigned char *smem, const uint32_t off, double &x, double &y, double &z, double &w) {
    const uint32_t addr = off + (uint32_t) (uintptr_t) smem;  // LDS pointers are 32-bit offsets
    asm("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:256\n\tds_read_b64 %2, %4 offset:512\n\tds_read_b64 %3, %4 offset:768\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w)
        : "v"(addr));
}
__device__ __forceinline__ void admm_read_rows(unsigned char *smem, const uint32_t off, float &x, float &y, float &z, float &w) {
    const float *pu = reinterpret_cast<const float *>(smem + off);
    x = pu[0];
    y = pu[32];
    z = pu[64];
    w = pu[96];
}

// The first ADMM_VK entries of a variable's list as ONE asm statement: a two-deep pipeline of the row reads (the four
// ds_read of entry k+1 are in flight while the four fma/add of entry k run; LDS operations complete in order, so
// `s_waitcnt lgkmcnt(4)` after issuing entry k+1 means entry k has arrived).  It has to be a single statement: a value the
// LDS has not delivered yet must never be visible to the compiler, which is free to copy it (measured: it does).  The
// buffers are fixed registers v40..v59 named as clobbers, because a 64-bit operand cannot name its high half, which
// the +-1 multiplier is built in ((flag & 0x80000000) | high word of 1.0).  Same operations in the same order as
// admm_read_rows + fma: B = fma(+-1, u0, B); fma(+-1, u1, B); fma(+-1, u2, B); B + u3 per entry.  The entries carry
// ABSOLUTE LDS addresses in their low halves (the kernel adds the start of its dynamic LDS when it loads them): one
// v_and_b32 per entry, no SDWA add (half rate).
#define ACH_ISSUE(RD, O1, O2, O3, X0, X1, X2, X3, E)                                                                       \
    "v_and_b32 v59, 0xffff, " E "\n\t" RD " " X0 ", v59\n\t" RD " " X1 ", v59 offset:" O1 "\n\t" RD " " X2 ", v59 offset:" O2 \
    "\n\t" RD " " X3 ", v59 offset:" O3 "\n\t"
// The +-1 multipliers of rows 0..2 (flag "coefficient is -1" of row r at bit 31-r of the entry): row 0 by and-or; row 1
// from e + e (v_add_u32 issues at twice the rate of v_lshlrev_b32, profiles/r02_valu_op_rates.txt); row 2 by
// exclusive-or of the first two sign bits — exactly one of the three rows has coefficient +1 (the row in which the member
// sits, qp_admm.h:48-70), so s2 = s0 ^ s1; for the one- and two-variable checks the rows beyond hold u = 0 and the sign
// does not matter.  17 issue cycles of integer work per entry instead of 21.
#define ACH_ACC(FMA, ADD, PMH, PM, X0, X1, X2, X3, E)                                                                      \
    "v_and_or_b32 " PMH ", " E ", %[k80], %[one]\n\t" FMA " %[B], " PM ", " X0 ", %[B]\n\t"                                  \
    "v_add_u32 v58, " E ", " E "\n\tv_and_b32 v59, %[k80], " PMH "\n\tv_and_or_b32 " PMH ", v58, %[k80], %[one]\n\t" FMA        \
    " %[B], " PM ", " X1 ", %[B]\n\t"                                                                                        \
    "v_xor_b32 " PMH ", v59, " PMH "\n\t" FMA " %[B], " PM ", " X2 ", %[B]\n\t" ADD " %[B], " X3 ", %[B]\n\t"
#define ACH_BODY(ISS_A, ISS_B, ACC_A, ACC_B)                                                                               \
    ISS_A("%[e0]") "s_cmp_lt_u32 %[ml], 2\n\ts_cbranch_scc1 .Lach0_%=\n\t"                                                 \
    ISS_B("%[e1]") "s_waitcnt lgkmcnt(4)\n\t" ACC_A("%[e0]") "s_cmp_lt_u32 %[ml], 3\n\ts_cbranch_scc1 .Lach1_%=\n\t"      \
    ISS_A("%[e2]") "s_waitcnt lgkmcnt(4)\n\t" ACC_B("%[e1]") "s_cmp_lt_u32 %[ml], 4\n\ts_cbranch_scc1 .Lach2_%=\n\t"      \
    ISS_B("%[e3]") "s_waitcnt lgkmcnt(4)\n\t" ACC_A("%[e2]") "s_cmp_lt_u32 %[ml], 5\n\ts_cbranch_scc1 .Lach3_%=\n\t"      \
    ISS_A("%[e4]") "s_waitcnt lgkmcnt(4)\n\t" ACC_B("%[e3]") "s_cmp_lt_u32 %[ml], 6\n\ts_cbranch_scc1 .Lach4_%=\n\t"      \
    ISS_B("%[e5]") "s_waitcnt lgkmcnt(4)\n\t" ACC_A("%[e4]")                                                              \
    "s_waitcnt lgkmcnt(0)\n\t" ACC_B("%[e5]") "s_branch .Lachend_%=\n"                                                     \
    ".Lach0_%=:\n\ts_waitcnt lgkmcnt(0)\n\t" ACC_A("%[e0]") "s_branch .Lachend_%=\n"                                        \
    ".Lach1_%=:\n\ts_waitcnt lgkmcnt(0)\n\t" ACC_B("%[e1]") "s_branch .Lachend_%=\n"                                        \
    ".Lach2_%=:\n\ts_waitcnt lgkmcnt(0)\n\t" ACC_A("%[e2]") "s_branch .Lachend_%=\n"                                        \
    ".Lach3_%=:\n\ts_waitcnt lgkmcnt(0)\n\t" ACC_B("%[e3]") "s_branch .Lachend_%=\n"                                        \
    ".Lach4_%=:\n\ts_waitcnt lgkmcnt(0)\n\t" ACC_A("%[e4]") ".Lachend_%=:"
#define ACH64_ISS_A(E) ACH_ISSUE("ds_read_b64", "256", "512", "768", "v[40:41]", "v[42:43]", "v[44:45]", "v[46:47]", E)
#define ACH64_ISS_B(E) ACH_ISSUE("ds_read_b64", "256", "512", "768", "v[48:49]", "v[50:51]", "v[52:53]", "v[54:55]", E)
#define ACH64_ACC_A(E) ACH_ACC("v_fma_f64", "v_add_f64", "v57", "v[56:57]", "v[40:41]", "v[42:43]", "v[44:45]", "v[46:47]", E)
#define ACH64_ACC_B(E) ACH_ACC("v_fma_f64", "v_add_f64", "v57", "v[56:57]", "v[48:49]", "v[50:51]", "v[52:53]", "v[54:55]", E)
#define ACH32_ISS_A(E) ACH_ISSUE("ds_read_b32", "128", "256", "384", "v40", "v41", "v42", "v43", E)
#define ACH32_ISS_B(E) ACH_ISSUE("ds_read_b32", "128", "256", "384", "v44", "v45", "v46", "v47", E)
#define ACH32_ACC_A(E) ACH_ACC("v_fma_f32", "v_add_f32", "v56", "v56", "v40", "v41", "v42", "v43", E)
#define ACH32_ACC_B(E) ACH_ACC("v_fma_f32", "v_add_f32", "v56", "v56", "v44", "v45", "v46", "v47", E)
static_assert(ADMM_VK == 6, "the asm chain below is written for six register-resident entries");
__device__ __forceinline__ void admm_v_chain(const uint32_t (&e)[ADMM_VK], const uint32_t ml, const uint32_t k80, const uint32_t one_hi, double &B) {
    asm volatile("v_mov_b32 v56, 0\n\t" ACH_BODY(ACH64_ISS_A, ACH64_ISS_B, ACH64_ACC_A, ACH64_ACC_B)
                 : [B] "+v"(B)
                 : [ml] "s"(ml), [k80] "s"(k80), [one] "v"(one_hi), [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]),
                   [e4] "v"(e[4]), [e5] "v"(e[5])
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
                   "v59", "scc", "memory");
}
__device__ __forceinline__ void admm_v_chain(const uint32_t (&e)[ADMM_VK], const uint32_t ml, const uint32_t k80, const uint32_t one_hi, float &B) {
    asm volatile(ACH_BODY(ACH32_ISS_A, ACH32_ISS_B, ACH32_ACC_A, ACH32_ACC_B)
                 : [B] "+v"(B)
                 : [ml] "s"(ml), [k80] "s"(k80), [one] "v"(one_hi), [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]),
                   [e4] "v"(e[4]), [e5] "v"(e[5])
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v56", "v58", "v59", "scc", "memory");
}

// One constraint group (slot order, see above).  GENERIC also handles one- and two-variable checks, whose missing
// slots must stay all-zero; the host puts those groups into passes/wavefronts flagged for the GENERIC instance so the
// common instance carries no selects.  State yl[] = max(0, yl - r) of the previous sweep (qp_admm.h:157).
template <typename T, bool EE, bool GENERIC>
__device__ __forceinline__ void admm_group_update(unsigned char *smem, const uint32_t lds0, const uint32_t m0, const uint32_t m1,
                                                  const uint32_t m2, const uint32_t u3_addr, const uint32_t ty, const T mu,
                                                  T (&yl)[4], T &sum2) {
    using X = AdmmVec<T>;
    // (m0..m2 are opaque copies made inside the sweep loop, so these stay one v_and_b32 / v_lshrrev_b32 each — full-rate
    // operations, unlike the SDWA add they replace — instead of being hoisted into seven address registers per pass)
    const T v0 = *reinterpret_cast<const T *>(smem + (m0 & 0xFFFFu));
    const T v1 = *reinterpret_cast<const T *>(smem + (m1 & 0xFFFFu));
    const T v2 = *reinterpret_cast<const T *>(smem + (m2 & 0xFFFFu));
    const T d01 = v0 - v1;
    T r[4];
    r[0] = v2 - d01;        // ((0 - v0) + v1) + v2
    r[1] = d01 + v2;        // ((0 + v0) - v1) + v2
    r[2] = (v0 + v1) - v2;  // ((0 + v0) + v1) - v2
    r[3] = (((T) 2 - v0) - v1) - v2;
    if (GENERIC && ty != 3u) {
        if (ty < 2u) r[1] = (T) 0;
        r[2] = (T) 0;
        r[3] = (T) 0;
    }
    T u[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const T wn = r[s] - yl[s];
        const T z = X::max(wn, (T) 0);
        yl[s] = X::max(-wn, (T) 0);
        // rows 0..2: yl + mu*(z - 0), and one of yl, z is zero (x + 0 is exact), so it is max(mu*z, -wn); without the
        // stopping rule z itself is not needed: mu*max(wn, 0) == max(mu*wn, 0) for mu > 0 (the same product, or zero),
        // and one of mu*wn, -wn is >= 0, so the value is max(mu*wn, -wn) — one operation less
        if (s < 3) u[s] = EE ? X::max(mu * z, -wn) : X::max(mu * wn, -wn);
        else u[s] = yl[s] + mu * (z - (T) 2);
        if (EE) {
            const T dd = z - r[s];
            sum2 += dd * dd;
        }
    }
    if (GENERIC && ty != 3u) u[3] = (T) 0;
    *reinterpret_cast<T *>(smem + (m0 >> 16)) = u[0];
    *reinterpret_cast<T *>(smem + (m1 >> 16)) = u[1];
    *reinterpret_cast<T *>(smem + (m2 >> 16)) = u[2];
    *reinterpret_cast<T *>(smem + (u3_addr + lds0)) = u[3];
}

// BP = passes (of blockDim.x constraint groups / variables) the register-resident structure is sized for; fewer passes
// = fewer registers = more wavefronts per SIMD (launch bound: 4, 5, 6 workgroups of 4 wavefronts per CU for BP = 4, 3, 2).
// LEAN: the instance for problems that need none of the general paths inside the sweep — no one-/two-variable checks, every
// list within the register-resident entries of its pass, V cell = thread slot (all true for the quasi-cyclic tuple placement
// of H05 / optimalH).  The wavefronts of this kernel are bound by how many instructions they have to get through per sweep,
// scalar tests and branches included, so the paths are compiled out, not branched around.
template <typename T, bool EE, int BP, bool LEAN>
__global__ void __launch_bounds__(ADMM_BLK, ADMM_OCC - BP + (sizeof(T) == 4 ? ADMM_OCC_F32 : 0)) admm_block_kernel(const AdmmDevTables t, const DecodeArgs a, const T alpha,
                                                              const T mu, const T eps_stop) {
    using X = AdmmVec<T>;
    extern __shared__ __attribute__((aligned(32))) unsigned char smem[];
    const int L = blockDim.x;  // 128, 192 or 256 threads = one frame
    __builtin_amdgcn_s_setreg((0 << 11) | (8 << 6) | 1, 0);  // hwreg(HW_REG_MODE, offset 8, width 1) = DX10_CLAMP := 0: the clamp modifier passes NaN (see the v-update)
    __shared__ T red[4];
    __shared__ unsigned long long fr_lds;
    const int l = threadIdx.x, lane = l & 63;
    const int wave = __builtin_amdgcn_readfirstlane(l >> 6);
    // LDS: V[V_pad] by variable id (+ zero cell at n_var) | U[G_pad][4] | packed hard decisions
    T *V = reinterpret_cast<T *>(smem);
    const uint32_t u_base = (uint32_t) t.V_pad * (uint32_t) sizeof(T);
    uint32_t *OB = reinterpret_cast<uint32_t *>(smem + u_base + (size_t) 4 * t.U_slots * sizeof(T));
    // ---- loop-invariant per-thread structure -> registers ------------------------------------------------------------
    uint32_t mem[BP][3];  // member k of my group in pass p: LDS byte address of V[member] | address of U[group][wpos k] << 16
    uint32_t tys = 0;          // group type of pass p at bits 2p..2p+1 (0 = padding slot)
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        mem[p][0] = mem[p][1] = mem[p][2] = 0;
        if (p < t.n_gpass) {
            const int gs = p * L + l;
            tys |= (uint32_t) t.grp_type_slot[gs] << (2 * p);
#pragma unroll
            for (int k = 0; k < 3; ++k) mem[p][k] = t.blk_mem[(size_t) k * t.G_pad + gs];
        }
    }
    uint32_t ent[BP][ADMM_VK];  // list entries: ABSOLUTE LDS byte address of U[group][0] | "coefficient is -1" flags, row r at bit 31-r
    const uint32_t smem_abs = (uint32_t) (uintptr_t) smem;  // start of the dynamic LDS (behind the static words)
    uint32_t mlw_pk = 0;             // list length of (pass p, my wavefront) at bits 8p..8p+7
    uint32_t gen_pk = 0;             // bit p: (pass p, my wavefront) holds one- or two-variable checks
    // LDS byte address of my variable in pass p: its thread slot when the placement says so (cell_is_slot), otherwise read
    // from the table once per sweep (ahead of the list, so the list hides the load); bit 8+p of tys = "I own a variable"
    const bool cell_slot = LEAN || t.cell_is_slot != 0;
    T inv[BP];
    const T *inv_coef = reinterpret_cast<const T *>(t.inv_coef);
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        inv[p] = (T) 0;
#pragma unroll
        for (int k = 0; k < ADMM_VK; ++k)
            if (k < admm_vk(p)) ent[p][k] = 0;
        if (p < t.n_gpass) gen_pk |= (uint32_t) (t.blk_generic[p * 4 + wave] != 0) << p;
        if (p < t.n_vpass) {
            const int ml = t.blk_mlw[p * 4 + wave];
            mlw_pk |= (uint32_t) ml << (8 * p);
            const int cell = t.blk_cell[p * L + l];
            tys |= (cell >= 0 ? 1u : 0u) << (8 + p);
            inv[p] = inv_coef[p * L + l];
#pragma unroll
            for (int k = 0; k < ADMM_VK; ++k)
                if (k < admm_vk(p) && k < ml) ent[p][k] = t.blk_list[(size_t) t.v_list_off[p] + (size_t) k * L + l] + smem_abs;  // (no carry into the flags: host check)
        }
    }
    mlw_pk = (uint32_t) __builtin_amdgcn_readfirstlane((int) mlw_pk);
    gen_pk = (uint32_t) __builtin_amdgcn_readfirstlane((int) gen_pk);
    T ylreg[BP][4], qreg[BP];
    // U is tiled 32 slots x 4 rows: my slot in pass p has row 3 at u3_0 + p * u3_step, row r 32 words before per row
    const uint32_t u3_0 = u_base + (uint32_t) ((l >> 5) * 128 + 96 + (l & 31)) * (uint32_t) sizeof(T);
    const uint32_t u3_step = (uint32_t) L * 4u * (uint32_t) sizeof(T);

    for (;;) {
        __syncthreads();
        if (l == 0) fr_lds = atomicAdd(a.work_counter, 1ull);  // dynamic frame hand-out
        __syncthreads();
        const int64_t frame = (int64_t) fr_lds;
        if (frame >= a.frames) break;
        // ---- start of a frame --------------------------------------------------------------------------------
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            T q = (T) 0;  // auxiliaries: q = 0 (qp_admm.h:24)
            if (p < t.n_vpass) {
                const int i = t.var_of_slot[p * L + l];
                if (i >= 0 && i < t.n) {
                    if (a.y_is_f64) q = (T) (2 * reinterpret_cast<const double *>(a.y)[(size_t) frame * t.n + i] / a.var);
                    else q = (T) (2 * (double) reinterpret_cast<const float *>(a.y)[(size_t) frame * t.n + i] / a.var);
                }
            }
            qreg[p] = q + (alpha / 2);  // CalculateCoef, algo/algo.h:13-20; the v-update starts from q_i + alpha/2 (qp_admm.h:133)
        }
        for (int w = l; w < t.V_pad; w += L) V[w] = (T) 0;
#pragma unroll
        for (int p = 0; p < BP; ++p) {
#pragma unroll
            for (int row = 0; row < 4; ++row) ylreg[p][row] = (T) 0;  // z = yl = 0 (qp_admm.h:120-121)
            if (p < t.n_gpass && p * L + l < t.U_slots) {  // (thread slots beyond the U array hold no group)
                T *pu = reinterpret_cast<T *>(smem + u3_0 + (uint32_t) p * u3_step) - 96;  // U[slot][row] = pu[32 * row]
                pu[0] = pu[32] = pu[64] = (T) 0 + mu * ((T) 0 - (T) 0);
                pu[96] = (((tys >> (2 * p)) & 3u) == 3u) ? (T) 0 + mu * ((T) 0 - (T) 2) : (T) 0 + mu * ((T) 0 - (T) 0);
            }
        }
        __syncthreads();
        // ---- sweeps (qp_admm.h:130-164) ------------------------------------------------------------------------
        int it = 0;
        while (it < a.max_iter) {
            // Constants the compiler must not see through: everything derived from the register-resident tables
            // (addresses, +-1 patterns) would otherwise be hoisted out of the sweep loop into ~5 registers per entry.
            uint32_t k80, k1 = 1, k2 = 2, lds0, one_hi, mlw_o = mlw_pk, gen_o = gen_pk;
            asm volatile("s_mov_b32 %0, 0x80000000\n\ts_mov_b32 %1, 0" : "=s"(k80), "=s"(lds0));
            if (!LEAN) asm volatile("s_mov_b32 %0, 1\n\ts_mov_b32 %1, 2" : "=s"(k1), "=s"(k2));  // (only the long-list path shifts by them)
            if (sizeof(T) == 8) asm volatile("v_mov_b32 %0, 0x3ff00000" : "=v"(one_hi));  // in a VGPR: (x & k80) | one_hi is one v_and_or_b32
            else asm volatile("v_mov_b32 %0, 1.0" : "=v"(one_hi));
            asm volatile("" : "+s"(mlw_o), "+s"(gen_o));
            uint32_t l_o = (uint32_t) l;  // (opaque too: one shift-add per pass instead of a hoisted, then spilled, address)
            asm volatile("" : "+v"(l_o));
            auto pm1 = [&](uint32_t x) -> T {  // bit 31 of x set -> -1, else +1
                const uint32_t hi = (x & k80) | one_hi;
                if constexpr (sizeof(T) == 8) return (T) __hiloint2double((int) hi, 0);
                else return (T) __uint_as_float(hi);
            };
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                const int ml = (int) ((mlw_o >> (8 * p)) & 0xFFu);
                if (ml > 0) {  // v-update (qp_admm.h:132-142); ml == 0: no variable of my wavefront here (or no such pass)
                    T B = qreg[p];
                    const uint32_t v_cell = cell_slot ? 0u : (uint32_t) t.blk_cell[(uint32_t) (p * L) + l_o];
                    {  // (entries past admm_vk(p) are never touched: the chain runs min(ml, admm_vk(p)) entries)
                        uint32_t ec[ADMM_VK];
#pragma unroll
                        for (int k = 0; k < ADMM_VK; ++k) ec[k] = ent[p][k < admm_vk(p) ? k : 0];
                        admm_v_chain(ec, (uint32_t) (ml < admm_vk(p) ? ml : admm_vk(p)), k80, one_hi, B);
                    }
                    if constexpr (!LEAN)
                    for (int k = admm_vk(p); k < ml; ++k) {  // lists longer than the register file holds
                        const uint32_t e = t.blk_list[(size_t) t.v_list_off[p] + (size_t) k * L + l];
                        T ux, uy, uz, uw;
                        admm_read_rows(smem, (e & 0xFFFFu) + lds0, ux, uy, uz, uw);
                        B = X::fma(pm1(e), ux, B);
                        B = X::fma(pm1(e << k1), uy, B);
                        B = X::fma(pm1(e << k2), uz, B);
                        B = B + uw;
                    }
                    // std::min(std::max(v, 0.0), 1.0) (qp_admm.h:140-141) as ONE instruction, the output clamp of a v_max: it can
                    // differ from the comparisons only in the sign of a zero, which no later value, comparison or decision
                    // depends on — and a NaN passes through it as through std::max / std::min, because this kernel runs with
                    // MODE.DX10_CLAMP cleared (set at its top; with the bit set the clamp would turn a NaN into 0)
                    const T v = X::clamp01(B * inv[p]);
                    if ((tys >> (8 + p)) & 1u) {
                        const uint32_t va = cell_slot ? l_o * (uint32_t) sizeof(T) + (lds0 + (uint32_t) (p * L) * (uint32_t) sizeof(T)) : v_cell * (uint32_t) sizeof(T);
                        *reinterpret_cast<T *>(smem + va) = v;
                    }
                }
            }
            __syncthreads();
            T sum2 = (T) 0;  // residual, multiplier and slack update (qp_admm.h:144-159)
#pragma unroll
            for (int p = 0; p < BP; ++p) {  // (a pass the problem does not have: ty == 0 in every lane)
                    const uint32_t ty = (tys >> (2 * p)) & 3u;
                    const uint32_t u3 = u3_0 + (uint32_t) p * u3_step;
                    asm volatile("" : "+v"(mem[p][0]), "+v"(mem[p][1]), "+v"(mem[p][2]));  // opaque in place: no copies
                    const uint32_t mo0 = mem[p][0], mo1 = mem[p][1], mo2 = mem[p][2];
                    if (!LEAN && ((gen_o >> p) & 1u)) {  // wavefront-uniform
                        if (ty != 0u) admm_group_update<T, EE, true>(smem, lds0, mo0, mo1, mo2, u3, ty, mu, ylreg[p], sum2);
                    } else {
                        if (ty != 0u) admm_group_update<T, EE, false>(smem, lds0, mo0, mo1, mo2, u3, ty, mu, ylreg[p], sum2);
                    }
                }
            it += 1;
            if (EE) {
                // every term of the residual is a square: as soon as one lane's partial sum reaches eps the total does, and the
                // wavefront can report "not below eps" (+inf) without the cross-lane fp64 sum (12 DPP moves, 4 adds, 8 lane
                // reads) — the case of every sweep but the last few of a frame.  The stopping decision is unchanged: inf (or
                // NaN) < eps is false exactly when the exact total >= eps (or NaN) is.
                if (__ballot(sum2 >= eps_stop) != 0ull) sum2 = (T) INFINITY;
                else sum2 = wave_sum(sum2);
                if (lane == 0) red[wave] = sum2;
                if (l < 4 && l >= (L >> 6)) red[l] = (T) 0;  // workgroups of fewer than 4 wavefronts
                __syncthreads();
                const T tot = ((red[0] + red[1]) + red[2]) + red[3];
                if (tot < eps_stop) break;  // qp_admm.h:161-163 (identical in every thread)
            } else {
                __syncthreads();
            }
        }
        // ---- outputs (qp_admm.h:166-177) -----------------------------------------------------------------------
        for (int w = l; w < t.nwords; w += L) OB[w] = 0u;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < BP; ++p)
            if (p < t.n_vpass) {  // every thread reports the variables it owns (V is indexed by cell, not by variable id)
                const int i = t.var_of_slot[p * L + l];
                if (i >= 0 && i < t.n) {
                    const uint32_t va = (cell_slot ? (uint32_t) (p * L + l) : (uint32_t) t.blk_cell[p * L + l]) * (uint32_t) sizeof(T);
                    if (!(*reinterpret_cast<const T *>(smem + va) <= (T) 0.5)) atomicOr(&OB[i >> 5], 1u << (i & 31));
                }
            }
        __syncthreads();
        if (a.out_bits)
            for (int w = l; w < t.nwords; w += L) a.out_bits[(size_t) frame * t.nwords + w] = OB[w];
        if (l == 0) {
            if (a.out_ok) a.out_ok[frame] = 1;
            if (a.out_iters) a.out_iters[frame] = it;
        }
    }
}

// guard path (qp_admm.h:112-114): all-zero word, ok = false, no sweeps
__global__ void admm_guard_kernel(DecodeArgs a, int nwords) {
    for (int64_t f = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; f < a.frames; f += (int64_t) gridDim.x * blockDim.x) {
        if (a.out_bits)
            for (int w = 0; w < nwords; ++w) a.out_bits[(size_t) f * nwords + w] = 0u;
        if (a.out_ok) a.out_ok[f] = 0;
        if (a.out_iters) a.out_iters[f] = 0;
    }
}

// sweep budget 0: the loop of qp_admm.h:130 never runs, so the word returned (qp_admm.h:166-175) is the initial guess
// v_i = (q_i > 0) of qp_admm.h:116-119 — a dead store for every other budget.  ok = true, no sweeps.  Dispatched by the
// host (max_iter is launch-uniform); the sweep kernels never see max_iter == 0.
template <typename T>
__global__ void admm_budget0_kernel(DecodeArgs a, int n, int nwords) {
    const int64_t total = a.frames * nwords;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t f = i / nwords;
        const int w = (int) (i - f * nwords);
        uint32_t word = 0;
        for (int b = 0; b < 32 && w * 32 + b < n; ++b) {
            const size_t idx = (size_t) f * n + (size_t) (w * 32 + b);
            const double yv = a.y_is_f64 ? reinterpret_cast<const double *>(a.y)[idx] : (double) reinterpret_cast<const float *>(a.y)[idx];
            const T q = (T) (2 * yv / a.var);  // CalculateCoef, algo/algo.h:13-20 (same expression as the sweep kernels)
            word |= (q > (T) 0 ? 1u : 0u) << b;
        }
        if (a.out_bits) a.out_bits[i] = word;
        if (w == 0) {
            if (a.out_ok) a.out_ok[f] = 1;  // qp_admm.h:165,177
            if (a.out_iters) a.out_iters[f] = 0;
        }
    }
}

constexpr int ADMM_NGP = 12;  // register-resident row state for codes with <= 12*64 constraint groups

template <typename T, int L, int NGP>
static const void *admm_ptr(bool mc) {
    return mc ? (const void *) admm_fused_kernel<T, L, true, NGP> : (const void *) admm_fused_kernel<T, L, false, NGP>;
}

static const void *admm_kernel_ptr(int f32, int L, bool mc, bool reg) {
    if (f32) {
        if (L == 64) return reg ? admm_ptr<float, 64, ADMM_NGP>(mc) : admm_ptr<float, 64, 0>(mc);
        if (L == 32) return admm_ptr<float, 32, 0>(mc);
        return admm_ptr<float, 16, 0>(mc);
    }
    if (L == 64) return reg ? admm_ptr<double, 64, ADMM_NGP>(mc) : admm_ptr<double, 64, 0>(mc);
    if (L == 32) return admm_ptr<double, 32, 0>(mc);
    return admm_ptr<double, 16, 0>(mc);
}

template <typename T, bool EE, bool LEAN>
static const void *admm_block_ptr_t(int passes) {
    if (passes <= 2) return (const void *) admm_block_kernel<T, EE, 2, LEAN>;
    if (passes == 3) return (const void *) admm_block_kernel<T, EE, 3, LEAN>;
    return (const void *) admm_block_kernel<T, EE, 4, LEAN>;
}

static const void *admm_block_ptr(int f32, bool ee, int passes, bool lean) {
    if (lean) {
        if (f32) return ee ? admm_block_ptr_t<float, true, true>(passes) : admm_block_ptr_t<float, false, true>(passes);
        return ee ? admm_block_ptr_t<double, true, true>(passes) : admm_block_ptr_t<double, false, true>(passes);
    }
    if (f32) return ee ? admm_block_ptr_t<float, true, false>(passes) : admm_block_ptr_t<float, false, false>(passes);
    return ee ? admm_block_ptr_t<double, true, false>(passes) : admm_block_ptr_t<double, false, false>(passes);
}

template <typename T>
static void *upload_vec(const std::vector<T> &h, std::vector<void *> &allocs, std::string &err) {
    void *d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    if (hipMalloc(&d, bytes) != hipSuccess) {
        err = "hipMalloc failed";
        return nullptr;
    }
    if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        err = "hipMemcpy failed";
        return nullptr;
    }
    allocs.push_back(d);
    return d;
}

AdmmDevice *admm_device_create(const Code &c, const acg_ldpc_params &p, int cu_count, std::string &err) {
    const AdmmLayout &A = c.admm;
    auto *d = new AdmmDevice();
    d->alpha = p.alpha;
    d->mu = p.mu;
    d->eps = p.eps_stop;
    if (p.precision == ACG_LDPC_PREC_F16) {
        err = "ACG_LDPC_PREC_F16 exists for the fused min-sum decoder only";
        delete d;
        return nullptr;
    }
    d->f32 = (p.precision == ACG_LDPC_PREC_F32) ? 1 : 0;
    int L = p.lanes_per_frame ? p.lanes_per_frame : 64;
    // auto / 256: one workgroup (128, 192 or 256 threads) per frame when the problem has at most 4 passes of it.
    // Choice: most workgroups resident per CU (LDS: 160 KiB; registers: 8 - passes workgroups of 4 wavefronts, see
    // admm_block_kernel) per pass of serial work; ties go to the larger workgroup.
    const bool can_block = (A.n_grp + 1 <= ADMM_BP * ADMM_BLK) && (A.n_var <= ADMM_BP * ADMM_BLK);
    if ((p.lanes_per_frame == 0 || p.lanes_per_frame == ADMM_BLK) && can_block) {
        int bestL = 0;
        double best_score = -1;
        const char *force = getenv("ACG_ADMM_BLOCK_L");  // developer A/B only
        for (int cand : {256, 192, 128}) {
            const int gp = (A.n_grp + 1 + cand - 1) / cand, vp = (A.n_var + cand - 1) / cand;
            const int passes = std::max(std::max(gp, vp), 2);
            if (passes > ADMM_BP) continue;
            const size_t ts_ = (p.precision == ACG_LDPC_PREC_F32) ? 4 : 8;
            const size_t lds = (size_t) (4 * gp * cand + vp * cand + 4) * ts_ + 64;
            const int by_lds = (int) ((160 * 1024) / lds);
            const int by_reg = ((ADMM_OCC - passes + (ts_ == 4 ? ADMM_OCC_F32 : 0)) * 4) / (cand / 64);
            double score = (double) std::min(by_lds, by_reg) / passes;
            if (force && atoi(force) == cand) score = 1e9;
            if (score > best_score) {
                best_score = score;
                bestL = cand;
            }
        }
        L = bestL;
        d->blockmode = true;
    } else if (p.lanes_per_frame == ADMM_BLK) {
        err = "lanes_per_frame = 256 needs at most 1024 constraint groups and variables";
        delete d;
        return nullptr;
    }
    if (!d->blockmode && L != 16 && L != 32 && L != 64) {
        err = "lanes_per_frame must be 0, 16, 32, 64 (or 256 for QP-ADMM)";
        delete d;
        return nullptr;
    }
    d->L = L;
    // guard: double e_min = 1e9; min over e (qp_admm.h:108-111)
    double e_min = 1e9;
    for (double e : A.e) e_min = std::min(e_min, e);
    d->guard = (e_min * p.mu <= p.alpha);
    d->budget0 = (p.max_iter == 0);

    AdmmDevTables &t = d->t;
    t.n = c.n;
    t.m = c.m;
    t.n_var = A.n_var;
    t.n_grp = A.n_grp;
    t.nwords = (c.n + 31) / 32;
    t.n_gpass = (A.n_grp + 1 + L - 1) / L;  // +1: at least one padding slot that stays all-zero
    t.G_pad = t.n_gpass * L;
    t.U_slots = t.G_pad;
    t.zero_gslot = A.n_grp;
    if (t.G_pad >= (1 << 20) || A.n_var >= (1 << 24)) {
        err = "code too large for the fused QP-ADMM kernel";
        delete d;
        return nullptr;
    }
    t.n_vpass = (A.n_var + L - 1) / L;
    t.V_pad = (A.n_var + 1 + 3) & ~3;

    std::vector<uint32_t> grp_mem((size_t) 3 * t.G_pad, 0xFFFFFFFFu);
    std::vector<uint8_t> grp_type((size_t) t.G_pad, 0);
    for (int g = 0; g < A.n_grp; g++) {
        const int ty = A.grp_type[g];
        grp_type[g] = (uint8_t) ty;
        int order[3] = {0, 1, 2};
        std::sort(order, order + ty, [&](int x, int y) { return A.grp_var[(size_t) g * 3 + x] < A.grp_var[(size_t) g * 3 + y]; });
        for (int k = 0; k < ty; k++) {
            const int wpos = order[k];
            grp_mem[(size_t) k * t.G_pad + g] = (uint32_t) A.grp_var[(size_t) g * 3 + wpos] | ((uint32_t) wpos << 24);
        }
    }
    // thread slot -> variable.  Wavefront kernels: variables sorted by list length (descending) so a pass has a uniform
    // trip count.  Workgroup-per-frame kernel: the three maps (variable -> thread slot, group -> U slot, variable ->
    // V cell) come from admm_block_placement (code.cpp), which places them against LDS bank conflicts.
    auto llen = [&](int i) { return A.var_ptr[i + 1] - A.var_ptr[i]; };
    AdmmBlockPlacement P;
    std::vector<int32_t> var_of_slot((size_t) t.n_vpass * L, -1), v_maxlist(t.n_vpass), v_list_off(t.n_vpass);
    if (d->blockmode) {
        // ACG_ADMM_NO_PLACEMENT / ACG_ADMM_NO_QC: developer A/B only
        const int mode = (getenv("ACG_ADMM_NO_PLACEMENT") || p.fast_setup) ? 0 : (getenv("ACG_ADMM_NO_QC") ? 1 : 2);
        admm_block_placement(c, L, d->f32 != 0, mode, P);
        if (getenv("ACG_ADMM_PLACEMENT_DEBUG"))
            fprintf(stderr, "[acg_ldpc] QP-ADMM placement (%s%s): modelled LDS cycles per frame-sweep: U-row reads %ld (conflict-free %ld), "
                            "V reads %ld (%ld), V writes %ld (%ld); v-update trips per wavefront %d %d %d %d\n",
                    P.qc ? "quasi-cyclic tuples, Z = " : (mode ? "annealed" : "none"), P.qc ? std::to_string(P.Z).c_str() : "", P.cyc_u_reads,
                    P.ideal_u_reads, P.cyc_v_reads, P.ideal_v_reads, P.cyc_v_writes, P.ideal_v_writes, P.wave_cost[0], P.wave_cost[1],
                    P.wave_cost[2], P.wave_cost[3]);
        for (size_t sidx = 0; sidx < var_of_slot.size(); sidx++) var_of_slot[sidx] = P.var_of_slot[sidx];
        t.zero_gslot = P.zero_gslot;
        t.V_pad = (P.n_cells + 3) & ~3;
        t.U_slots = P.u_slots;
    } else {
        std::vector<int> vorder(A.n_var);
        for (int i = 0; i < A.n_var; i++) vorder[i] = i;
        std::stable_sort(vorder.begin(), vorder.end(), [&](int x, int y) { return llen(x) > llen(y); });
        for (int sidx = 0; sidx < A.n_var; sidx++) var_of_slot[sidx] = vorder[sidx];
    }
    int off = 0;
    for (int p_ = 0; p_ < t.n_vpass; p_++) {
        int ml = 0;
        for (int l = 0; l < L; l++)
            if (var_of_slot[(size_t) p_ * L + l] >= 0) ml = std::max(ml, llen(var_of_slot[(size_t) p_ * L + l]));
        v_maxlist[p_] = ml;
        v_list_off[p_] = off;
        off += ml * L;
    }
    std::vector<uint32_t> v_list((size_t) std::max(off, 1), (uint32_t) A.n_grp);  // padding: slot n_grp of the wavefront kernels, type 0, wpos 0 -> adds 0
    std::vector<double> inv64((size_t) t.n_vpass * L, 0.0);
    for (int sidx = 0; sidx < t.n_vpass * L; sidx++) {
        const int i = var_of_slot[sidx];
        if (i < 0) continue;
        const int p_ = sidx / L, l = sidx % L;
        for (int k = 0; k < llen(i); k++) {
            const int ent = A.var_grp[A.var_ptr[i] + k];
            const int g = ent >> 2, wpos = ent & 3;
            v_list[(size_t) v_list_off[p_] + (size_t) k * L + l] =
                (uint32_t) g | ((uint32_t) wpos << 20) | ((uint32_t) A.grp_type[g] << 22);
        }
        const double Acoef = (p.mu * A.e[i] - p.alpha) / 2;  // qp_admm.h:125
        inv64[sidx] = -1.0 / (2 * Acoef);                     // qp_admm.h:126
    }
    bool ok = true;
    if (d->blockmode) {
        const uint32_t ts = d->f32 ? 4 : 8;
        const uint32_t u_base = (uint32_t) t.V_pad * ts;
        // U is tiled: 32 slots x 4 rows per tile, row-major inside a tile, so the four rows of a slot are a fixed
        // 32-word stride apart (immediate offsets) and the bank of every access is slot mod 32
        auto u_addr = [&](int gs, int row) {
            return u_base + (((uint32_t) gs >> 5) * 128u + (uint32_t) row * 32u + ((uint32_t) gs & 31u)) * ts;
        };
        if (u_addr(t.U_slots, 0) + (uint32_t) t.nwords * 4 + 1024u > 0xFFFFu) {  // (+1 KiB: the kernel's static LDS words sit in front and are added to the 16-bit list addresses)
            err = "QP-ADMM frame state exceeds the 64 KiB the workgroup-per-frame kernel addresses";
            admm_device_destroy(d);
            return nullptr;
        }
        const std::vector<int> &slot_of = P.slot_of_grp, &cell_of = P.cell_of_var;
        std::vector<int> grp_of(t.G_pad, -1);
        for (int g = 0; g < A.n_grp; g++) grp_of[slot_of[g]] = g;
        std::vector<uint32_t> blk_mem((size_t) 3 * t.G_pad, 0);
        std::vector<uint8_t> type_slot((size_t) t.G_pad, 0), blk_generic((size_t) t.n_gpass * 4, 0);
        for (int sl = 0; sl < t.G_pad; sl++) {
            const int g = grp_of[sl];
            const int ty = g >= 0 ? A.grp_type[g] : 0;
            type_slot[sl] = (uint8_t) ty;
            if (ty == 1 || ty == 2) blk_generic[(size_t) (sl / L) * 4 + (sl % L) / 64] = 1;
            bool row_used[4] = {false, false, false, false};
            for (int k = 0; k < ty; k++) row_used[grp_mem[(size_t) k * t.G_pad + g] >> 24] = true;
            for (int k = 0; k < 3; k++) {
                uint32_t cell = (uint32_t) P.zero_cell, row = 0;  // absent member: the zero cell ...
                if (k < ty) {
                    const uint32_t e = grp_mem[(size_t) k * t.G_pad + g];
                    cell = (uint32_t) cell_of[e & 0xFFFFFFu];
                    row = e >> 24;
                } else {  // ... and a row of this group that no member owns (it stays zero)
                    while (row_used[row]) row++;
                    row_used[row] = true;
                }
                blk_mem[(size_t) k * t.G_pad + sl] = (cell * ts) | (u_addr(sl, (int) row) << 16);
            }
        }
        std::vector<uint32_t> blk_list(v_list.size(), u_addr(t.zero_gslot, 0));  // padding: an all-zero slot, coefficients +1
        std::vector<int32_t> blk_mlw((size_t) t.n_vpass * 4, 0), blk_cell((size_t) t.n_vpass * L, -1);
        bool list_ok = true;
        for (int sidx = 0; sidx < t.n_vpass * L; sidx++) {
            const int i = var_of_slot[sidx];
            const int p_ = sidx / L, l = sidx % L;
            const int len = i >= 0 ? llen(i) : 0;
            if (i >= 0) {
                blk_cell[sidx] = cell_of[i];
                blk_mlw[(size_t) p_ * 4 + l / 64] = std::max(blk_mlw[(size_t) p_ * 4 + l / 64], len);
                list_ok = list_ok && len <= 255;
            }
            for (int k = 0; k < v_maxlist[p_]; k++) {
                uint32_t &dst = blk_list[(size_t) v_list_off[p_] + (size_t) k * L + l];
                if (k < len) {
                    const int ent = A.var_grp[A.var_ptr[i] + k];
                    const int g = ent >> 2, wpos = ent & 3, ty = A.grp_type[g];
                    uint32_t flags = 0;
                    for (int row = 0; row < 4; row++) {
                        const bool plus = (ty == 3 && row == 3) || (row == wpos);
                        if (!plus) flags |= 0x80000000u >> row;
                    }
                    dst = u_addr(slot_of[g], 0) | flags;
                } else if (k < P.max_list) {
                    const int pad = P.pad_gslot[(size_t) sidx * P.max_list + k];  // an all-zero slot in a bank nobody else reads
                    dst = u_addr(pad >= 0 ? pad : t.zero_gslot, 0);
                }
            }
        }
        if (!list_ok) {
            err = "a variable in more than 255 checks: use lanes_per_frame 16/32/64 for QP-ADMM";
            admm_device_destroy(d);
            return nullptr;
        }
        t.blk_mem = (const uint32_t *) upload_vec(blk_mem, d->allocs, err);
        t.blk_list = (const uint32_t *) upload_vec(blk_list, d->allocs, err);
        t.blk_mlw = (const int32_t *) upload_vec(blk_mlw, d->allocs, err);
        t.grp_type_slot = (const uint8_t *) upload_vec(type_slot, d->allocs, err);
        t.blk_generic = (const uint8_t *) upload_vec(blk_generic, d->allocs, err);
        t.blk_cell = (const int32_t *) upload_vec(blk_cell, d->allocs, err);
        t.cell_is_slot = 1;
        for (int sidx = 0; sidx < t.n_vpass * L; sidx++)
            if (blk_cell[sidx] >= 0 && blk_cell[sidx] != sidx) t.cell_is_slot = 0;
        // the lean kernel instance (admm_block_kernel<..., LEAN>): nothing of the general paths is needed inside a sweep
        d->blk_lean = t.cell_is_slot && !getenv("ACG_ADMM_NO_LEAN");
        for (uint8_t g_ : blk_generic) d->blk_lean = d->blk_lean && g_ == 0;
        for (int p_ = 0; p_ < t.n_vpass; p_++)
            for (int w_ = 0; w_ < 4; w_++) d->blk_lean = d->blk_lean && blk_mlw[(size_t) p_ * 4 + w_] <= admm_vk(p_);
        ok = t.blk_mem && t.blk_list && t.blk_mlw && t.grp_type_slot && t.blk_generic && t.blk_cell;
    }
    t.grp_mem = (const uint32_t *) upload_vec(grp_mem, d->allocs, err);
    t.grp_type = (const uint8_t *) upload_vec(grp_type, d->allocs, err);
    t.var_of_slot = (const int32_t *) upload_vec(var_of_slot, d->allocs, err);
    t.v_maxlist = (const int32_t *) upload_vec(v_maxlist, d->allocs, err);
    t.v_list_off = (const int32_t *) upload_vec(v_list_off, d->allocs, err);
    t.v_list = (const uint32_t *) upload_vec(v_list, d->allocs, err);
    if (d->f32) {
        std::vector<float> inv32(inv64.begin(), inv64.end());
        t.inv_coef = upload_vec(inv32, d->allocs, err);
    } else {
        t.inv_coef = upload_vec(inv64, d->allocs, err);
    }
    t.row_ptr = (const int32_t *) upload_vec(c.row_ptr, d->allocs, err);
    t.edge_var = (const int32_t *) upload_vec(c.edge_var, d->allocs, err);
    ok = ok && t.grp_mem && t.grp_type && t.var_of_slot && t.v_maxlist && t.v_list_off && t.v_list && t.inv_coef && t.row_ptr && t.edge_var;
    if (!ok) {
        admm_device_destroy(d);
        return nullptr;
    }
    const size_t ts = d->f32 ? 4 : 8;
    size_t per_frame = (size_t) (4 * (d->blockmode ? t.U_slots : t.G_pad) + t.V_pad + (d->blockmode ? 0 : t.n_vpass * L)) * ts + (size_t) t.nwords * 4;
    per_frame = (per_frame + 15) & ~(size_t) 15;
    t.lds_bytes_per_frame = (int) per_frame;
    if (d->blockmode) {
        d->block = L;
        d->frames_per_block = 1;
        d->lds_block = per_frame;
        if (per_frame > 160 * 1024) {
            err = "QP-ADMM frame state does not fit in LDS (160 KiB per CU)";
            admm_device_destroy(d);
            return nullptr;
        }
        int per_cu = 0;
        for (int ee = 0; ee < 2; ee++) {  // kernel[0]: fixed sweep count, kernel[1]: with the residual stopping rule
            const int mc = ee;
            const int passes = std::max(std::max(t.n_gpass, t.n_vpass), 2);
            const void *kp = admm_block_ptr(d->f32, ee != 0, passes, d->blk_lean);
            d->kernel[mc] = kp;
            if (d->lds_block > 64 * 1024 &&
                hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) d->lds_block) != hipSuccess) {
                err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
                admm_device_destroy(d);
                return nullptr;
            }
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kp, d->block, d->lds_block) != hipSuccess) occ = 1;
            if (occ < 1) occ = 1;
            per_cu = (mc == 0) ? occ : std::max(per_cu, occ);  // frames are handed out dynamically: a generous grid is safe
        }
        d->grid_cap = per_cu * cu_count;
        return d;
    }
    const int fpw = 64 / L;
    // wavefronts per workgroup: whatever packs the most frames into the 160 KiB of a CU
    int waves = 1, best = 0;
    for (int w = 4; w >= 1; w >>= 1) {
        const size_t blk = per_frame * fpw * w;
        const int frames_cu = blk <= 160 * 1024 ? (int) ((160 * 1024) / blk) * w * fpw : 0;
        if (frames_cu > best) {
            best = frames_cu;
            waves = w;
        }
    }
    if (per_frame * fpw * waves > 160 * 1024) {
        err = "QP-ADMM frame state does not fit in LDS (160 KiB per CU)";
        admm_device_destroy(d);
        return nullptr;
    }
    d->block = waves * 64;
    d->frames_per_block = waves * fpw;
    d->lds_block = per_frame * fpw * waves;
    d->reg = (L == 64 && t.n_gpass <= ADMM_NGP);
    int per_cu = 0;
    for (int mc = 0; mc < 2; mc++) {
        const void *kp = admm_kernel_ptr(d->f32, L, mc != 0, d->reg);
        d->kernel[mc] = kp;
        if (d->lds_block > 64 * 1024 &&
            hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int) d->lds_block) != hipSuccess) {
            err = "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
            admm_device_destroy(d);
            return nullptr;
        }
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kp, d->block, d->lds_block) != hipSuccess) occ = 1;
        if (occ < 1) occ = 1;
        per_cu = (mc == 0) ? occ : std::min(per_cu, occ);
    }
    d->grid_cap = per_cu * cu_count;
    return d;
}

void admm_device_destroy(AdmmDevice *d) {
    if (!d) return;
    for (void *p : d->allocs) (void) hipFree(p);
    delete d;
}

// true when Monte-Carlo runs should go AWGN kernel -> decode -> classify kernel instead of the fused MC kernel
bool admm_device_unfused_mc(const AdmmDevice *d, const int32_t **row_ptr, const int32_t **edge_var) {
    if (row_ptr) *row_ptr = d->t.row_ptr;
    if (edge_var) *edge_var = d->t.edge_var;
    return (d->blockmode || d->budget0) && !d->guard;
}

void admm_device_layout(const AdmmDevice *d, int *lds_per_frame, int *lanes, int *frames_per_block, int *grid) {
    if (lds_per_frame) *lds_per_frame = d->t.lds_bytes_per_frame;
    if (lanes) *lanes = d->L;
    if (frames_per_block) *frames_per_block = d->frames_per_block;
    if (grid) *grid = d->grid_cap;
}

template <typename T>
static hipError_t admm_launch_t(AdmmDevice *d, const DecodeArgs &a, int grid, hipStream_t s) {
    AdmmDevTables tt = d->t;
    DecodeArgs aa = a;
    T alpha = (T) d->alpha, mu = (T) d->mu, eps = (T) d->eps;
    void *args[5] = {&tt, &aa, &alpha, &mu, &eps};
    // eps_stop <= 0: the residual (a sum of squares) is never below it, so the instance without the residual is exact
    const int which = d->blockmode ? ((a.early_exit && d->eps > 0) ? 1 : 0) : (a.mc ? 1 : 0);
    return hipLaunchKernel(d->kernel[which], dim3(grid), dim3(d->block), args, d->lds_block, s);
}

hipError_t admm_launch(AdmmDevice *d, const DecodeArgs &a, hipStream_t s, std::string &err) {
    if (d->guard) {
        if (a.mc) {
            err = "QP-ADMM guard e_min*mu <= alpha fires: every frame fails (qp_admm.h:112-114); Monte-Carlo run refused";
            return hipErrorInvalidValue;
        }
        int grid = (int) std::min<int64_t>((a.frames + 255) / 256, 4096);
        hipLaunchKernelGGL(admm_guard_kernel, dim3(grid), dim3(256), 0, s, a, d->t.nwords);
        return hipGetLastError();
    }
    if (a.max_iter == 0) {
        if (a.mc) {
            err = "internal: QP-ADMM with a sweep budget of 0 has no fused Monte-Carlo mode";
            return hipErrorInvalidValue;
        }
        const int64_t total = a.frames * d->t.nwords;
        int grid = (int) std::min<int64_t>((total + 255) / 256, 8192);
        if (d->f32) hipLaunchKernelGGL(admm_budget0_kernel<float>, dim3(grid), dim3(256), 0, s, a, d->t.n, d->t.nwords);
        else hipLaunchKernelGGL(admm_budget0_kernel<double>, dim3(grid), dim3(256), 0, s, a, d->t.n, d->t.nwords);
        return hipGetLastError();
    }
    if (d->blockmode && a.mc) {
        err = "the workgroup-per-frame QP-ADMM kernel has no fused Monte-Carlo mode (use AWGN kernel -> decode -> classify)";
        return hipErrorInvalidValue;
    }
    int64_t blocks = (a.frames + d->frames_per_block - 1) / d->frames_per_block;
    int grid = (int) std::min<int64_t>(blocks, d->grid_cap);
    const hipError_t e = d->f32 ? admm_launch_t<float>(d, a, grid, s) : admm_launch_t<double>(d, a, grid, s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace acg
