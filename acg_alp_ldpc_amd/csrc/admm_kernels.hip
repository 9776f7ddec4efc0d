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
    // syndrome (MC classification)
    const int32_t *row_ptr;
    const int32_t *edge_var;
    int32_t n, m, n_var, n_grp, n_gpass, n_vpass, G_pad, V_pad, zero_gslot, nwords;
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
    const void *kernel[2] = {nullptr, nullptr};
    size_t lds_block = 0;
    bool guard = false;  // e_min*mu <= alpha  (qp_admm.h:108-114)
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
        sum2 = group_sum<L, T>(sum2);
        it += 1;
        converged = (sum2 < eps_stop);
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
// Workgroup-per-frame variant: the 256 threads of a workgroup own ONE frame (requires <= 4*256 constraint groups
// and variables).  Same arithmetic, same order of every rounding step as the wavefront variant; the LDS footprint
// per frame is unchanged but four times as many wavefronts work on it, which is what this latency-bound sweep
// needs (one wavefront per frame left 5 waves per CU: 1.25 per SIMD).  Row state w and the channel terms q live in
// registers (4 passes at most), LDS holds u_j = yl_j + mu*(z_j - b_j) and v.
constexpr int ADMM_BLK = 256;
constexpr int ADMM_BP = 4;

template <typename T, bool MC>
__global__ void __launch_bounds__(ADMM_BLK) admm_block_kernel(const AdmmDevTables t, const DecodeArgs a, const T alpha,
                                                              const T mu, const T eps_stop) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = blockDim.x;  // 128, 192 or 256 threads = one frame
    __shared__ T red[4];
    __shared__ unsigned long long fr_lds;
    __shared__ int flag_lds[2];
    __shared__ int ham_lds;
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    T *U = reinterpret_cast<T *>(smem);  // [4][G_pad]
    T *V = U + 4 * t.G_pad;              // [V_pad] by variable id (+ zero cell at n_var)
    uint32_t *OB = reinterpret_cast<uint32_t *>(V + t.V_pad);
    const T *inv_coef = reinterpret_cast<const T *>(t.inv_coef);
    unsigned long long acc_correct = 0, acc_pseudo = 0, acc_total = 0, acc_ham = 0, acc_ham_ok = 0, acc_ham_wrong = 0, acc_iters = 0;
    T wreg[ADMM_BP][4], qreg[ADMM_BP];

    for (;;) {
        __syncthreads();
        if (l == 0) {
            fr_lds = atomicAdd(a.work_counter, 1ull);  // dynamic frame hand-out
            ham_lds = 0;
            flag_lds[0] = 0;
            flag_lds[1] = 0;
        }
        __syncthreads();
        const int64_t frame = (int64_t) fr_lds;
        if (frame >= a.frames) break;
        const int64_t gf = a.first_frame + frame;
        const uint32_t *cw = (MC && a.cw_packed) ? a.cw_packed + (size_t) (gf % a.n_cw) * t.nwords : nullptr;
        // ---- start of a frame --------------------------------------------------------------------------------
        if (MC) {
            const int nq = (t.n + 3) >> 2;
            for (int q = l; q < nq; q += L) {
                uint32_t r[4];
                philox((uint32_t) gf, (uint32_t) (gf >> 32), (uint32_t) q, 0u, (uint32_t) a.seed, (uint32_t) (a.seed >> 32), r);
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
            __syncthreads();
        }
        int my_ham = 0;
#pragma unroll
        for (int p = 0; p < ADMM_BP; ++p) {
            qreg[p] = (T) 0;
            if (p < t.n_vpass) {
                const int i = t.var_of_slot[p * L + l];
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
                qreg[p] = q;  // CalculateCoef, algo/algo.h:13-20
            }
        }
        if (MC) {
            my_ham = group_sum<64, int>(my_ham);
            if (lane == 0) atomicAdd(&ham_lds, my_ham);
        }
        __syncthreads();  // staged symbols consumed
        for (int w = l; w < t.V_pad; w += L) V[w] = (T) 0;
#pragma unroll
        for (int p = 0; p < ADMM_BP; ++p)
            if (p < t.n_gpass) {
                const int gs = p * L + l;
                const int ty = t.grp_type[gs];
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    wreg[p][row] = (T) 0;  // z = yl = 0 (qp_admm.h:120-121)
                    const T b = (ty == 3 && row == 3) ? (T) 2 : (T) 0;
                    U[row * t.G_pad + gs] = (T) 0 + mu * ((T) 0 - b);
                }
            }
        __syncthreads();
        // ---- sweeps (qp_admm.h:130-164) ------------------------------------------------------------------------
        int it = 0;
        while (it < a.max_iter) {
#pragma unroll
            for (int p = 0; p < ADMM_BP; ++p)
                if (p < t.n_vpass) {  // v-update (qp_admm.h:132-142)
                    const int slot = p * L + l;
                    const int ml = t.v_maxlist[p];
                    const uint32_t *lp = t.v_list + t.v_list_off[p] + l;
                    T B = qreg[p] + (alpha / 2);
                    for (int k = 0; k < ml; ++k) {
                        const uint32_t ent = lp[(size_t) k * L];
                        const int gs = (int) (ent & 0xFFFFFu);
                        const int wp = (int) ((ent >> 20) & 3u);
                        const int ty = (int) (ent >> 22);
#pragma unroll
                        for (int row = 0; row < 4; ++row) {
                            const bool plus = (ty == 3 && row == 3) || (row == wp);
                            B = __builtin_fma(plus ? (T) 1 : (T) -1, U[row * t.G_pad + gs], B);
                        }
                    }
                    T v = B * inv_coef[slot];
                    v = (v < (T) 0) ? (T) 0 : v;  // std::max(v, 0.0)
                    v = ((T) 1 < v) ? (T) 1 : v;  // std::min(v, 1.0)
                    const int i = t.var_of_slot[slot];
                    if (i >= 0) V[i] = v;
                }
            __syncthreads();
            T sum2 = (T) 0;  // residual, multiplier and slack update (qp_admm.h:144-159)
#pragma unroll
            for (int p = 0; p < ADMM_BP; ++p)
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
                        const int id = have[k] ? (int) (e & 0xFFFFFFu) : t.n_var;
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
                        if (row < rows) {
                            wreg[p][row] = wn;
                            const T nwn = -wn;
                            const T yln = ((T) 0 < nwn) ? nwn : (T) 0;
                            const T bb = (ty == 3 && row == 3) ? (T) 2 : (T) 0;
                            U[row * t.G_pad + gs] = yln + mu * (z - bb);
                            const T d = z - r;
                            sum2 += d * d;
                        }
                    }
                }
            sum2 = group_sum<64, T>(sum2);
            if (lane == 0) red[wave] = sum2;
            if (l < 4 && l >= (L >> 6)) red[l] = (T) 0;  // workgroups of fewer than 4 wavefronts
            __syncthreads();
            const T tot = ((red[0] + red[1]) + red[2]) + red[3];
            it += 1;
            if (a.early_exit && tot < eps_stop) break;  // qp_admm.h:161-163 (identical in every thread)
        }
        // ---- outputs (qp_admm.h:166-177) -----------------------------------------------------------------------
        for (int w = l; w < t.nwords; w += L) OB[w] = 0u;
        __syncthreads();
        for (int v = l; v < t.n; v += L)
            if (!(V[v] <= (T) 0.5)) atomicOr(&OB[v >> 5], 1u << (v & 31));
        __syncthreads();
        if (a.out_bits)
            for (int w = l; w < t.nwords; w += L) a.out_bits[(size_t) frame * t.nwords + w] = OB[w];
        if (l == 0) {
            if (a.out_ok) a.out_ok[frame] = 1;
            if (a.out_iters) a.out_iters[frame] = it;
        }
        if (MC) {
            for (int c = l; c < t.m; c += L) {  // IsCodeword (experiment.h:111)
                uint32_t sy = 0;
                for (int e = t.row_ptr[c]; e < t.row_ptr[c + 1]; ++e) {
                    const int v = t.edge_var[e];
                    sy ^= (OB[v >> 5] >> (v & 31)) & 1u;
                }
                if (sy) flag_lds[0] = 1;
            }
            for (int w = l; w < t.nwords; w += L)
                if (OB[w] != (cw ? cw[w] : 0u)) flag_lds[1] = 1;
            __syncthreads();
            if (l == 0) {
                const bool is_cw = flag_lds[0] == 0, differ = flag_lds[1] != 0;
                const bool correct = is_cw && !differ;
                const int ham = ham_lds;
                acc_correct += correct;
                acc_pseudo += (is_cw && differ);
                acc_total += 1;
                acc_ham += ham;
                acc_ham_ok += correct ? ham : 0;
                acc_ham_wrong += correct ? 0 : ham;
                acc_iters += it;
            }
        }
    }
    if (MC && l == 0 && acc_total) {
        atomicAdd(&a.counters[MC_CORRECT], acc_correct);
        atomicAdd(&a.counters[MC_PSEUDO], acc_pseudo);
        atomicAdd(&a.counters[MC_TOTAL], acc_total);
        atomicAdd(&a.counters[MC_HAM], acc_ham);
        atomicAdd(&a.counters[MC_HAM_OK], acc_ham_ok);
        atomicAdd(&a.counters[MC_HAM_WRONG], acc_ham_wrong);
        atomicAdd(&a.counters[MC_ITERS], acc_iters);
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
    d->f32 = (p.precision == ACG_LDPC_PREC_F32) ? 1 : 0;
    int L = p.lanes_per_frame ? p.lanes_per_frame : 64;
    // auto / 256: one workgroup (128, 192 or 256 threads) per frame when the problem has at most 4 passes of it;
    // the size with the fewest padded group slots wins (LDS per frame = 4 words per slot), ties go to the larger
    const bool can_block = (A.n_grp + 1 <= ADMM_BP * ADMM_BLK) && (A.n_var <= ADMM_BP * ADMM_BLK);
    if ((p.lanes_per_frame == 0 || p.lanes_per_frame == ADMM_BLK) && can_block) {
        int bestL = ADMM_BLK, best_pad = ((A.n_grp + 1 + ADMM_BLK - 1) / ADMM_BLK) * ADMM_BLK;
        for (int cand : {192, 128}) {
            const int gp = (A.n_grp + 1 + cand - 1) / cand, vp = (A.n_var + cand - 1) / cand;
            if (gp <= ADMM_BP && vp <= ADMM_BP && gp * cand < best_pad) {
                best_pad = gp * cand;
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

    AdmmDevTables &t = d->t;
    t.n = c.n;
    t.m = c.m;
    t.n_var = A.n_var;
    t.n_grp = A.n_grp;
    t.nwords = (c.n + 31) / 32;
    t.n_gpass = (A.n_grp + 1 + L - 1) / L;  // +1: at least one padding slot that stays all-zero
    t.G_pad = t.n_gpass * L;
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
    // variables sorted by list length (descending) so a pass has a uniform trip count
    std::vector<int> vorder(A.n_var);
    for (int i = 0; i < A.n_var; i++) vorder[i] = i;
    auto llen = [&](int i) { return A.var_ptr[i + 1] - A.var_ptr[i]; };
    std::stable_sort(vorder.begin(), vorder.end(), [&](int x, int y) { return llen(x) > llen(y); });
    std::vector<int32_t> var_of_slot((size_t) t.n_vpass * L, -1), v_maxlist(t.n_vpass), v_list_off(t.n_vpass);
    int off = 0;
    for (int p_ = 0; p_ < t.n_vpass; p_++) {
        v_maxlist[p_] = llen(vorder[(size_t) p_ * L]);
        v_list_off[p_] = off;
        off += v_maxlist[p_] * L;
    }
    std::vector<uint32_t> v_list((size_t) std::max(off, 1), (uint32_t) t.zero_gslot);  // type 0, wpos 0 -> adds 0
    std::vector<double> inv64((size_t) t.n_vpass * L, 0.0);
    for (int s = 0; s < A.n_var; s++) {
        const int i = vorder[s];
        const int p_ = s / L, l = s % L;
        var_of_slot[s] = i;
        for (int k = 0; k < llen(i); k++) {
            const int ent = A.var_grp[A.var_ptr[i] + k];
            const int g = ent >> 2, wpos = ent & 3;
            v_list[(size_t) v_list_off[p_] + (size_t) k * L + l] =
                (uint32_t) g | ((uint32_t) wpos << 20) | ((uint32_t) A.grp_type[g] << 22);
        }
        const double Acoef = (p.mu * A.e[i] - p.alpha) / 2;  // qp_admm.h:125
        inv64[s] = -1.0 / (2 * Acoef);                        // qp_admm.h:126
    }
    bool ok = true;
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
    ok = t.grp_mem && t.grp_type && t.var_of_slot && t.v_maxlist && t.v_list_off && t.v_list && t.inv_coef && t.row_ptr && t.edge_var;
    if (!ok) {
        admm_device_destroy(d);
        return nullptr;
    }
    const size_t ts = d->f32 ? 4 : 8;
    size_t per_frame = (size_t) (4 * t.G_pad + t.V_pad + (d->blockmode ? 0 : t.n_vpass * L)) * ts + (size_t) t.nwords * 4;
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
        for (int mc = 0; mc < 2; mc++) {
            const void *kp = d->f32 ? (mc ? (const void *) admm_block_kernel<float, true> : (const void *) admm_block_kernel<float, false>)
                                    : (mc ? (const void *) admm_block_kernel<double, true> : (const void *) admm_block_kernel<double, false>);
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
    return d->blockmode && !d->guard;
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
    return hipLaunchKernel(d->kernel[a.mc ? 1 : 0], dim3(grid), dim3(d->block), args, d->lds_block, s);
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
    int64_t blocks = (a.frames + d->frames_per_block - 1) / d->frames_per_block;
    int grid = (int) std::min<int64_t>(blocks, d->grid_cap);
    const hipError_t e = d->f32 ? admm_launch_t<float>(d, a, grid, s) : admm_launch_t<double>(d, a, grid, s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace acg
