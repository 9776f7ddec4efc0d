// Paired-frame min-sum with packed half-precision messages (precision = ACG_LDPC_PREC_F16; min-sum only — the variant
// north_star names, which is NOT in the reference: parity unpinned).
//
// For codes whose message array fills the LDS of a CU (the 5000 x 10000 (3,6) code of BASELINE configs[4]: 120 KB in
// fp32) bp_block_kernel runs ONE 1024-thread workgroup per CU, and its waves spend half their time parked at the two
// barriers of a sweep and behind LDS latency (49 % SQ_WAIT_ANY, VALU 39 %, LDS 35 % busy).  Here every 32-bit LDS word
// carries the same edge of TWO frames as a half2: the layout, the index tables and the number of LDS instructions and
// barriers per sweep stay those of one fp32 frame, but each sweep advances two frames.  The arithmetic is packed too
// (v_pk_min/max/add/mul_f16 and plain 32-bit logic on the sign / hard-decision bits of both halves at once), so the VALU
// work per frame drops as well.
//
// Message word (16 bits per frame): sign | fp16 magnitude | LSB = posterior hard decision of the sending variable (as in
// the fp32 kernels, bp_core.inc).  Schedule and rules are those of the fp32 min-sum kernels: flooding, x <= 0 -> sign -1,
// hard decision = (posterior <= 0), output latched at the first zero syndrome, bp.h:183-199.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace acg {
#include "bp_core.inc"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
using u32 = uint32_t;

__device__ __forceinline__ h2 as_h2(u32 x) { return __builtin_bit_cast(h2, x); }
__device__ __forceinline__ u32 as_u(h2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ u32 pk_sub_u16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (us2) (__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)));
}
constexpr u32 SIGN2 = 0x80008000u, LSB2 = 0x00010001u, MAG2 = 0x7FFE7FFEu, INF2 = 0x7C007C00u;

// per half: 1 if the fp16 value is <= 0 (negative or +-0), else 0
__device__ __forceinline__ u32 le0(u32 x) {
    const u32 nz = pk_min_u16(x & 0x7FFF7FFFu, LSB2);  // 1 where the magnitude is non-zero
    return ((x >> 15) & LSB2) | (nz ^ LSB2);
}
// the same predicate left in bit 15 of each half (other bits: garbage), in two instructions: as a 16-bit integer, x - 1 has
// bit 15 set exactly for +0 (0x0000 - 1 = 0xFFFF) and for every negative value except -0 (0x8000 - 1 = 0x7FFF), whose own
// sign bit the OR supplies; positive values (and positive NaNs) keep bit 15 clear in both terms.  Round 3: replaces
// le0(x) << 15 (and, pk_min, shift, and, xor, or, shift) in the variable sweep — same bits, a third of the instructions.
__device__ __forceinline__ u32 le0_bit15(u32 x) { return pk_sub_u16(x, LSB2) | x; }
__device__ __forceinline__ u32 pk_sub_sat_u16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ u32 pk_mad_u16(u32 a, u32 b, u32 c) {
    return __builtin_bit_cast(u32, (us2) (__builtin_bit_cast(us2, a) * __builtin_bit_cast(us2, b) + __builtin_bit_cast(us2, c)));
}

// One check pass: D words at Ap[j*L] -> c->v words in place; returns the XOR word (LSB of each half = syndrome bit)
template <int D, int L, bool UNIFORM>
__device__ __forceinline__ u32 pair_check(u32 *__restrict__ Ap, int slot, const int *cnt, h2 scale) {
    u32 x[D];
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = Ap[j * L];
    u32 S = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) S ^= x[j];
    u32 a[D];
    h2 m1 = as_h2(INF2), m2 = as_h2(INF2);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        a[j] = x[j] & MAG2;
        if (!UNIFORM && !(slot < cnt[j + 1])) a[j] = INF2;  // padding slots are +0 and must not take part in the minimum
        const h2 aj = as_h2(a[j]);
        const h2 t = __builtin_elementwise_max(m1, aj);
        m1 = __builtin_elementwise_min(m1, aj);
        m2 = __builtin_elementwise_min(m2, t);
    }
    // Round 3: the two minima are scaled once per check instead of the selected one once per edge, and "m2 where this edge
    // holds the minimum, else m1" is u1s + e * (u2s - u1s) with e = 1 - min(1, a_j ^ m1) per half: the magnitudes have their LSB
    // clear, so a non-zero XOR is >= 2 and the saturating 1 - (a_j ^ m1) is exactly that e (one v_pk_sub_u16 clamp), and the
    // select is one v_pk_mad_u16 — three instructions per edge where there were five (xor, min, sub, bfi, mul); same bits.
    const u32 u1 = as_u(m1);
    const u32 u1s = as_u(m1 * scale), u2s = as_u(m2 * scale);
    const u32 dd = pk_sub_u16(u2s, u1s);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        // per half: the minimum over the OTHER edges = m2 where this edge holds the minimum, else m1
        const u32 e = pk_sub_sat_u16(LSB2, a[j] ^ u1);  // 1 where a_j == m1, else 0
        const u32 sc = pk_mad_u16(e, dd, u1s);
        const u32 ob = (sc & 0x7FFF7FFFu) | ((S ^ x[j]) & SIGN2);  // sign product, bp.h:54
        if (UNIFORM ? (slot < cnt[1]) : (slot < cnt[j + 1])) Ap[j * L] = ob;
    }
    return S;
}

// One variable pass with the D message positions known; returns the posterior hard decisions (bit 0 / bit 16)
template <int D>
__device__ __forceinline__ u32 pair_var(u32 *__restrict__ A, const int (&pos)[D], h2 llr, int slot, const int *cnt) {
    h2 c[D], pre[D];
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = as_h2(A[pos[k]]);
    h2 s = as_h2(0u);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        pre[k] = s;
        s += c[k];
    }
    const u32 hard = (le0_bit15(as_u(llr + s)) >> 15) & LSB2;  // estimate() <= 0, bp.h:85-90,193
    h2 suf = as_h2(0u);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        const u32 xk = as_u(llr + (pre[k] + suf));  // bp.h:78-82
        suf += c[k];
        const u32 ob = (xk & MAG2) | ((le0_bit15(xk) & SIGN2) | hard);  // magnitude | sign (x <= 0 -> -1, bp.h:82) | hard-decision LSB
        if (slot < cnt[k + 1]) A[pos[k]] = ob;
    }
    return hard;
}

#define ACG_PAIR_CSWITCH(md, CALL)                                                                             \
    switch (md) {                                                                                              \
        case 1: CALL(1); break;                                                                                \
        case 2: CALL(2); break;                                                                                \
        case 3: CALL(3); break;                                                                                \
        case 4: CALL(4); break;                                                                                \
        case 5: CALL(5); break;                                                                                \
        case 6: CALL(6); break;                                                                                \
        case 7: CALL(7); break;                                                                                \
        case 8: CALL(8); break;                                                                                \
        default: break;                                                                                        \
    }

}  // namespace

// L threads = one PAIR of frames (2f, 2f + 1).  Variable degree <= 4, check degree <= 8, at most 12 passes of variables:
// the variable-side index table and the channel LLRs of both frames live in registers.
// REG (regular code: every check has one degree, every variable one degree): the pass loops are unrolled and the degree
// is dispatched once per sweep, so the per-pass registers (LLRs, index words) are never indexed dynamically.
template <int L, bool REG>
__global__ void __launch_bounds__(L) bp_pair_kernel(const BpTables t, const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NVP = 12;
    __shared__ unsigned long long fr_lds;
    __shared__ u32 bad_lds[2];  // syndrome flags of the two frames (bit 0 / bit 1), double-buffered by sweep parity
    const int l = threadIdx.x;
    u32 *A = reinterpret_cast<u32 *>(smem);
    u32 *OB = A + t.a_words;  // [2][nwords] packed hard decisions of the two frames
    int ccnt[10], vcnt[6];
#pragma unroll
    for (int j = 0; j < 10; ++j) ccnt[j] = sload(t.c_cnt_ge, j);
#pragma unroll
    for (int j = 0; j < 6; ++j) vcnt[j] = sload(t.v_cnt_ge, j);
    // loop-invariant per-thread structure: message positions of my variable in every pass (two 16-bit positions per
    // register).  The variable ids are re-read per frame pair (12 cached loads): kept in registers they would cost 12 of
    // them plus 24 hoisted 64-bit symbol addresses, which do not fit beside the rest at 1024 threads (128 registers).
    auto var_id = [&](int p) {
        int v = (p < t.n_vpass) ? t.v_var[p * L + l] : -1;
        asm volatile("" : "+v"(v));
        return v;
    };
    u32 ir[2 * NVP];
#pragma unroll
    for (int p = 0; p < NVP; ++p) {
        u32 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (u32) t.zero_pos;
        if (p < t.n_vpass) {
            const int md = sload(t.v_pass, 2 * p), ioff = sload(t.v_pass, 2 * p + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < md) w[k] = t.v_apos[ioff + k * L + l];
        }
        ir[2 * p] = w[0] | (w[1] << 16);
        ir[2 * p + 1] = w[2] | (w[3] << 16);
    }
    const h2 scale = {(_Float16) a.ms_scale, (_Float16) a.ms_scale};
    const int cd0 = sload(t.c_pass, 0);
    const int vd0 = sload(t.v_pass, 0);
    constexpr bool uniform_c = REG, uniform_v = REG;  // the host picks the instance (bp_pair_kernel_ptr)
    // padding words and the zero cell are +0 for the whole launch
    for (int w = l; w < t.a_words; w += L) A[w] = 0u;

    for (;;) {
        __syncthreads();
        if (l == 0) {
            fr_lds = atomicAdd(a.work_counter, 1ull);
            bad_lds[0] = bad_lds[1] = 0u;
        }
        __syncthreads();
        const int64_t pair = (int64_t) fr_lds;
        const int64_t f0 = 2 * pair, f1 = f0 + 1;
        if (f0 >= a.frames) break;
        const bool have1 = f1 < a.frames;
        // ---- channel LLRs (channel.h:14-16) of both frames -> registers, then the first v->c sweep (bp.h:184) ----
        h2 llr[NVP];
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            float y0 = 0.0f, y1 = 0.0f;
            const int v = var_id(p);
            if (p < t.n_vpass && v >= 0) {
                if (a.y_is_f64) {
                    y0 = (float) (2 * reinterpret_cast<const double *>(a.y)[(size_t) f0 * t.n + v] / a.var);
                    if (have1) y1 = (float) (2 * reinterpret_cast<const double *>(a.y)[(size_t) f1 * t.n + v] / a.var);
                } else {
                    y0 = (float) ((double) reinterpret_cast<const float *>(a.y)[(size_t) f0 * t.n + v] * a.inv_var2);
                    if (have1) y1 = (float) ((double) reinterpret_cast<const float *>(a.y)[(size_t) f1 * t.n + v] * a.inv_var2);
                }
            }
            llr[p] = h2{(_Float16) y0, (_Float16) y1};
            if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four passes of symbol loads in flight at a time
        }
        u32 hard = 0;  // bit p: frame 0, bit 16 + p: frame 1
#pragma unroll
        for (int p = 0; p < NVP; ++p) {
            if (p >= t.n_vpass) continue;
            const int md = REG ? vd0 : sload(t.v_pass, 2 * p), slot = p * L + l;
            const u32 x = as_u(llr[p]);
            const u32 h = le0(x);
            hard |= h << p;
            const u32 ob = (x & MAG2) | (h << 15) | h;  // every mailbox is (0, +1): all outgoing words are |llr| with its sign
            const u32 w0 = ir[2 * p], w1 = ir[2 * p + 1];
            if (md >= 1 && slot < vcnt[1]) A[w0 & 0xFFFFu] = ob;
            if (md >= 2 && slot < vcnt[2]) A[w0 >> 16] = ob;
            if (md >= 3 && slot < vcnt[3]) A[w1 & 0xFFFFu] = ob;
            if (md >= 4 && slot < vcnt[4]) A[w1 >> 16] = ob;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // ---- sweeps (bp.h:186-197): the check sweep also delivers the syndrome of the hard bits riding in the words ----
        int it = 0;
        bool lat0 = false, lat1 = !have1;
        for (;;) {
            u32 acc = 0;
            if constexpr (uniform_c) {
#define ACG_CALL(D) \
    for (int p = 0; p < t.n_cpass; ++p) acc |= pair_check<D, L, true>(A + p * (D * L) + l, p * L + l, ccnt + (D - 1), scale)
                ACG_PAIR_CSWITCH(cd0, ACG_CALL)
#undef ACG_CALL
            } else {
                for (int p = 0; p < t.n_cpass; ++p) {
                    const int md = sload(t.c_pass, 2 * p);
                    u32 *Ap = A + sload(t.c_pass, 2 * p + 1) + l;
#define ACG_CALL(D) acc |= pair_check<D, L, false>(Ap, p * L + l, ccnt, scale)
                    ACG_PAIR_CSWITCH(md, ACG_CALL)
#undef ACG_CALL
                }
            }
            // per-frame OR over the workgroup (__syncthreads_or only tells "any"): one ballot per frame and wavefront, one
            // LDS atomic per wavefront that has something to report, one barrier
            {
                const bool any0 = __ballot((acc & 1u) != 0u) != 0ull, any1 = __ballot((acc & 0x10000u) != 0u) != 0ull;
                if ((l & 63) == 0 && (any0 || any1)) atomicOr(&bad_lds[it & 1], (any0 ? 1u : 0u) | (any1 ? 0x10000u : 0u));
            }
            __syncthreads();
            const u32 bad = bad_lds[it & 1];  // bit 0: frame 0 fails a check, bit 16: frame 1
            if (l == 0) bad_lds[(it + 1) & 1] = 0u;  // next sweep's slot (its writers are two barriers away)
            const bool in_budget = it > 0 && it <= a.max_iter;     // bp.h:195 (max_iter = 0: never)
            const bool conv0 = in_budget && !(bad & 1), conv1 = in_budget && have1 && !(bad & 0x10000);
            const bool out0 = conv0 && !lat0, out1 = conv1 && !lat1;
            const bool last = it >= a.max_iter;
            const bool fail0 = last && !conv0 && !lat0, fail1 = last && have1 && !conv1 && !lat1;
            if (out0 || out1 || fail0 || fail1) {  // block-uniform
                for (int w = l; w < 2 * t.nwords; w += L) OB[w] = 0u;
                __syncthreads();
#pragma unroll
                for (int p = 0; p < NVP; ++p) {
                    const int v = var_id(p);
                    if (p >= t.n_vpass || v < 0) continue;
                    if (out0 && ((hard >> p) & 1u)) atomicOr(&OB[v >> 5], 1u << (v & 31));
                    if (out1 && ((hard >> (16 + p)) & 1u)) atomicOr(&OB[t.nwords + (v >> 5)], 1u << (v & 31));
                }
                __syncthreads();
                if (a.out_bits) {
                    if (out0 || fail0)
                        for (int w = l; w < t.nwords; w += L) a.out_bits[(size_t) f0 * t.nwords + w] = OB[w];  // failure: empty word, bp.h:198
                    if (out1 || fail1)
                        for (int w = l; w < t.nwords; w += L) a.out_bits[(size_t) f1 * t.nwords + w] = OB[t.nwords + w];
                }
                if (l == 0) {
                    const int itv = it < a.max_iter ? it : a.max_iter;
                    if (out0 || fail0) {
                        if (a.out_ok) a.out_ok[f0] = out0 ? 1 : 0;
                        if (a.out_iters) a.out_iters[f0] = itv;
                    }
                    if (out1 || fail1) {
                        if (a.out_ok) a.out_ok[f1] = out1 ? 1 : 0;
                        if (a.out_iters) a.out_iters[f1] = itv;
                    }
                }
                lat0 = lat0 || out0 || fail0;
                lat1 = lat1 || out1 || fail1;
            }
            if (last || (a.early_exit && lat0 && lat1)) break;
            hard = 0;
            // the unpacked message positions (three or four per pass) must not be hoisted out of the sweep loop: that would
            // turn the 24 packed index registers into 36-48 and spill.  Making them opaque per sweep costs no instruction.
#pragma unroll
            for (int i = 0; i < 2 * NVP; ++i) asm volatile("" : "+v"(ir[i]));
            if constexpr (uniform_v) {
                // every variable has the same degree: one dispatch per sweep, the pass loop unrolled so that the LLRs and the
                // index words are plain registers
#define ACG_VCALL(D)                                                                                              \
    _Pragma("unroll") for (int p = 0; p < NVP; ++p) if (p < t.n_vpass) {                                          \
        const u32 w0 = ir[2 * p], w1 = ir[2 * p + 1];                                                             \
        int pos[D];                                                                                               \
        _Pragma("unroll") for (int k = 0; k < D; ++k) pos[k] = (int) (((k < 2 ? w0 : w1) >> (16 * (k & 1))) & 0xFFFFu); \
        hard |= pair_var<D>(A, pos, llr[p], p * L + l, vcnt) << p;                                                \
        if (p & 1) __builtin_amdgcn_sched_barrier(0); /* two passes in flight at a time: more would not fit in 128 registers */ \
    }
                switch (vd0) {
                    case 1: ACG_VCALL(1) break;
                    case 2: ACG_VCALL(2) break;
                    case 3: ACG_VCALL(3) break;
                    default: ACG_VCALL(4) break;
                }
#undef ACG_VCALL
            } else {
                for (int p = 0; p < t.n_vpass; ++p) {
                    const int md = sload(t.v_pass, 2 * p), slot = p * L + l;
                    const u32 w0 = ir[2 * p], w1 = ir[2 * p + 1];
                    u32 h;
                    switch (md) {
                        case 1: { const int pos[1] = {(int) (w0 & 0xFFFFu)}; h = pair_var<1>(A, pos, llr[p], slot, vcnt); } break;
                        case 2: { const int pos[2] = {(int) (w0 & 0xFFFFu), (int) (w0 >> 16)}; h = pair_var<2>(A, pos, llr[p], slot, vcnt); } break;
                        case 3: { const int pos[3] = {(int) (w0 & 0xFFFFu), (int) (w0 >> 16), (int) (w1 & 0xFFFFu)}; h = pair_var<3>(A, pos, llr[p], slot, vcnt); } break;
                        case 4: { const int pos[4] = {(int) (w0 & 0xFFFFu), (int) (w0 >> 16), (int) (w1 & 0xFFFFu), (int) (w1 >> 16)}; h = pair_var<4>(A, pos, llr[p], slot, vcnt); } break;
                        default: h = le0(as_u(llr[p])); break;  // isolated variable: estimate() == channel LLR
                    }
                    hard |= h << p;
                }
            }
            __syncthreads();
            it += 1;
        }
    }
}

const void *bp_pair_kernel_ptr(int L, bool regular) {
    if (L == 1024) return regular ? (const void *) bp_pair_kernel<1024, true> : (const void *) bp_pair_kernel<1024, false>;
    if (L == 256) return regular ? (const void *) bp_pair_kernel<256, true> : (const void *) bp_pair_kernel<256, false>;
    return nullptr;
}

}  // namespace acg
