// Fused belief-propagation kernels for gfx950 (MI355X).
//
// Replaces, for a whole batch of frames per launch, the reference's per-frame
//   BeliefPropagation::decode            algo/bp.h:183-199
//   c_receive_messages / VNode::message  algo/bp.h:160-169, 77-83   (variable -> check)
//   v_receive_messages / CNode::message  algo/bp.h:171-181, 49-57   (check -> variable)
//   VNode::estimate + IsCodeword         algo/bp.h:85-90, 191-196 ; utils/codeword.h:90-95
// and, in Monte-Carlo mode, transmit (utils/channel.h:18-26) and the per-frame classification
// of exp() (experiment.h:109-120).
//
// Mapping: L lanes of a 64-wide wavefront (L = 64, 32 or 16) cooperate on ONE
// frame; the frame's E messages live in LDS for all iterations, updated in place, so HBM sees
// only the channel symbols in and the packed hard decisions out.  The Tanner graph is shared by
// every frame of the launch, so all graph indices are either implicit in the layout (check side:
// unit-stride, bank-conflict-free) or small read-only tables (variable side).  No cross-lane
// traffic is needed inside a phase: each lane owns whole nodes and forms the exclude-self sums
// (bp.h:50-55, 78-81) with a prefix/suffix scan in registers.  The per-check parity of the hard
// decisions rides in the LSB of each v->c magnitude, so the syndrome test costs one XOR per edge
// and a wave ballot.
//
// A wavefront is a persistent worker: each L-lane group walks frames g, g+G, g+2G, ... and
// restarts on a new frame the moment its current one reaches a zero syndrome (the reference's
// early exit, bp.h:195-196), independently of the other groups in the wave.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace acg {

// ------------------------------------------------------------------------------------------
// bit-level helpers
template <typename T> struct FpBits;
template <> struct FpBits<float> {
    using U = uint32_t;
    static constexpr U SIGN = 0x80000000u;
    static __device__ __forceinline__ U to(float x) { return __float_as_uint(x); }
    static __device__ __forceinline__ float from(U u) { return __uint_as_float(u); }
};
template <> struct FpBits<double> {
    using U = uint64_t;
    static constexpr U SIGN = 0x8000000000000000ull;
    static __device__ __forceinline__ U to(double x) { return (U) __double_as_longlong(x); }
    static __device__ __forceinline__ double from(U u) { return __longlong_as_double((long long) u); }
};

// Single-wave producer/consumer ordering through LDS: the DS queue of one wavefront is in-order,
// so only the compiler has to be stopped from moving accesses across the phase boundary.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// phi(x) = -log(tanh(x/2))   (bp.h:34), x >= 0.
//
// The reference evaluates it in x87 long double, where tanh(x/2) rounds to 1 and phi becomes
// (minus) zero for x >= 45.7477...; phi(0) = +inf; phi(NaN) = NaN.  Those three behaviours are
// reproduced; in between the value is computed to ~1e-7 relative (fp32) from
//   x >= 2   : 2*atanh(t), t = e^-x, as the odd series 2t(1 + u/3 + u^2/5 + u^3/7), u = t^2
//   x <  2   : ln(b) - ln(a) with a/b = tanh(x/2):  a = 1-t, b = 1+t          (0.25 <= x)
//                                                   a = y*(1 - y^2/3 + 2y^4/15 - 17y^6/315), b = 1,  y = x/2
// (the direct 1-t loses all relative accuracy for small x, the series has none left for large t).
#define ACG_PHI_SAT 45.7477139f

__device__ __forceinline__ float phi_f(float x) {
    const float t = __builtin_amdgcn_exp2f(x * -1.44269504088896341f);
    const float u = t * t;
    float hi = fmaf(u, 1.0f / 7.0f, 0.2f);
    hi = fmaf(hi, u, 1.0f / 3.0f);
    hi = fmaf(hi, u, 1.0f);
    hi = (t + t) * hi;
    const float y = 0.5f * x;
    const float y2 = y * y;
    float ps = fmaf(y2, -17.0f / 315.0f, 2.0f / 15.0f);
    ps = fmaf(ps, y2, -1.0f / 3.0f);
    ps = fmaf(ps, y2, 1.0f);
    const bool small = x < 0.25f;
    const float a = small ? y * ps : 1.0f - t;
    const float b = small ? 1.0f : 1.0f + t;
    const float lo = 0.693147180559945309f * (__builtin_amdgcn_logf(b) - __builtin_amdgcn_logf(a));
    float r = (x >= 2.0f) ? hi : lo;
    r = (x >= ACG_PHI_SAT) ? 0.0f : r;  // also maps +inf -> 0; NaN compares false everywhere -> lo = NaN
    return r;
}

__device__ __forceinline__ double phi_f(double x) {
    // fp64 "strict" mode: library accuracy (~1e-16), same saturation points as the reference
    if (x >= 45.747713916956390) return 0.0;
    if (x >= 1.0) {
        const double t = exp(-x);
        return log1p(2.0 * t / (1.0 - t));
    }
    return -log(tanh(0.5 * x));
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter = (frame_lo, frame_hi, quad, 0), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// two uint32 -> two N(0,1) (Box-Muller; u1 in (0,1) with a 2^-33 floor: tails to 6.7 sigma)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1) {
    const float u1 = ((float) a + 0.5f) * 2.3283064365386963e-10f;          // (a+0.5)/2^32
    const float u2 = (float) (b >> 8) * 5.9604644775390625e-08f;           // [0,1) 24 bits, in revolutions
    const float r = sqrtf(-2.0f * 0.693147180559945309f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

// ------------------------------------------------------------------------------------------
// Wave-uniform tables are read through the constant address space so the compiler issues scalar
// (SMEM) loads into SGPRs instead of per-lane vector loads followed by v_readfirstlane.
typedef const int32_t __attribute__((address_space(4))) *sconst_i32;
__device__ __forceinline__ int sload(const int32_t *p, int i) { return ((sconst_i32) (p))[i]; }

// ------------------------------------------------------------------------------------------
// One pass = the L nodes dealt to the L lanes of a frame group.  D (the largest degree in the
// pass) is a compile-time constant: the D LDS reads are issued back to back, then everything is
// straight-line register code, then the D LDS writes.  Dispatch on D is a wave-uniform switch.
template <typename T, int D, int L, int ALGO>
struct BpPass {
    using B = FpBits<T>;
    using U = typename B::U;
    static constexpr U SIGN = B::SIGN;
    static constexpr U ONE = (U) 1;

    // syndrome contribution of the checks in this pass (LSB of the XOR of the v->c words)
    static __device__ __forceinline__ U syn(const T *__restrict__ Ap) {
        T x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = Ap[j * L];
        U S = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) S ^= B::to(x[j]);
        return S;
    }

    // check -> variable (bp.h:171-181 / CNode::message bp.h:49-57), in place.  cnt[j] = number of
    // checks of degree >= j: slot < cnt[j+1] <=> edge j of this lane's check exists.
    static __device__ __forceinline__ void check(T *__restrict__ Ap, int slot, const int *cnt, bool write, T ms_scale) {
        T x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = Ap[j * L];
        U S = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) S ^= B::to(x[j]);
        T out[D];
        if (ALGO == 0) {
            // exclude-self sums of the phi magnitudes: prefix + suffix (bp.h:50-55)
            T mag[D], pre[D];
            T s = 0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                mag[j] = B::from(B::to(x[j]) & ~SIGN);
                pre[j] = s;
                s += mag[j];
            }
            T suf = 0;
#pragma unroll
            for (int j = D - 1; j >= 0; --j) {
                out[j] = phi_f(pre[j] + suf);
                suf += mag[j];
            }
        } else {
            // min-sum: two smallest magnitudes (padding slots are +0 and must not take part)
            T m1 = (T) INFINITY, m2 = (T) INFINITY;
            int am = -1;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const T a = B::from(B::to(x[j]) & ~SIGN);
                const bool real = slot < cnt[j + 1];
                const bool lt1 = real && (a < m1);
                const bool lt2 = real && (a < m2);
                m2 = lt1 ? m1 : (lt2 ? a : m2);
                am = lt1 ? j : am;
                m1 = lt1 ? a : m1;
            }
#pragma unroll
            for (int j = 0; j < D; ++j) out[j] = ms_scale * ((j == am) ? m2 : m1);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const U ob = (B::to(out[j]) & ~SIGN) | ((S ^ B::to(x[j])) & SIGN);  // sign product, bp.h:54
            if (write && slot < cnt[j + 1]) Ap[j * L] = B::from(ob);
        }
    }

    // variable -> check (bp.h:160-169 / VNode::message bp.h:77-83) + posterior hard decision
    // (bp.h:85-90,193), whose bit rides in the LSB of every outgoing magnitude.
    template <typename IdxPtr>
    static __device__ __forceinline__ void var(T *__restrict__ A, IdxPtr ip, T llr, int slot, const int *cnt, bool write) {
        int pos[D];
#pragma unroll
        for (int k = 0; k < D; ++k) pos[k] = ip[k * L];
        T c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) c[k] = A[pos[k]];
        T pre[D];
        T s = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            pre[k] = s;
            s += c[k];
        }
        const T total = llr + s;                            // estimate(), bp.h:85-90
        const U hard = (total <= (T) 0) ? ONE : (U) 0;      // bp.h:193 (NaN -> 0)
        T suf = 0;
        U ob[D];
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
            const T xk = llr + (pre[k] + suf);              // bp.h:78-82
            suf += c[k];
            const T ax = B::from(B::to(xk) & ~SIGN);
            const T mg = (ALGO == 0) ? phi_f(ax) : ax;
            ob[k] = (B::to(mg) & ~SIGN & ~ONE) | hard | ((xk <= (T) 0) ? SIGN : (U) 0);
        }
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (write && slot < cnt[k + 1]) A[pos[k]] = B::from(ob[k]);
    }
};

#define ACG_PASS_SWITCH(md, CALL)                                                                              \
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

// One frame-group's view of the kernel state.
template <typename T, int MAXD, int L, int ALGO, bool IDXLDS>
struct BpCore {
    using B = FpBits<T>;
    using U = typename B::U;
    static constexpr U SIGN = B::SIGN;
    static constexpr U ONE = (U) 1;

    const BpTables &t;
    T *__restrict__ A;          // messages, in place (LDS)
    T *__restrict__ LLR;        // channel LLR per variable slot (LDS)
    uint32_t *__restrict__ OB;  // packed hard decisions staging (LDS)
    const uint16_t *IDX;        // variable-side index table: block-shared LDS copy or global
    const int l;                // lane within the frame group
    const T ms_scale;
    int ccnt[MAXD + 2], vcnt[MAXD + 2];  // wave-uniform (SGPR) degree histograms

    __device__ BpCore(const BpTables &t_, T *A_, T *LLR_, uint32_t *OB_, const uint16_t *IDX_, int l_, float s_)
        : t(t_), A(A_), LLR(LLR_), OB(OB_), IDX(IDX_), l(l_), ms_scale((T) s_) {
#pragma unroll
        for (int j = 0; j < MAXD + 2; ++j) {
            ccnt[j] = sload(t.c_cnt_ge, j);
            vcnt[j] = sload(t.v_cnt_ge, j);
        }
    }

    // XOR of the hard-decision bits of each check's variables -> true if any check of this lane fails
    __device__ __forceinline__ bool syndrome_bad() const {
        U acc = 0;
        for (int p = 0; p < t.n_cpass; ++p) {
            const int md = sload(t.c_pass, 2 * p);
            const T *Ap = A + sload(t.c_pass, 2 * p + 1) + l;
            if (MAXD <= 8 || md <= 8) {
#define ACG_CALL(D) acc |= BpPass<T, D, L, ALGO>::syn(Ap)
                ACG_PASS_SWITCH(md, ACG_CALL)
#undef ACG_CALL
            } else {
                U S = 0;
                for (int j = 0; j < md; ++j) S ^= B::to(Ap[j * L]);
                acc |= S;
            }
        }
        return (acc & ONE) != 0;
    }

    __device__ __forceinline__ void check_phase(bool write) {
        for (int p = 0; p < t.n_cpass; ++p) {
            const int md = sload(t.c_pass, 2 * p);
            T *Ap = A + sload(t.c_pass, 2 * p + 1) + l;
            const int slot = p * L + l;
            if (MAXD <= 8 || md <= 8) {
#define ACG_CALL(D) BpPass<T, D, L, ALGO>::check(Ap, slot, ccnt, write, ms_scale)
                ACG_PASS_SWITCH(md, ACG_CALL)
#undef ACG_CALL
            } else {
                check_generic(Ap, slot, md, write);
            }
        }
    }

    __device__ __forceinline__ void var_phase(bool write) {
        for (int p = 0; p < t.n_vpass; ++p) {
            const int md = sload(t.v_pass, 2 * p);
            const int ioff = sload(t.v_pass, 2 * p + 1);
            const int slot = p * L + l;
            const T llr = LLR[slot];
            const uint16_t *ip = IDX + ioff + l;
            if (MAXD <= 8 || md <= 8) {
#define ACG_CALL(D) BpPass<T, D, L, ALGO>::var(A, ip, llr, slot, vcnt, write)
                ACG_PASS_SWITCH(md, ACG_CALL)
#undef ACG_CALL
            } else {
                var_generic(ip, llr, slot, md, write);
            }
        }
    }

    // ---- degree > 8: rolled loops (rare: high-rate codes); same arithmetic -------------------
    __device__ __noinline__ void check_generic(T *__restrict__ Ap, int slot, int md, bool write) {
        T x[MAXD], pre[MAXD];
        U S = 0;
        T s = 0;
        for (int j = 0; j < md; ++j) {
            x[j] = Ap[j * L];
            S ^= B::to(x[j]);
            pre[j] = s;
            s += B::from(B::to(x[j]) & ~SIGN);
        }
        if (ALGO == 0) {
            T suf = 0;
            for (int j = md - 1; j >= 0; --j) {
                const T ph = phi_f(pre[j] + suf);
                suf += B::from(B::to(x[j]) & ~SIGN);
                const U ob = (B::to(ph) & ~SIGN) | ((S ^ B::to(x[j])) & SIGN);
                if (write && slot < ccnt[j + 1]) Ap[j * L] = B::from(ob);
            }
        } else {
            T m1 = (T) INFINITY, m2 = (T) INFINITY;
            int am = -1;
            for (int j = 0; j < md; ++j) {
                const T a = B::from(B::to(x[j]) & ~SIGN);
                const bool real = slot < ccnt[j + 1];
                const bool lt1 = real && (a < m1);
                const bool lt2 = real && (a < m2);
                m2 = lt1 ? m1 : (lt2 ? a : m2);
                am = lt1 ? j : am;
                m1 = lt1 ? a : m1;
            }
            for (int j = 0; j < md; ++j) {
                const T mg = ms_scale * ((j == am) ? m2 : m1);
                const U ob = (B::to(mg) & ~SIGN) | ((S ^ B::to(x[j])) & SIGN);
                if (write && slot < ccnt[j + 1]) Ap[j * L] = B::from(ob);
            }
        }
    }

    __device__ __noinline__ void var_generic(const uint16_t *ip, T llr, int slot, int md, bool write) {
        int pos[MAXD];
        T c[MAXD], pre[MAXD];
        T s = 0;
        for (int k = 0; k < md; ++k) {
            pos[k] = ip[k * L];
            c[k] = A[pos[k]];
            pre[k] = s;
            s += c[k];
        }
        const T total = llr + s;
        const U hard = (total <= (T) 0) ? ONE : (U) 0;
        T suf = 0;
        for (int k = md - 1; k >= 0; --k) {
            const T xk = llr + (pre[k] + suf);
            suf += c[k];
            const T ax = B::from(B::to(xk) & ~SIGN);
            const T mg = (ALGO == 0) ? phi_f(ax) : ax;
            const U ob = (B::to(mg) & ~SIGN & ~ONE) | hard | ((xk <= (T) 0) ? SIGN : (U) 0);
            if (write && slot < vcnt[k + 1]) A[pos[k]] = B::from(ob);
        }
    }

    // hard decision of the variable in (pass p, this lane) as stored by the last var_phase
    __device__ __forceinline__ uint32_t hard_bit(int p) const {
        const int slot = p * L + l;
        if (slot < vcnt[1]) {
            const int pos0 = IDX[sload(t.v_pass, 2 * p + 1) + l];
            return (uint32_t) (B::to(A[pos0]) & ONE);
        }
        return (LLR[slot] <= (T) 0) ? 1u : 0u;  // isolated variable: estimate() == channel LLR
    }

    // pack the frame's hard decisions into OB[0..nwords)
    __device__ __forceinline__ void pack_bits() {
        for (int w = l; w < t.nwords; w += L) OB[w] = 0u;
        wave_sync();
        for (int p = 0; p < t.n_vpass; ++p) {
            const int v = t.v_var[p * L + l];
            if (v >= 0 && hard_bit(p)) atomicOr(&OB[v >> 5], 1u << (v & 31));
        }
        wave_sync();
    }
};

// group-wide OR of a per-lane predicate (groups are L consecutive lanes of the wavefront)
template <int L>
__device__ __forceinline__ bool group_any(bool pred, int g) {
    const unsigned long long b = __ballot(pred);
    if (L == 64) return b != 0ull;
    const unsigned long long mask = ((1ull << (L & 63)) - 1ull) << (g * L);
    return (b & mask) != 0ull;
}

template <int L>
__device__ __forceinline__ int group_sum(int v, int l) {
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
template <typename T, int MAXD, int L, int ALGO, bool MC, bool IDXLDS>
__global__ void __launch_bounds__(256) bp_fused_kernel(const BpTables t, const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Core = BpCore<T, MAXD, L, ALGO, IDXLDS>;
    constexpr int FPW = 64 / L;  // frames in flight per wavefront
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l = lane % L;
    const int g = lane / L;
    const int waves_per_block = blockDim.x >> 6;
    const int grp_in_block = wave * FPW + g;
    // block-shared copy of the variable-side index table (read every iteration by every wave)
    const uint16_t *IDX = t.v_apos;
    if (IDXLDS) {
        uint16_t *idx_lds = reinterpret_cast<uint16_t *>(smem);
        for (int i = threadIdx.x; i < t.v_apos_len; i += blockDim.x) idx_lds[i] = t.v_apos[i];
        __syncthreads();
        IDX = idx_lds;
    }
    unsigned char *base = smem + t.idx_lds_bytes + (size_t) grp_in_block * t.lds_bytes_per_frame;
    T *A = reinterpret_cast<T *>(base);
    T *LLR = A + t.a_words;
    uint32_t *OB = reinterpret_cast<uint32_t *>(LLR + t.llr_words);
    Core core(t, A, LLR, OB, IDX, l, a.ms_scale);

    const int64_t n_groups = (int64_t) gridDim.x * waves_per_block * FPW;
    int64_t frame = ((int64_t) blockIdx.x * waves_per_block + wave) * FPW + g;
    bool active = frame < a.frames;
    bool need_init = active;
    bool latched = false;
    int it = 0;
    int ham = 0;  // raw-channel errors of the current frame (MC)
    // per-group MC accumulators (flushed once at the end)
    unsigned int acc_correct = 0, acc_pseudo = 0, acc_total = 0;
    unsigned long long acc_ham = 0, acc_ham_ok = 0, acc_ham_wrong = 0, acc_iters = 0;

    for (;;) {
        // ---- syndrome of the estimate produced by the previous variable phase -----------------
        // (a group that has not been initialised yet has it == 0 and ignores the result)
        const bool bad = group_any<L>(core.syndrome_bad(), g);
        const bool conv = active && it > 0 && !bad;                         // bp.h:195
        const bool out_now = conv && !latched;
        const bool finish = active && ((a.early_exit && conv) || it >= a.max_iter);
        const bool fail_now = finish && !conv && !latched;
        if (__ballot(out_now || fail_now) != 0ull) {
            if (out_now) core.pack_bits();
            else if (fail_now) {
                for (int w = l; w < t.nwords; w += L) OB[w] = 0u;  // reference returns an empty vector, bp.h:198
                wave_sync();
            }
            if (out_now || fail_now) {
                if (a.out_bits)
                    for (int w = l; w < t.nwords; w += L) a.out_bits[(size_t) frame * t.nwords + w] = OB[w];
                if (l == 0) {
                    if (a.out_ok) a.out_ok[frame] = out_now ? 1 : 0;
                    if (a.out_iters) a.out_iters[frame] = it;
                }
                if (MC) {
                    bool neq = false;
                    const int64_t gf = a.first_frame + frame;
                    for (int w = l; w < t.nwords; w += L) {
                        const uint32_t cwv = a.cw_packed ? a.cw_packed[(size_t) (gf % a.n_cw) * t.nwords + w] : 0u;
                        neq |= (OB[w] != cwv);
                    }
                    // every lane of a group is in this branch together: the ballot is complete per group
                    const bool differ = group_any<L>(neq, g);
                    const bool correct = out_now && !differ;  // experiment.h:110-114
                    acc_correct += correct;
                    acc_pseudo += (out_now && differ);          // experiment.h:115-116
                    acc_total += 1;
                    acc_ham += ham;
                    acc_ham_ok += correct ? ham : 0;
                    acc_ham_wrong += correct ? 0 : ham;
                    acc_iters += it;
                }
                latched = true;
            }
        }
        if (finish) {
            frame += n_groups;
            active = frame < a.frames;
            need_init = active;
        }
        if (__ballot(active) == 0ull) break;

        // ---- (re)start groups on a new frame -------------------------------------------------
        if (__ballot(need_init) != 0ull) {
            wave_sync();
            const uint32_t *cw = nullptr;
            if (MC && need_init) {
                const int64_t gf = a.first_frame + frame;
                if (a.cw_packed) cw = a.cw_packed + (size_t) (gf % a.n_cw) * t.nwords;
                // stage the frame's noisy symbols in natural order in A[0..n)
                const int nq = (t.n + 3) >> 2;
                for (int q = l; q < nq; q += L) {
                    uint32_t r[4];
                    philox4x32_10((uint32_t) gf, (uint32_t) (gf >> 32), (uint32_t) q, 0u, (uint32_t) a.seed,
                                  (uint32_t) (a.seed >> 32), r);
                    float z[4];
                    box_muller(r[0], r[1], z[0], z[1]);
                    box_muller(r[2], r[3], z[2], z[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = 4 * q + e;
                        if (v < t.n) {
                            const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
                            A[v] = (T) ((bit ? -1.0f : 1.0f) + a.sigma * z[e]);  // channel.h:24
                        }
                    }
                }
            }
            wave_sync();
            int my_ham = 0;
            if (need_init) {
                // channel LLRs (channel.h:14-16) into slot order
                for (int p = 0; p < t.n_vpass; ++p) {
                    const int slot = p * L + l;
                    const int v = t.v_var[slot];
                    T llr = (T) 0;
                    if (v >= 0) {
                        if (MC) {
                            const T yv = A[v];
                            const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
                            // HammingDistanceTracker (experiment.h:33-46)
                            my_ham += ((!bit && yv <= (T) 0) || (bit && yv > (T) 0)) ? 1 : 0;
                            llr = (T) (yv * (T) a.inv_var2);
                        } else if (a.y_is_f64) {
                            const double yv = reinterpret_cast<const double *>(a.y)[(size_t) frame * t.n + v];
                            llr = (T) (2 * yv / a.var);
                        } else {
                            const float yv = reinterpret_cast<const float *>(a.y)[(size_t) frame * t.n + v];
                            llr = (T) (2 * (double) yv / a.var);
                        }
                    }
                    LLR[slot] = llr;
                }
            }
            wave_sync();  // all reads of the staged symbols are done before A is cleared
            if (MC) {
                const int hs = group_sum<L>(need_init ? my_ham : 0, l);
                if (need_init) ham = hs;
            }
            if (need_init)
                for (int w = l; w < t.a_words; w += L) A[w] = (T) 0;  // mailboxes (0, +1): bp.h:40-43,70-73
            wave_sync();
            core.var_phase(need_init);  // initial c_receive_messages(), bp.h:184
            wave_sync();
            if (need_init) {
                it = 0;
                latched = false;
                need_init = false;
            }
            // a fresh frame with max_iter == 0 must fail without iterating; handled by the next round's test
        }

        // ---- one flooding iteration (bp.h:186-188) -------------------------------------------
        core.check_phase(active);
        wave_sync();
        core.var_phase(active);
        wave_sync();
        it += 1;
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

// ------------------------------------------------------------------------------------------
// host-visible launcher table
template <typename T, int MAXD, int L, int ALGO>
static const void *kernel_ptr(bool mc, bool idxlds) {
    if (mc) return idxlds ? (const void *) bp_fused_kernel<T, MAXD, L, ALGO, true, true>
                          : (const void *) bp_fused_kernel<T, MAXD, L, ALGO, true, false>;
    return idxlds ? (const void *) bp_fused_kernel<T, MAXD, L, ALGO, false, true>
                  : (const void *) bp_fused_kernel<T, MAXD, L, ALGO, false, false>;
}

#define ACG_DISPATCH_L(T, MAXD, ALGO, CALL)           \
    switch (L) {                                      \
        case 64: return CALL<T, MAXD, 64, ALGO>;      \
        case 32: return CALL<T, MAXD, 32, ALGO>;      \
        case 16: return CALL<T, MAXD, 16, ALGO>;      \
        default: return nullptr;                      \
    }

using PtrFn = const void *(*) (bool, bool);

template <typename T, int ALGO>
static PtrFn pick_ptr(int maxd, int L) {
    if (maxd <= 8) { ACG_DISPATCH_L(T, 8, ALGO, kernel_ptr) }
    if (maxd <= 16) { ACG_DISPATCH_L(T, 16, ALGO, kernel_ptr) }
    if (maxd <= 32) { ACG_DISPATCH_L(T, 32, ALGO, kernel_ptr) }
    return nullptr;
}

// algo: 0 sum-product, 1 min-sum; f64: 0/1
const void *bp_kernel_ptr(int algo, int f64, int maxd, int L, bool mc, bool idxlds) {
    PtrFn fn = nullptr;
    if (algo == 0) fn = f64 ? pick_ptr<double, 0>(maxd, L) : pick_ptr<float, 0>(maxd, L);
    else fn = f64 ? pick_ptr<double, 1>(maxd, L) : pick_ptr<float, 1>(maxd, L);
    return fn ? fn(mc, idxlds) : nullptr;
}

hipError_t bp_launch(const void *kernel, const BpTables &t, const DecodeArgs &a, int grid, int block, size_t lds,
                     hipStream_t s) {
    BpTables tt = t;
    DecodeArgs aa = a;
    void *args[2] = {&tt, &aa};
    return hipLaunchKernel(kernel, dim3(grid), dim3(block), args, lds, s);
}

// ------------------------------------------------------------------------------------------
// debug / test kernels
__global__ void phi_debug_kernel(const float *x, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = phi_f(x[i]);
}
__global__ void phi_debug_kernel_f64(const double *x, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = phi_f(x[i]);
}

hipError_t phi_debug_launch(const void *x, void *out, int n, int f64, hipStream_t s) {
    if (f64)
        hipLaunchKernelGGL(phi_debug_kernel_f64, dim3((n + 255) / 256), dim3(256), 0, s, (const double *) x, (double *) out, n);
    else
        hipLaunchKernelGGL(phi_debug_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float *) x, (float *) out, n);
    return hipGetLastError();
}

// standalone AWGN generator (utils/channel.h:18-26 with Philox): y[frame][v] natural order
__global__ void awgn_kernel(float *y, int64_t frames, int n, int nwords, int64_t first_frame, uint64_t seed,
                            const uint32_t *cw_packed, int64_t n_cw, float sigma) {
    const int nq = (n + 3) >> 2;
    const int64_t total = frames * nq;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t f = i / nq;
        const int q = (int) (i - f * nq);
        const int64_t gf = first_frame + f;
        uint32_t r[4];
        philox4x32_10((uint32_t) gf, (uint32_t) (gf >> 32), (uint32_t) q, 0u, (uint32_t) seed, (uint32_t) (seed >> 32), r);
        float z[4];
        box_muller(r[0], r[1], z[0], z[1]);
        box_muller(r[2], r[3], z[2], z[3]);
        const uint32_t *cw = cw_packed ? cw_packed + (size_t) (gf % n_cw) * nwords : nullptr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = 4 * q + e;
            if (v < n) {
                const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
                y[(size_t) f * n + v] = (bit ? -1.0f : 1.0f) + sigma * z[e];
            }
        }
    }
}

hipError_t awgn_launch(float *y, int64_t frames, int n, int nwords, int64_t first_frame, uint64_t seed,
                       const uint32_t *cw_packed, int64_t n_cw, float sigma, hipStream_t s) {
    const int64_t total = frames * ((n + 3) >> 2);
    int grid = (int) ((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(awgn_kernel, dim3(grid), dim3(256), 0, s, y, frames, n, nwords, first_frame, seed, cw_packed,
                       n_cw, sigma);
    return hipGetLastError();
}

}  // namespace acg
