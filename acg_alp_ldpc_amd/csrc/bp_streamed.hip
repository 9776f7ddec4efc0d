// Streamed ("HBM") belief-propagation engine — the layout north_star describes, for codes whose
// per-frame message state does not fit in LDS (e.g. the 5000 x 10000 (3,6)-regular stress code of
// BASELINE configs[4]) and as an honest HBM-roofline reference point for the small codes.
//
// Same arithmetic and schedule as the fused engine (bp_core.inc; reference algo/bp.h:183-199), different
// mapping: ONE LANE = ONE FRAME.  A wavefront owns a tile of 64 frames and a private slab of HBM
//     M[e][lane]  (E x 64 words, message of edge e for the 64 frames; in place: v->c before the check
//                  sweep, c->v after it),  LLR[v][lane],  HB[w][lane] (packed hard decisions)
// so every global access is one fully used 256-byte line per wave instruction, the Tanner-graph indices
// are wave-uniform (scalar loads of a plain CSR) and no lane ever needs another lane's data.
// Per frame and iteration the check sweep reads E and writes E words, the variable sweep reads E + n and
// writes E: exactly the (4E + n) * b bytes of the streamed model in SURVEY §8(d) — here the HBM roofline
// is the real bound, and rocprofv3's FETCH_SIZE/WRITE_SIZE can be read against it.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace acg {
#include "bp_core.inc"

template <typename T, int ALGO>
struct StreamPass {
    using B = FpBits<T>;
    using U = typename B::U;
    static constexpr U SIGN = B::SIGN;
    static constexpr U ONE = (U) 1;

    // check c: edges base .. base+D-1 (check-major order) — returns the XOR word (sign parity | syndrome LSB)
    template <int D>
    static __device__ __forceinline__ U check(T *__restrict__ Mp, bool write, bool sonly, T ms_scale) {
        T x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = Mp[j * 64];
        U S = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) S ^= B::to(x[j]);
        if (sonly) return S;
        T out[D];
        if (ALGO == 0) {
            T mag[D], pre[D];
            T s = 0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                mag[j] = B::from(B::to(x[j]) & ~SIGN & ~ONE);  // LSB = hard bit, not magnitude
                pre[j] = s;
                s += mag[j];
            }
            T suf = 0;
#pragma unroll
            for (int j = D - 1; j >= 0; --j) {
                out[j] = Dom<T>::phi(pre[j] + suf);
                suf += mag[j];
            }
        } else {
            T m1 = (T) INFINITY, m2 = (T) INFINITY;
            int am = -1;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const T a = B::from(B::to(x[j]) & ~SIGN & ~ONE);
                const bool lt1 = a < m1, lt2 = a < m2;
                m2 = lt1 ? m1 : (lt2 ? a : m2);
                am = lt1 ? j : am;
                m1 = lt1 ? a : m1;
            }
#pragma unroll
            for (int j = 0; j < D; ++j) out[j] = ms_scale * ((j == am) ? m2 : m1);
        }
        if (write) {
#pragma unroll
            for (int j = 0; j < D; ++j)
                Mp[j * 64] = B::from((B::to(out[j]) & ~SIGN) | ((S ^ B::to(x[j])) & SIGN));
        }
        return S;
    }

    // variable v with edge ids eid[0..D) (wave-uniform); returns the posterior hard bit
    template <int D>
    static __device__ __forceinline__ uint32_t var(T *__restrict__ M, const int *eid, int lane, T llr, bool write) {
        T c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) c[k] = M[(size_t) eid[k] * 64 + lane];
        T pre[D];
        T s = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            pre[k] = s;
            s += c[k];
        }
        const T total = llr + s;
        const U hard = (total <= (T) 0) ? ONE : (U) 0;
        T suf = 0;
        U ob[D];
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
            const T xk = llr + (pre[k] + suf);
            suf += c[k];
            const T ax = B::from(B::to(xk) & ~SIGN);
            const T mg = (ALGO == 0) ? Dom<T>::phi(ax) : ax;
            ob[k] = (B::to(mg) & ~SIGN & ~ONE) | hard | ((xk <= (T) 0) ? SIGN : (U) 0);
        }
        if (write) {
#pragma unroll
            for (int k = 0; k < D; ++k) M[(size_t) eid[k] * 64 + lane] = B::from(ob[k]);
        }
        return (uint32_t) hard;
    }
};

#define ACG_DEG_SWITCH(d, CALL)                                                                                 \
    switch (d) {                                                                                                \
        case 1: CALL(1); break;                                                                                 \
        case 2: CALL(2); break;                                                                                 \
        case 3: CALL(3); break;                                                                                 \
        case 4: CALL(4); break;                                                                                 \
        case 5: CALL(5); break;                                                                                 \
        case 6: CALL(6); break;                                                                                 \
        case 7: CALL(7); break;                                                                                 \
        case 8: CALL(8); break;                                                                                 \
        case 9: CALL(9); break;                                                                                 \
        case 10: CALL(10); break;                                                                               \
        case 11: CALL(11); break;                                                                               \
        case 12: CALL(12); break;                                                                               \
        case 13: CALL(13); break;                                                                               \
        case 14: CALL(14); break;                                                                               \
        case 15: CALL(15); break;                                                                               \
        case 16: CALL(16); break;                                                                               \
        default: break;                                                                                         \
    }

// G nodes of equal degree D at once: all G*D loads are issued before the first use, so a wavefront keeps
// G*D 256-byte lines in flight instead of D (the sweeps are latency-bound otherwise: every store to M
// fences the loads behind it, as far as the compiler can tell).
template <typename T, int ALGO, int D, int G>
__device__ __forceinline__ typename FpBits<T>::U check_group(T *__restrict__ Mp, bool sonly, T ms_scale) {
    using B = FpBits<T>;
    using U = typename B::U;
    T x[G][D];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < D; ++j) x[g][j] = Mp[(size_t) (g * D + j) * 64];
    U acc = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        U S = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) S ^= B::to(x[g][j]);
        acc |= S;
        if (!sonly) {
            T out[D];
            if (ALGO == 0) {
                T mag[D], pre[D];
                T s = 0;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    mag[j] = B::from(B::to(x[g][j]) & ~B::SIGN & ~(U) 1);  // LSB = hard bit, not magnitude
                    pre[j] = s;
                    s += mag[j];
                }
                T suf = 0;
#pragma unroll
                for (int j = D - 1; j >= 0; --j) {
                    out[j] = Dom<T>::phi(pre[j] + suf);
                    suf += mag[j];
                }
            } else {
                T m1 = (T) INFINITY, m2 = (T) INFINITY;
                int am = -1;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const T a = B::from(B::to(x[g][j]) & ~B::SIGN & ~(U) 1);
                    const bool lt1 = a < m1, lt2 = a < m2;
                    m2 = lt1 ? m1 : (lt2 ? a : m2);
                    am = lt1 ? j : am;
                    m1 = lt1 ? a : m1;
                }
#pragma unroll
                for (int j = 0; j < D; ++j) out[j] = ms_scale * ((j == am) ? m2 : m1);
            }
#pragma unroll
            for (int j = 0; j < D; ++j)
                Mp[(size_t) (g * D + j) * 64] = B::from((B::to(out[j]) & ~B::SIGN) | ((S ^ B::to(x[g][j])) & B::SIGN));
        }
    }
    return acc;
}

// G variables of equal degree D; eid = G*D wave-uniform edge ids; returns the G hard bits in bits 0..G-1
template <typename T, int ALGO, int D, int G>
__device__ __forceinline__ uint32_t var_group(T *__restrict__ M, const T *__restrict__ LLRv, const int *eid, int lane) {
    using B = FpBits<T>;
    using U = typename B::U;
    T c[G][D], llr[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        llr[g] = LLRv[(size_t) g * 64];
#pragma unroll
        for (int k = 0; k < D; ++k) c[g][k] = M[(size_t) eid[g * D + k] * 64 + lane];
    }
    uint32_t hb = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        T pre[D];
        T s = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            pre[k] = s;
            s += c[g][k];
        }
        const T total = llr[g] + s;
        const U hard = (total <= (T) 0) ? (U) 1 : (U) 0;
        hb |= (uint32_t) hard << g;
        T suf = 0;
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
            const T xk = llr[g] + (pre[k] + suf);
            suf += c[g][k];
            const T ax = B::from(B::to(xk) & ~B::SIGN);
            const T mg = (ALGO == 0) ? Dom<T>::phi(ax) : ax;
            const U ob = (B::to(mg) & ~B::SIGN & ~(U) 1) | hard | ((xk <= (T) 0) ? B::SIGN : (U) 0);
            M[(size_t) eid[g * D + k] * 64 + lane] = B::from(ob);
        }
    }
    return hb;
}

#define ACG_DEG8_SWITCH(d, CALL)                                                                                \
    switch (d) {                                                                                                \
        case 1: CALL(1); break;                                                                                 \
        case 2: CALL(2); break;                                                                                 \
        case 3: CALL(3); break;                                                                                 \
        case 4: CALL(4); break;                                                                                 \
        case 5: CALL(5); break;                                                                                 \
        case 6: CALL(6); break;                                                                                 \
        case 7: CALL(7); break;                                                                                 \
        case 8: CALL(8); break;                                                                                 \
        default: break;                                                                                         \
    }

// One workgroup (W wavefronts) per 64-frame tile: the waves split the checks / variables of a sweep
// (groups of 4 nodes, interleaved), `__syncthreads()` separates the sweeps.
template <typename T, int ALGO>
__global__ void __launch_bounds__(512) bp_streamed_kernel(const StreamTables t, const DecodeArgs a, uint32_t *ws) {
    using P = StreamPass<T, ALGO>;
    using B = FpBits<T>;
    using U = typename B::U;
    __shared__ uint32_t bad_lds[8][64];
    constexpr int G = (sizeof(T) == 8) ? 2 : 4;  // nodes per group (register budget: 256 VGPRs at 2 waves/SIMD)
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int W = blockDim.x >> 6;
    uint32_t *base = ws + (size_t) blockIdx.x * t.ws_words_per_wave;
    T *M = reinterpret_cast<T *>(base);
    T *LLR = M + (size_t) t.E * 64;
    uint32_t *HB = reinterpret_cast<uint32_t *>(LLR + (size_t) t.n * 64);
    const T ms_scale = (T) a.ms_scale;
    const int64_t n_tiles = (a.frames + 63) / 64;
    const int n_task = (t.n + 31) / 32;  // variable tasks of 32 consecutive variables = one output word

    __shared__ unsigned long long tile_lds;
    for (;;) {
        // dynamic tile hand-out: early exit makes tiles finish after very different numbers of sweeps
        __syncthreads();
        if (threadIdx.x == 0) tile_lds = atomicAdd(a.work_counter, 1ull);
        __syncthreads();
        const int64_t tile = (int64_t) tile_lds;
        if (tile >= n_tiles) break;
        const int64_t frame = tile * 64 + lane;
        const bool valid = frame < a.frames;
        // ---- channel LLRs (channel.h:14-16) and the initial v->c sweep (bp.h:184: mailboxes are zero) ----
        for (int v = w; v < t.n; v += W) {
            T llr = (T) 0;
            if (valid) {
                if (a.y_is_f64) llr = (T) (2 * reinterpret_cast<const double *>(a.y)[(size_t) frame * t.n + v] / a.var * Dom<T>::scale);
                else llr = (T) ((double) reinterpret_cast<const float *>(a.y)[(size_t) frame * t.n + v] * (a.inv_var2 * Dom<T>::scale));
            }
            LLR[(size_t) v * 64 + lane] = llr;
            const T ax = B::from(B::to(llr) & ~B::SIGN);
            const T mg = (ALGO == 0) ? Dom<T>::phi(ax) : ax;
            const U hard = (llr <= (T) 0) ? (U) 1 : (U) 0;
            const U ob = (B::to(mg) & ~B::SIGN & ~(U) 1) | hard | ((llr <= (T) 0) ? B::SIGN : (U) 0);
            const int b = sload(t.col_ptr, v), e = sload(t.col_ptr, v + 1);
            for (int k = b; k < e; ++k) M[(size_t) sload(t.col_edge, k) * 64 + lane] = B::from(ob);
        }
        __syncthreads();
        bool latched = false;
        int lat_it = 0;
        for (int it = 0;; ++it) {
            // ---- check sweep (bp.h:171-181); its XOR also yields the syndrome of the previous estimate ----
            const bool sonly = (it >= a.max_iter);
            U acc = 0;
            for (int c0 = G * w; c0 < t.m; c0 += G * W) {
                const int b0 = sload(t.row_ptr, c0);
                const int nc = min(G, t.m - c0);
                const int d0 = sload(t.row_ptr, c0 + 1) - b0;
                bool uni = (nc == G) && d0 >= 1 && d0 <= 8;
                for (int g = 2; uni && g <= G; ++g) uni = (sload(t.row_ptr, c0 + g) - b0 == g * d0);
                T *Mp = M + (size_t) b0 * 64 + lane;
                if (uni) {
#define ACG_CALL(D) acc |= check_group<T, ALGO, D, G>(Mp, sonly, ms_scale)
                    ACG_DEG8_SWITCH(d0, ACG_CALL)
#undef ACG_CALL
                } else {
                    for (int c = c0; c < c0 + nc; ++c) {
                        const int b = sload(t.row_ptr, c);
                        const int d = sload(t.row_ptr, c + 1) - b;
                        T *Mq = M + (size_t) b * 64 + lane;
#define ACG_CALL(D) acc |= P::template check<D>(Mq, true, sonly, ms_scale)
                        ACG_DEG_SWITCH(d, ACG_CALL)
#undef ACG_CALL
                    }
                }
            }
            bad_lds[w][lane] = (uint32_t) (acc & (U) 1);
            __syncthreads();
            uint32_t badw = 0;
            for (int i = 0; i < W; ++i) badw |= bad_lds[i][lane];
            if (valid && !latched && it > 0 && !badw) {  // bp.h:195-196
                latched = true;
                lat_it = it;
            }
            const bool done = latched || !valid;
            if (sonly) break;
            if (a.dbg_c2v && tile == 0 && it == a.max_iter - 1) {  // diagnostics: c->v words of the last sweep
                T *dc = reinterpret_cast<T *>(a.dbg_c2v);
                for (int e = w; e < t.E; e += W) dc[(size_t) e * 64 + lane] = M[(size_t) e * 64 + lane];
            }
            if (a.early_exit && __ballot(!done) == 0ull) break;  // identical in every wave of the block
            // ---- variable sweep (bp.h:160-169) + posterior hard decisions (bp.h:191-193) ----
            for (int task = w; task < n_task; task += W) {
                uint32_t word = 0;
                const int v_end = min(t.n, task * 32 + 32);
                for (int v0 = task * 32; v0 < v_end; v0 += G) {
                    const int nv = min(G, v_end - v0);
                    const int b0 = sload(t.col_ptr, v0);
                    const int d0 = sload(t.col_ptr, v0 + 1) - b0;
                    bool uni = (nv == G) && d0 >= 1 && d0 <= 8;
                    for (int g = 2; uni && g <= G; ++g) uni = (sload(t.col_ptr, v0 + g) - b0 == g * d0);
                    if (uni) {
                        int eid[8 * G];
#pragma unroll
                        for (int k = 0; k < 8 * G; ++k)
                            if (k < G * d0) eid[k] = sload(t.col_edge, b0 + k);
                        uint32_t hb = 0;
#define ACG_CALL(D) hb = var_group<T, ALGO, D, G>(M, LLR + (size_t) v0 * 64 + lane, eid, lane)
                        ACG_DEG8_SWITCH(d0, ACG_CALL)
#undef ACG_CALL
                        word |= hb << (v0 & 31);
                    } else {
                        for (int v = v0; v < v0 + nv; ++v) {
                            const int b = sload(t.col_ptr, v);
                            const int d = sload(t.col_ptr, v + 1) - b;
                            const T llr = LLR[(size_t) v * 64 + lane];
                            uint32_t hard = (llr <= (T) 0) ? 1u : 0u;  // isolated variable: estimate() == channel LLR
                            int eid[16];
#pragma unroll
                            for (int k = 0; k < 16; ++k)
                                if (k < d) eid[k] = sload(t.col_edge, b + k);
#define ACG_CALL(D) hard = P::template var<D>(M, eid, lane, llr, true)
                            ACG_DEG_SWITCH(d, ACG_CALL)
#undef ACG_CALL
                            word |= hard << (v & 31);
                        }
                    }
                }
                if (!latched) HB[(size_t) task * 64 + lane] = word;  // frozen once the frame has converged
            }
            __syncthreads();
            if (a.dbg_v2c && tile == 0 && it == a.max_iter - 1) {  // diagnostics: v->c words + posteriors
                T *dv = reinterpret_cast<T *>(a.dbg_v2c);
                T *dc = reinterpret_cast<T *>(a.dbg_c2v);
                T *dp = reinterpret_cast<T *>(a.dbg_post);
                for (int e = w; e < t.E; e += W) dv[(size_t) e * 64 + lane] = M[(size_t) e * 64 + lane];
                for (int v = w; v < t.n; v += W) {
                    T sum = 0;  // estimate() = llr + sum of the c->v mailbox (bp.h:85-90), from the dumped c->v words
                    for (int k = sload(t.col_ptr, v); k < sload(t.col_ptr, v + 1); ++k)
                        sum += dc[(size_t) sload(t.col_edge, k) * 64 + lane];
                    dp[(size_t) v * 64 + lane] = LLR[(size_t) v * 64 + lane] + sum;
                }
                __syncthreads();
            }
        }
        // ---- outputs ----
        if (valid) {
            if (w == 0) {
                if (a.out_ok) a.out_ok[frame] = latched ? 1 : 0;
                if (a.out_iters) a.out_iters[frame] = latched ? lat_it : a.max_iter;
            }
            if (a.out_bits)
                for (int k = w; k < t.nwords; k += W)
                    a.out_bits[(size_t) frame * t.nwords + k] = latched ? HB[(size_t) k * 64 + lane] : 0u;  // bp.h:198
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// LDS-DMA ring variant (fp32).  Same slab layout, same arithmetic (StreamPass), same outputs; what changes is how the
// 256-byte message lines travel.  bp_streamed_kernel holds every line in flight in VGPRs (253 of them, two wavefronts per
// SIMD) and alternates load burst / arithmetic / store burst, so the memory pipe idles while a wavefront computes.  Here
// each wavefront owns a ring of RING_SLOTS slots of LDS (4 KiB each) and runs a software pipeline over its tasks:
//     issue the LDS-DMA loads of task i+3  ->  wait until task i has landed  ->  read it from LDS, compute, store
// The loads (global_load_lds_dwordx4: 16 lanes fetch one line, so one instruction lands FOUR lines, gathered from four
// different places in the variable sweep) need no VGPR destination, so three tasks (12 KiB per wavefront, 96 KiB per CU)
// are always in flight behind the arithmetic, at ~70 VGPRs.  vmcnt retires vector-memory operations in issue order and
// counts loads, stores and LDS-DMA alike, so "task i has landed" is `s_waitcnt vmcnt(N)` with N = the operations issued
// behind its loads: the stores of the tasks computed meanwhile stay in flight and never enter the wait.  N comes from
// the task table (the schedule per wavefront is static) and counts only operations that are certainly issued —
// under-counting waits a little longer, over-counting would read a slot before it has landed.
// The DMA is inline assembly (M0 = LDS destination) so the compiler neither tracks it nor drains it early; a wavefront
// reads only slots it filled itself, for which its own counted vmcnt is the ordering the hardware asks for.
template <bool NT>
__device__ __forceinline__ void ring_dma16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
}
template <bool NT>
__device__ __forceinline__ void ring_store(float *p, float v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// wait until at most n (rounded down to a multiple of 4) vector-memory operations of this wavefront are outstanding
__device__ __forceinline__ void ring_wait_vmcnt(int n) {
    switch (n >> 2) {
#define ACG_W(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * k) : "memory"); break;
        ACG_W(0) ACG_W(1) ACG_W(2) ACG_W(3) ACG_W(4) ACG_W(5) ACG_W(6) ACG_W(7)
        ACG_W(8) ACG_W(9) ACG_W(10) ACG_W(11) ACG_W(12) ACG_W(13) ACG_W(14)
#undef ACG_W
        default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    }
}

template <int ALGO, bool NT>
struct RingPass {
    using T = float;
    using B = FpBits<float>;
    using U = uint32_t;
    // check with its D incoming words in LDS (in[j*64]) -> c->v words to HBM (out[j*64]); returns the XOR word
    template <int D>
    static __device__ __forceinline__ U check(const T *__restrict__ in, T *__restrict__ out, bool write, T ms_scale) {
        T x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = in[j * 64];
        U S = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) S ^= B::to(x[j]);
        if (!write) return S;
        T o[D];
        if (ALGO == 0) {
            T mag[D], pre[D];
            T s = 0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                mag[j] = B::from(B::to(x[j]) & ~B::SIGN & ~(U) 1);  // LSB = hard bit, not magnitude
                pre[j] = s;
                s += mag[j];
            }
            T suf = 0;
#pragma unroll
            for (int j = D - 1; j >= 0; --j) {
                o[j] = Dom<T>::phi(pre[j] + suf);
                suf += mag[j];
            }
        } else {
            T m1 = (T) INFINITY, m2 = (T) INFINITY;
            int am = -1;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const T a = B::from(B::to(x[j]) & ~B::SIGN & ~(U) 1);
                const bool lt1 = a < m1, lt2 = a < m2;
                m2 = lt1 ? m1 : (lt2 ? a : m2);
                am = lt1 ? j : am;
                m1 = lt1 ? a : m1;
            }
#pragma unroll
            for (int j = 0; j < D; ++j) o[j] = ms_scale * ((j == am) ? m2 : m1);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) ring_store<NT>(out + j * 64, B::from((B::to(o[j]) & ~B::SIGN) | ((S ^ B::to(x[j])) & B::SIGN)));
        return S;
    }

    // variable with its D incoming words in LDS (in[k*64]) -> v->c words to M[eid[k]*64 + lane]; returns the hard bit
    template <int D>
    static __device__ __forceinline__ uint32_t var(const T *__restrict__ in, T *__restrict__ M, const int *eid, int lane, T llr) {
        T c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) c[k] = in[k * 64];
        T pre[D];
        T s = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            pre[k] = s;
            s += c[k];
        }
        const T total = llr + s;
        const U hard = (total <= (T) 0) ? (U) 1 : (U) 0;
        T suf = 0;
        U ob[D];
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
            const T xk = llr + (pre[k] + suf);
            suf += c[k];
            const T ax = B::from(B::to(xk) & ~B::SIGN);
            const T mg = (ALGO == 0) ? Dom<T>::phi(ax) : ax;
            ob[k] = (B::to(mg) & ~B::SIGN & ~(U) 1) | hard | ((xk <= (T) 0) ? B::SIGN : (U) 0);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) ring_store<NT>(M + (size_t) eid[k] * 64 + lane, B::from(ob[k]));
        return (uint32_t) hard;
    }
};

#define ACG_DEG12_SWITCH(d, CALL)                                                                               \
    switch (d) {                                                                                                \
        case 1: CALL(1); break;                                                                                 \
        case 2: CALL(2); break;                                                                                 \
        case 3: CALL(3); break;                                                                                 \
        case 4: CALL(4); break;                                                                                 \
        case 5: CALL(5); break;                                                                                 \
        case 6: CALL(6); break;                                                                                 \
        case 7: CALL(7); break;                                                                                 \
        case 8: CALL(8); break;                                                                                 \
        case 9: CALL(9); break;                                                                                 \
        case 10: CALL(10); break;                                                                               \
        case 11: CALL(11); break;                                                                               \
        case 12: CALL(12); break;                                                                               \
        default: break;                                                                                         \
    }

// NT: non-temporal loads and stores for slabs that cannot stay in the 256 MiB Infinity Cache between two sweeps anyway
// DBG (tests only, acg_ldpc_debug_bp_trace): the message slab of the first tile is copied out after the check sweep and after the
// variable sweep of the last iteration (the messages live in memory, so the trace is a copy of M), with the posteriors
// LLR + sum of the dumped c->v words (bp.h:85-90); the product instances are compiled with DBG = false.
template <int ALGO, bool NT, bool DBG = false>
__global__ void __launch_bounds__(RING_WAVES * 64, (RING_LDS_BYTES <= 32 * 1024 ? 5 : (RING_LDS_BYTES <= 40 * 1024 ? 4 : (RING_LDS_BYTES <= 52 * 1024 ? 3 : 2)))) bp_streamed_ring_kernel(const StreamTables t, const DecodeArgs a, uint32_t *ws) {
    using T = float;
    using B = FpBits<float>;
    using U = uint32_t;
    using P = RingPass<ALGO, NT>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring_lds[];
    __shared__ uint32_t bad_lds[RING_WAVES][64];
    __shared__ unsigned long long tile_lds;
    constexpr int W = RING_WAVES, R = RING_SLOTS, SLOT_BYTES = RING_SLOT_LINES * 256;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *base = ws + (size_t) blockIdx.x * t.ws_words_per_wave;
    T *M = reinterpret_cast<T *>(base);
    T *LLR = M + (size_t) t.E * 64;
    uint8_t *HB = reinterpret_cast<uint8_t *>(LLR + (size_t) t.n * 64);   // [variable task][64] bytes: the task's hard decisions
    const unsigned char *Mb = reinterpret_cast<const unsigned char *>(M), *LLRb = reinterpret_cast<const unsigned char *>(LLR);
    const T ms_scale = (T) a.ms_scale;
    const int64_t n_tiles = (a.frames + 63) / 64;
    unsigned char *my_ring = ring_lds + (size_t) w * R * SLOT_BYTES;
    const uint32_t ring_addr = (uint32_t) (uintptr_t) my_ring;   // LDS byte address of this wavefront's ring (wave-uniform)
    const int q16 = lane >> 4, l16 = lane & 15;
    // tasks of this wavefront: w, w + W, ...
    const int n_ct = (t.n_ctask - w + W - 1) / W, n_vt = (t.n_vtask - w + W - 1) / W;

    auto issue_check = [&](int i) {  // loads of my i-th check task into slot i % R
        const int ti = w + i * W;
        const int first_line = sload(t.ctask, 4 * ti + 2), nl = sload(t.ctask, 4 * ti + 3) & 0xFF;
        const uint32_t dst = ring_addr + (uint32_t) ((i % R) * SLOT_BYTES);
        const unsigned char *src = Mb + (size_t) first_line * 256 + (size_t) lane * 16;   // the task's lines are contiguous
#pragma unroll
        for (int j = 0; j < RING_SLOT_LINES / 4; ++j)
            if (4 * j < nl)
                if (lane < 16 * (nl - 4 * j)) ring_dma16<NT>(src + j * 1024, dst + j * 1024);
    };
    auto issue_var = [&](int i) {  // edge lines gathered four per instruction + the LLR lines of the task's variables
        const int ti = w + i * W;
        const int v0 = sload(t.vtask, 4 * ti), nv = sload(t.vtask, 4 * ti + 1), cp0 = sload(t.vtask, 4 * ti + 2);
        const int nel = sload(t.vtask, 4 * ti + 3) & 0xFF;
        const uint32_t dst = ring_addr + (uint32_t) ((i % R) * SLOT_BYTES);
#pragma unroll
        for (int j = 0; j < RING_VAR_EDGE_LINES / 4; ++j)
            if (4 * j < nel) {
                const int e0 = sload(t.col_edge, cp0 + 4 * j), e1 = sload(t.col_edge, cp0 + 4 * j + 1);
                const int e2 = sload(t.col_edge, cp0 + 4 * j + 2), e3 = sload(t.col_edge, cp0 + 4 * j + 3);
                const int e = (q16 == 0) ? e0 : ((q16 == 1) ? e1 : ((q16 == 2) ? e2 : e3));
                if (4 * j + q16 < nel) ring_dma16<NT>(Mb + (size_t) e * 256 + (size_t) l16 * 16, dst + j * 1024);
            }
        if (q16 < nv) ring_dma16<NT>(LLRb + (size_t) v0 * 256 + (size_t) lane * 16, dst + RING_VAR_EDGE_LINES * 256);
    };

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) tile_lds = atomicAdd(a.work_counter, 1ull);
        __syncthreads();
        const int64_t tile = (int64_t) tile_lds;
        if (tile >= n_tiles) break;
        const int64_t frame = tile * 64 + lane;
        const bool valid = frame < a.frames;
        // ---- channel LLRs (channel.h:14-16) and the initial v->c sweep (bp.h:184: mailboxes are zero) ----
        for (int v = w; v < t.n; v += W) {
            T llr = (T) 0;
            if (valid) {
                if (a.y_is_f64) llr = (T) (2 * reinterpret_cast<const double *>(a.y)[(size_t) frame * t.n + v] / a.var * Dom<T>::scale);
                else llr = (T) ((double) reinterpret_cast<const float *>(a.y)[(size_t) frame * t.n + v] * (a.inv_var2 * Dom<T>::scale));
            }
            LLR[(size_t) v * 64 + lane] = llr;
            const T ax = B::from(B::to(llr) & ~B::SIGN);
            const T mg = (ALGO == 0) ? Dom<T>::phi(ax) : ax;
            const U hard = (llr <= (T) 0) ? (U) 1 : (U) 0;
            const U ob = (B::to(mg) & ~B::SIGN & ~(U) 1) | hard | ((llr <= (T) 0) ? B::SIGN : (U) 0);
            const int b = sload(t.col_ptr, v), e = sload(t.col_ptr, v + 1);
            for (int k = b; k < e; ++k) M[(size_t) sload(t.col_edge, k) * 64 + lane] = B::from(ob);
        }
        __syncthreads();
        bool latched = false;
        int lat_it = 0;
        for (int it = 0;; ++it) {
            // ---- check sweep (bp.h:171-181); its XOR also yields the syndrome of the previous estimate ----
            const bool sonly = (it >= a.max_iter);
            U acc = 0;
            for (int i = 0; i < R - 1 && i < n_ct; ++i) issue_check(i);
            for (int i = 0; i < n_ct; ++i) {
                if (i + R - 1 < n_ct) issue_check(i + R - 1);
                const int ti = w + i * W;
                const int c0 = sload(t.ctask, 4 * ti), nc = sload(t.ctask, 4 * ti + 1), first_line = sload(t.ctask, 4 * ti + 2);
                const int pk = sload(t.ctask, 4 * ti + 3);
                ring_wait_vmcnt(sonly ? ((pk >> 16) & 0xFF) : ((pk >> 8) & 0xFF));
                const T *slot = reinterpret_cast<const T *>(my_ring + (size_t) (i % R) * SLOT_BYTES) + lane;
                T *outp = M + (size_t) first_line * 64 + lane;
                int b0 = sload(t.row_ptr, c0);
                for (int c = 0; c < nc; ++c) {
                    const int b1 = sload(t.row_ptr, c0 + c + 1);
                    const int d = b1 - b0, o = b0 - first_line;
#define ACG_CALL(D) acc |= P::template check<D>(slot + o * 64, outp + (size_t) o * 64, !sonly, ms_scale)
                    ACG_DEG_SWITCH(d, ACG_CALL)
#undef ACG_CALL
                    b0 = b1;
                }
            }
            bad_lds[w][lane] = (uint32_t) (acc & (U) 1);
            __syncthreads();
            uint32_t badw = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) badw |= bad_lds[i][lane];
            if (valid && !latched && it > 0 && !badw) {  // bp.h:195-196
                latched = true;
                lat_it = it;
            }
            const bool done = latched || !valid;
            if (sonly) break;
            if (DBG && a.dbg_c2v && tile == 0 && it == a.max_iter - 1) {  // c->v words of the last sweep (the barrier above drained the stores)
                T *dc = reinterpret_cast<T *>(a.dbg_c2v);
                for (int e = w; e < t.E; e += W) dc[(size_t) e * 64 + lane] = M[(size_t) e * 64 + lane];
            }
            if (a.early_exit && __ballot(!done) == 0ull) break;  // identical in every wave of the block
            // ---- variable sweep (bp.h:160-169) + posterior hard decisions (bp.h:191-193) ----
            for (int i = 0; i < R - 1 && i < n_vt; ++i) issue_var(i);
            for (int i = 0; i < n_vt; ++i) {
                if (i + R - 1 < n_vt) issue_var(i + R - 1);
                const int ti = w + i * W;
                const int nv = sload(t.vtask, 4 * ti + 1), cp0 = sload(t.vtask, 4 * ti + 2);
                const int pk = sload(t.vtask, 4 * ti + 3);
                ring_wait_vmcnt((pk >> 8) & 0xFF);
                const T *slot = reinterpret_cast<const T *>(my_ring + (size_t) (i % R) * SLOT_BYTES) + lane;
                uint32_t bits = 0;
                int b0 = cp0;
                for (int vi = 0; vi < nv; ++vi) {
                    const int b1 = sload(t.col_ptr, sload(t.vtask, 4 * ti) + vi + 1);
                    const int d = b1 - b0;
                    const T llr = slot[(RING_VAR_EDGE_LINES + vi) * 64];
                    uint32_t hard = (llr <= (T) 0) ? 1u : 0u;  // isolated variable: estimate() == channel LLR
                    int eid[RING_MAX_VDEG];
#pragma unroll
                    for (int k = 0; k < RING_MAX_VDEG; ++k)
                        if (k < d) eid[k] = sload(t.col_edge, b0 + k);
#define ACG_CALL(D) hard = P::template var<D>(slot + (b0 - cp0) * 64, M, eid, lane, llr)
                    ACG_DEG12_SWITCH(d, ACG_CALL)
#undef ACG_CALL
                    bits |= hard << vi;
                    b0 = b1;
                }
                if (!latched) HB[(size_t) ti * 64 + lane] = (uint8_t) bits;  // frozen once the frame has converged
            }
            __syncthreads();
            if (DBG && a.dbg_v2c && tile == 0 && it == a.max_iter - 1) {  // v->c words + posteriors of the last sweep
                T *dv = reinterpret_cast<T *>(a.dbg_v2c);
                T *dc = reinterpret_cast<T *>(a.dbg_c2v);
                T *dp = reinterpret_cast<T *>(a.dbg_post);
                for (int e = w; e < t.E; e += W) dv[(size_t) e * 64 + lane] = M[(size_t) e * 64 + lane];
                for (int v = w; v < t.n; v += W) {
                    T sum = 0;  // estimate() = llr + sum of the c->v mailbox (bp.h:85-90), from the dumped c->v words
                    for (int k = sload(t.col_ptr, v); k < sload(t.col_ptr, v + 1); ++k)
                        sum += dc[(size_t) sload(t.col_edge, k) * 64 + lane];
                    dp[(size_t) v * 64 + lane] = LLR[(size_t) v * 64 + lane] + sum;
                }
                __syncthreads();
            }
        }
        // ---- outputs: words assembled from the per-task bytes ----
        if (valid) {
            if (w == 0) {
                if (a.out_ok) a.out_ok[frame] = latched ? 1 : 0;
                if (a.out_iters) a.out_iters[frame] = latched ? lat_it : a.max_iter;
            }
        }
        if (a.out_bits) {
            for (int k = w; k < t.nwords; k += W) {
                uint32_t word = 0;
                if (latched) {  // bp.h:198: a failed frame returns the empty word
                    // variable tasks are runs of consecutive variables: walk the tasks overlapping [32k, 32k + 32)
                    int ti = sload(t.vtask_of_word, k);
                    for (;;) {
                        if (ti >= t.n_vtask) break;
                        const int v0 = sload(t.vtask, 4 * ti), nv = sload(t.vtask, 4 * ti + 1);
                        if (v0 >= 32 * k + 32) break;
                        const uint32_t by = HB[(size_t) ti * 64 + lane];
                        for (int vi = 0; vi < nv; ++vi) {
                            const int v = v0 + vi;
                            if (v >= 32 * k && v < 32 * k + 32) word |= ((by >> vi) & 1u) << (v & 31);
                        }
                        ++ti;
                    }
                }
                if (valid) a.out_bits[(size_t) frame * t.nwords + k] = word;
            }
        }
    }
}

const void *bp_streamed_ring_ptr(int algo, bool nt) {
    if (nt) return algo == 0 ? (const void *) bp_streamed_ring_kernel<0, true> : (const void *) bp_streamed_ring_kernel<1, true>;
    return algo == 0 ? (const void *) bp_streamed_ring_kernel<0, false> : (const void *) bp_streamed_ring_kernel<1, false>;
}

// debug instance (sum-product, default cache policy): the arithmetic and the counted waits are those of the product kernel
const void *bp_streamed_ring_ptr_dbg() { return (const void *) bp_streamed_ring_kernel<0, false, true>; }

hipError_t bp_streamed_ring_launch(const void *kernel, const StreamTables &t, const DecodeArgs &a, uint32_t *ws, int grid, hipStream_t s) {
    StreamTables tt = t;
    DecodeArgs aa = a;
    void *args[3] = {&tt, &aa, &ws};
    return hipLaunchKernel(kernel, dim3(grid), dim3(RING_WAVES * 64), args, (size_t) RING_LDS_BYTES, s);
}

const void *bp_streamed_ptr(int algo, int f64) {
    if (algo == 0) return f64 ? (const void *) bp_streamed_kernel<double, 0> : (const void *) bp_streamed_kernel<float, 0>;
    return f64 ? (const void *) bp_streamed_kernel<double, 1> : (const void *) bp_streamed_kernel<float, 1>;
}

hipError_t bp_streamed_launch(const void *kernel, const StreamTables &t, const DecodeArgs &a, uint32_t *ws, int grid,
                              int block, hipStream_t s) {
    StreamTables tt = t;
    DecodeArgs aa = a;
    void *args[3] = {&tt, &aa, &ws};
    return hipLaunchKernel(kernel, dim3(grid), dim3(block), args, 0, s);
}

}  // namespace acg
