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
