// Layered (row-block sequential) normalised min-sum for gfx950 — SURVEY §8(f) N4, BP side.
//
// NOT the reference's schedule: algo/bp.h:183-199 floods (all checks, then all variables).  Here the checks are cut into
// LAYERS — sets of checks of one degree that share no variable; for the quasi-cyclic matrices of the reference (H05 /
// optimalH: 8 x 14 arrays of 20 x 20 cyclic-shift blocks, optimize_H.cpp:27-63) the layers are the 8 block rows — and the
// posteriors are updated in place after every layer, so the second layer already sees what the first one learnt.  One
// sweep over all layers does the work of about two flooding sweeps; parity with the reference is FER-level only (and
// min-sum itself is build-added, SURVEY D2: parity unpinned).
//
// Mapping.  G lanes of a wavefront (G = 16, 20, 32 or 64; H05: G = Z = 20, three frames per wavefront) own one frame; lane l
// handles check l of the current layer.  Per frame LDS holds the posteriors P[v] (natural variable order) and the
// check-to-variable messages R[layer][edge j][lane] — the latter are only ever touched by the lane that owns the check, at
// immediate offsets from one per-layer address, so they need no index at all.  Which posterior cell edge j of check (layer,
// lane) reads comes from a 16-bit byte-offset table shared by the workgroup; for a quasi-cyclic H that table is never stored
// in memory: the kernel builds it when it starts from the (block column, shift) list of the block rows —
// variable = C_j * Z + (k + s_j) mod Z, optimize_H.cpp:41 — which is all the graph description this engine is given.
//
// One layer step, per lane (degree D compile time, dispatched wave-uniformly):
//     pos_j <- table;  P_j <- LDS;  R_j <- LDS           (3 D LDS reads)
//     Q_j = P_j - R_j;   two smallest |Q_j| by v_med3 / v_min;   sign product by XOR
//     R'_j = scale * (|Q_j| == min1 ? min2 : min1) with sign;   P'_j = Q_j + R'_j       (2 D LDS writes)
// Stopping rule (exact, no separate syndrome pass): a layer step is QUIET when every check of the layer is satisfied by the
// signs of the posteriors it READ and none of the posteriors it WROTE changed sign.  An iteration (one round over all layers)
// in which every step was quiet has left the hard decisions untouched and has seen them satisfy every check: H x = 0, the
// frame stops there (early exit) or its output is latched (fixed work).  A frame that runs out of iterations without a quiet
// round gets one explicit syndrome pass over its final posteriors.  Results are deterministic: every frame starts at layer 0.
// Frames are handed out dynamically as in the flooding kernel (bp_core.inc), at round boundaries.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"

namespace acg {
#include "bp_core.inc"   // Dom<float>::phi (the fp32 phi of the flooding kernels, log2(e)-scaled domain) for the sum-product variant

typedef const int32_t __attribute__((address_space(4))) *lsconst_i32;
__device__ __forceinline__ int lsload(const int32_t *p, int i) { return ((lsconst_i32) (p))[i]; }

__device__ __forceinline__ void lwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A layer step is software-pipelined over the layers (occupancy is bounded by LDS — 4.6 KB per frame, 10 wavefronts per CU — so
// registers are plentiful and latency is what there is to hide):
//   fetch<D>   (issued during the PREVIOUS step) positions and old messages of this lane's check: they depend on the frame's
//              previous iteration only, never on the layer before
//   front<D>   posterior reads through the prefetched positions, Q = P - R
//   ... the next layer's fetch is issued here, behind the posterior reads and ahead of the arithmetic ...
//   back<D>    two minima, signs, R', P' — all 2 D stores under ONE predicate at the end (a predicate per edge costs an
//              EXEC save / branch / restore each: 6 scalar instructions per edge in the first version, by the counters)
constexpr int LMAXD = 8;
constexpr float LAYERED_SATURATION = 59968.0f;   // message of a one-variable check ("certainly 0"); representable in fp16
// RT: storage type of the check-to-variable messages — float, or _Float16 (precision = ACG_LDPC_PREC_F16: 2.9 KB of LDS per
// frame instead of 4.6, 16 wavefronts per CU instead of 10; the posteriors stay fp32 and always see the ROUNDED message, so the
// iteration stays self-consistent: Q = P - R subtracts exactly what was added)
template <int D, int G, typename RT>
__device__ __forceinline__ void layer_fetch(const RT *__restrict__ Rl, const uint16_t *__restrict__ Tl, int (&pos)[LMAXD], float (&r)[LMAXD]) {
#pragma unroll
    for (int j = 0; j < D; ++j) pos[j] = Tl[j * G];
#pragma unroll
    for (int j = 0; j < D; ++j) r[j] = (float) Rl[j * G];
}
template <int D>
__device__ __forceinline__ void layer_front(unsigned char *__restrict__ Pb, const int (&pos)[LMAXD], const float (&r)[LMAXD],
                                            float *(&addr)[LMAXD], float (&p)[LMAXD], float (&q)[LMAXD]) {
#pragma unroll
    for (int j = 0; j < D; ++j) addr[j] = reinterpret_cast<float *>(Pb + pos[j]);
#pragma unroll
    for (int j = 0; j < D; ++j) p[j] = *addr[j];
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] = p[j] - r[j];
}
// -> sign bit set <=> the step was not quiet for this lane
template <int D, int G, typename RT>
__device__ __forceinline__ uint32_t layer_back(RT *__restrict__ Rl, float *const (&addr)[LMAXD], const float (&p)[LMAXD], const float (&q)[LMAXD],
                                               const bool store, const float scale) {
    uint32_t S = 0, noisy = 0;
    float a[D];
    float m1 = INFINITY, m2 = INFINITY;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        noisy ^= __float_as_uint(p[j]);                     // parity of the hard decisions this check sees
        S ^= __float_as_uint(q[j]);
        a[j] = __uint_as_float(__float_as_uint(q[j]) & 0x7FFFFFFFu);
        m2 = __builtin_amdgcn_fmed3f(a[j], m1, m2);
        m1 = __builtin_fminf(m1, a[j]);
    }
    // the two minima are scaled once per check (made opaque: the compiler would otherwise turn select(s*m2, s*m1) back into
    // s * select(m2, m1), one multiply per edge)
    // (RT = _Float16: rounded to the storage type here, once per check, so that P' adds exactly what the next iteration subtracts)
    uint32_t m1s = __float_as_uint((float) (RT) (scale * m1)) & 0x7FFFFFFFu, m2s = __float_as_uint((float) (RT) (scale * m2)) & 0x7FFFFFFFu;
    // a check with ONE variable pins it to 0: the minimum over its (empty) set of other edges is +inf, and an infinite message
    // would turn the next Q = P - R into inf - inf.  It saturates at a value far above any real message instead (exact in fp16).
    if constexpr (D == 1) m2s = __float_as_uint((float) (RT) LAYERED_SATURATION);
    asm volatile("" : "+v"(m1s), "+v"(m2s));
    float rn[D], pn[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const uint32_t mag = (a[j] == m1) ? m2s : m1s;      // a tie makes m2 == m1: either answer is the same
        rn[j] = __uint_as_float(mag | ((S ^ __float_as_uint(q[j])) & 0x80000000u));
        pn[j] = q[j] + rn[j];
        noisy |= __float_as_uint(pn[j]) ^ __float_as_uint(p[j]);   // a hard decision flipped
    }
    if (store) {
#pragma unroll
        for (int j = 0; j < D; ++j) *addr[j] = pn[j];
#pragma unroll
        for (int j = 0; j < D; ++j) Rl[j * G] = (RT) rn[j];   // (exact: the magnitude is already a value of RT)
    }
    return noisy;
}

// The same step for SUM-PRODUCT (ALGO = 0; the reference's check rule, bp.h:49-57, in the layered schedule): magnitudes through
// phi, exclude-self sums by prefix / suffix (never total - own: an infinite term would turn into NaN), phi again — two phi per
// edge and iteration, as in the flooding kernels, but about half the iterations.  The posteriors and messages live in the
// log2(e)-scaled domain of Dom<float>.  A message saturates at LAYERED_SPA_SATURATION (57.7 in natural units, far beyond the
// reference's own saturation of phi at 45.7): an infinite message would turn the next P - R into inf - inf.
constexpr float LAYERED_SPA_SATURATION = 83.25f;
template <int D, int G, typename RT>
__device__ __forceinline__ uint32_t layer_back_spa(RT *__restrict__ Rl, float *const (&addr)[LMAXD], const float (&p)[LMAXD], const float (&q)[LMAXD],
                                                   const bool store) {
    uint32_t S = 0, noisy = 0;
    float mag[D], pre[D];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        noisy ^= __float_as_uint(p[j]);                     // parity of the hard decisions this check sees
        S ^= __float_as_uint(q[j]);
        mag[j] = Dom<float>::phi(__uint_as_float(__float_as_uint(q[j]) & 0x7FFFFFFFu));
        pre[j] = s;
        s += mag[j];
    }
    float rn[D], pn[D];
    float suf = 0.0f;
#pragma unroll
    for (int j = D - 1; j >= 0; --j) {
        float out = __builtin_fminf(Dom<float>::phi(pre[j] + suf), LAYERED_SPA_SATURATION);
        suf += mag[j];
        out = (float) (RT) out;                              // (fp16 storage: P' adds exactly what the next iteration subtracts)
        rn[j] = __uint_as_float((__float_as_uint(out) & 0x7FFFFFFFu) | ((S ^ __float_as_uint(q[j])) & 0x80000000u));
        pn[j] = q[j] + rn[j];
        noisy |= __float_as_uint(pn[j]) ^ __float_as_uint(p[j]);
    }
    if (store) {
#pragma unroll
        for (int j = 0; j < D; ++j) *addr[j] = pn[j];
#pragma unroll
        for (int j = 0; j < D; ++j) Rl[j * G] = (RT) rn[j];
    }
    return noisy;
}

// Posterior cells of a quasi-cyclic layer WITHOUT the table: variable = C_j * Z + (k + s_j) mod Z (optimize_H.cpp:41), all in
// bytes: proto word j of the block row = C_j * Z * 4 << 16 | s_j * 4 (wave-uniform, scalar loads); k4 = 4 * (row of this lane in its
// block row).  (k + s) mod Z as an unsigned minimum: k4 + s4 - 4 Z wraps to a huge number exactly when no reduction is due.
// Four VALU instructions per edge in place of a ds_read_u16 and the LDS round trip it puts in front of the posterior reads.
template <int D, int G, typename RT>
__device__ __forceinline__ void layer_fetch_qc(const RT *__restrict__ Rl, const int32_t *proto_b, const int k4, const int z4,
                                               int (&pos)[LMAXD], float (&r)[LMAXD]) {
#pragma unroll
    for (int j = 0; j < D; ++j) r[j] = (float) Rl[j * G];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const int w = lsload(proto_b, j);
        const uint32_t tt = (uint32_t) (k4 + (w & 0xFFFF));
        pos[j] = (int) (__builtin_elementwise_min(tt, tt - (uint32_t) z4) + ((uint32_t) w >> 16));
    }
}

// one layer step: positions, posteriors, messages, arithmetic, stores.  QCA: positions by arithmetic (quasi-cyclic H), else
// from the workgroup's table
template <int D, int G, bool QCA, typename RT, int ALGO>
__device__ __forceinline__ uint32_t layer_step(unsigned char *__restrict__ Pb, RT *__restrict__ Rl, const uint16_t *__restrict__ Tl,
                                               const int32_t *proto_b, const int k4, const int z4, const bool store, const float scale) {
    int pos[LMAXD];
    float r[LMAXD];
    if constexpr (QCA) layer_fetch_qc<D, G, RT>(Rl, proto_b, k4, z4, pos, r);
    else layer_fetch<D, G, RT>(Rl, Tl, pos, r);
    float *addr[LMAXD];
    float p[LMAXD], q[LMAXD];
    layer_front<D>(Pb, pos, r, addr, p, q);
    if constexpr (ALGO == 0) return layer_back_spa<D, G, RT>(Rl, addr, p, q, store);
    else return layer_back<D, G, RT>(Rl, addr, p, q, store, scale);
}

// parity of the posteriors' signs over this lane's check of a layer (explicit syndrome pass)
template <int D, int G>
__device__ __forceinline__ uint32_t layer_parity(const unsigned char *__restrict__ Pb, const uint16_t *__restrict__ Tl) {
    uint32_t S = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) S ^= __float_as_uint(*reinterpret_cast<const float *>(Pb + Tl[j * G]));
    return S;
}

#define ACG_LAYER_SWITCH(md, CALL) \
    switch (md) {                  \
        case 1: CALL(1); break;    \
        case 2: CALL(2); break;    \
        case 3: CALL(3); break;    \
        case 4: CALL(4); break;    \
        case 5: CALL(5); break;    \
        case 6: CALL(6); break;    \
        case 7: CALL(7); break;    \
        case 8: CALL(8); break;    \
        default: break;            \
    }

template <int G>
__device__ __forceinline__ bool lgroup_any(bool pred, int g) {
    const unsigned long long b = __ballot(pred);
    if (G == 64) return b != 0ull;
    const unsigned long long mask = ((1ull << (G & 63)) - 1ull) << (g * G);
    return (b & mask) != 0ull;
}

// QCA: the hot loop computes the posterior addresses of a quasi-cyclic H arithmetically (no table read); the table is still
// built once per workgroup for the rare explicit syndrome pass.
// ALGO: 1 = normalised min-sum, 0 = sum-product (phi domain)
// MC: Monte-Carlo mode — the frame's AWGN symbols are generated in the kernel (Philox4x32-10 keyed on (seed, global frame, symbol
// quad) + Box-Muller: the same symbols, bit for bit, as awgn_kernel and the flooding kernels produce), the decoded word is
// classified against the sent one at exit and the seven counters of experiment.h:25-68,109-120 are accumulated per group and
// flushed with one atomic per counter per group when the kernel ends.
template <int G, int WAVES, bool QCA, typename RT, int ALGO, bool MC>
__global__ void __launch_bounds__(WAVES * 64) bp_layered_kernel(const LayerTables t, const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / G;                 // (lanes beyond FPW * G idle: 60..63 for G = 20)
    const int l = lane - g * G;
    const bool lane_used = g < FPW;

    // ---- position table of the workgroup: byte offset of the posterior cell of (layer, edge, lane) ----------------------------
    uint16_t *TAB = reinterpret_cast<uint16_t *>(smem);
    if (t.proto) {
        // quasi-cyclic H: edge addressing is arithmetic, evaluated once per workgroup
        for (int b = 0; b < t.n_layers; ++b) {
            const int deg = lsload(t.layer, 4 * b), off = lsload(t.layer, 4 * b + 1), cnt = lsload(t.layer, 4 * b + 2);
            const int w3 = lsload(t.layer, 4 * b + 3), p0 = w3 & 0xFFFF, row0 = w3 >> 16;
            for (int i = threadIdx.x; i < deg * G; i += blockDim.x) {
                const int j = i / G, k = i - j * G;
                int v = t.n;                                                           // the neutral cell
                if (k < cnt) {
                    const int C = t.proto[2 * (p0 + j)], s = t.proto[2 * (p0 + j) + 1];
                    v = C * t.Z + (row0 + k + s) % t.Z;                                // optimize_H.cpp:41
                }
                TAB[off + i] = (uint16_t) (4 * v);
            }
        }
    } else {
        for (int i = threadIdx.x; i < t.e_pad; i += blockDim.x) TAB[i] = (uint16_t) (4 * t.pos[i]);
    }
    __syncthreads();

    const int grp_in_block = wave * FPW + (lane_used ? g : 0);
    unsigned char *base = smem + t.tab_lds_bytes + (size_t) grp_in_block * t.lds_bytes_per_frame;
    unsigned char *Pb = base;                                             // P[n] + neutral cell (+ padding)
    float *P = reinterpret_cast<float *>(base);
    RT *R = reinterpret_cast<RT *>(P + t.p_words);
    uint32_t *OB = reinterpret_cast<uint32_t *>(P + t.p_words + t.r_words);
    uint32_t *HAMW = OB + t.nwords;   // MC: raw-channel error count of the frame (one word; the host reserves it)
    const float scale = a.ms_scale;
    const int NL = t.n_layers;

    constexpr int CHUNK = 4 * FPW;
    int64_t wnext = 0, wend = 0;
    int64_t frame = 0;
    bool active = false, want = lane_used, need_init = false, latched = false;
    int it = 0;             // iterations (rounds over all layers) this frame has been through
    uint32_t noisy_acc = 0; // sign bit: some step of the current round was not quiet for this lane's checks
    int ham = 0;            // MC: raw-channel errors of the current frame
    unsigned int acc_correct = 0, acc_pseudo = 0, acc_total = 0;
    unsigned long long acc_ham = 0, acc_ham_ok = 0, acc_ham_wrong = 0, acc_iters = 0;

    auto emit = [&](const bool out_now, const bool fail_now) {
        if (__ballot(out_now || fail_now) != 0ull) {
            if (out_now || fail_now)
                for (int w = l; w < t.nwords; w += G) OB[w] = 0u;
            lwave_sync();
            if (out_now)
                for (int v = l; v < t.n; v += G)
                    if (__float_as_uint(P[v]) >> 31) atomicOr(&OB[v >> 5], 1u << (v & 31));
            lwave_sync();
            if (out_now || fail_now) {
                if (a.out_bits)
                    for (int w = l; w < t.nwords; w += G) a.out_bits[(size_t) frame * t.nwords + w] = OB[w];
                if (l == 0) {
                    if (a.out_ok) a.out_ok[frame] = out_now ? 1 : 0;
                    if (a.out_iters) a.out_iters[frame] = std::min(it, a.max_iter);
                }
                if (MC) {
                    bool neq = false;
                    const int64_t gf = a.first_frame + frame;
                    for (int w = l; w < t.nwords; w += G) {
                        const uint32_t cwv = a.cw_packed ? a.cw_packed[(size_t) (gf % a.n_cw) * t.nwords + w] : 0u;
                        neq |= (OB[w] != cwv);
                    }
                    const bool differ = lgroup_any<G>(neq, g);   // every lane of the group is in this branch together
                    const bool correct = out_now && !differ;     // experiment.h:110-114
                    acc_correct += correct;
                    acc_pseudo += (out_now && differ);            // experiment.h:115-116
                    acc_total += 1;
                    acc_ham += ham;
                    acc_ham_ok += correct ? ham : 0;
                    acc_ham_wrong += correct ? 0 : ham;
                    acc_iters += std::min(it, a.max_iter);
                }
                latched = true;
            }
        }
    };
    // explicit syndrome of the posteriors' signs (frames that ran out of iterations without a quiet round)
    auto syndrome_bad = [&]() -> bool {
        uint32_t acc = 0;
        for (int bb = 0; bb < NL; ++bb) {
            const int deg = lsload(t.layer, 4 * bb), off = lsload(t.layer, 4 * bb + 1), cnt = lsload(t.layer, 4 * bb + 2);
            const uint16_t *Tl = TAB + off + l;
            uint32_t S = 0;
#define ACG_CALL(D) S = layer_parity<D, G>(Pb, Tl)
            ACG_LAYER_SWITCH(deg, ACG_CALL)
#undef ACG_CALL
            acc |= (l < cnt) ? S : 0u;
        }
        return lgroup_any<G>((acc >> 31) != 0u, g);
    };

    for (;;) {
        // ---- round boundary: frames that are done -------------------------------------------------------------------------
        const bool loud = lgroup_any<G>((noisy_acc >> 31) != 0u, g);
        noisy_acc = 0;
        const bool conv = active && it > 0 && !loud;
        const bool out_of_sweeps = active && it >= a.max_iter;
        bool conv2 = conv;
        if (__ballot(out_of_sweeps && !conv && !latched) != 0ull) {
            const bool bad = syndrome_bad();            // wave-uniform control flow; the result is per group
            if (out_of_sweeps && !conv && !latched && !bad && a.max_iter > 0) conv2 = true;
        }
        const bool out_now = conv2 && !latched;
        const bool finish = active && ((a.early_exit && conv2) || out_of_sweeps);
        const bool fail_now = finish && !conv2 && !latched;
        emit(out_now, fail_now);
        if (finish) {
            active = false;
            want = lane_used;
        }
        // ---- deal new frames to the groups that want one ------------------------------------------------------------------
        unsigned long long wm = __ballot(want && l == 0);
        while (wm != 0ull) {
            const int leader = __builtin_ctzll(wm);
            wm &= wm - 1ull;
            if (wnext >= wend) {
                unsigned long long nb = 0;
                if (lane == 0) nb = atomicAdd(a.work_counter, (unsigned long long) CHUNK);
                const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t) nb);
                const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t) (nb >> 32));
                wnext = (int64_t) (((unsigned long long) bhi << 32) | blo);
                wend = wnext + CHUNK < a.frames ? wnext + CHUNK : a.frames;
            }
            const bool got = wnext < wend;
            const int64_t f = wnext;
            if (got) wnext += 1;
            if (lane_used && g == leader / G) {
                frame = f;
                active = got;
                need_init = got;
                want = false;
            }
        }
        if (__ballot(active) == 0ull) break;
        // ---- (re)start groups on a new frame: P = channel LLR (channel.h:14-16), R = 0 ------------------------------------
        if (__ballot(need_init) != 0ull) {
            lwave_sync();
            if (MC) {
                if (need_init && l == 0) *HAMW = 0u;
                lwave_sync();
            }
            if (need_init && MC) {
                // transmit (channel.h:18-26) + llr (channel.h:14-16) + HammingDistanceTracker (experiment.h:33-46)
                const int64_t gf = a.first_frame + frame;
                const uint32_t *cw = a.cw_packed ? a.cw_packed + (size_t) (gf % a.n_cw) * t.nwords : nullptr;
                int my_ham = 0;
                const int nq = (t.n + 3) >> 2;
                for (int qd = l; qd < nq; qd += G) {
                    uint32_t rr[4];
                    philox4x32_10((uint32_t) gf, (uint32_t) (gf >> 32), (uint32_t) qd, 0u, (uint32_t) a.seed, (uint32_t) (a.seed >> 32), rr);
                    float z[4];
                    box_muller(rr[0], rr[1], z[0], z[1]);
                    box_muller(rr[2], rr[3], z[2], z[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = 4 * qd + e;
                        if (v < t.n) {
                            const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
                            const float yv = __builtin_fmaf(a.sigma, z[e], bit ? -1.0f : 1.0f);   // explicit fma: same symbol in every kernel
                            my_ham += ((!bit && yv <= 0.0f) || (bit && yv > 0.0f)) ? 1 : 0;
                            const float llr = (float) ((double) yv * a.inv_var2);
                            P[v] = (ALGO == 0) ? llr * (float) Dom<float>::scale : llr;
                        }
                    }
                }
                if (my_ham) atomicAdd(HAMW, (uint32_t) my_ham);
            }
            if (need_init) {
                if (!MC)
                    for (int v = l; v < t.n; v += G) {
                        float llr;
                        if (a.y_is_f64) llr = (float) (2 * reinterpret_cast<const double *>(a.y)[(size_t) frame * t.n + v] / a.var);
                        else llr = (float) ((double) reinterpret_cast<const float *>(a.y)[(size_t) frame * t.n + v] * a.inv_var2);
                        P[v] = (ALGO == 0) ? llr * (float) Dom<float>::scale : llr;
                    }
                for (int w = t.n + l; w < t.p_words; w += G) P[w] = INFINITY;   // neutral cell: never the minimum, sign +
                for (int w = l; w < t.e_pad; w += G) R[w] = (RT) 0.0f;
                it = 0;
                latched = false;
                need_init = false;
            }
            lwave_sync();
            if (MC && active && it == 0) ham = (int) *HAMW;   // (groups that were not re-initialised re-read their own count)
        }
        // ---- one iteration: every layer in turn, posteriors updated in place ------------------------------------------------
        for (int b = 0; b < NL; ++b) {
            const int deg = lsload(t.layer, 4 * b), off = lsload(t.layer, 4 * b + 1), cnt = lsload(t.layer, 4 * b + 2);
            const bool mine = active && l < cnt;
            uint32_t noisy = 0;
            const int w3 = QCA ? lsload(t.layer, 4 * b + 3) : 0;
            const int32_t *proto_b = t.proto_packed + (w3 & 0xFFFF);
            const int k4 = 4 * ((w3 >> 16) + l);
#define ACG_CALL(D) noisy = layer_step<D, G, QCA, RT, ALGO>(Pb, R + off + l, TAB + off + l, proto_b, k4, 4 * t.Z, mine, scale)
            ACG_LAYER_SWITCH(deg, ACG_CALL)
#undef ACG_CALL
            lwave_sync();
            noisy_acc |= mine ? noisy : 0u;
        }
        it += active ? 1 : 0;
    }
    if (MC && lane_used && l == 0 && acc_total) {
        atomicAdd(&a.counters[MC_CORRECT], (unsigned long long) acc_correct);
        atomicAdd(&a.counters[MC_PSEUDO], (unsigned long long) acc_pseudo);
        atomicAdd(&a.counters[MC_TOTAL], (unsigned long long) acc_total);
        atomicAdd(&a.counters[MC_HAM], acc_ham);
        atomicAdd(&a.counters[MC_HAM_OK], acc_ham_ok);
        atomicAdd(&a.counters[MC_HAM_WRONG], acc_ham_wrong);
        atomicAdd(&a.counters[MC_ITERS], acc_iters);
    }
}

template <int G, bool QCA, typename RT, int ALGO, bool MC>
static const void *layered_ptr_w(int waves) {
    switch (waves) {
        case 1: return (const void *) bp_layered_kernel<G, 1, QCA, RT, ALGO, MC>;
        case 2: return (const void *) bp_layered_kernel<G, 2, QCA, RT, ALGO, MC>;
        default: return (const void *) bp_layered_kernel<G, 4, QCA, RT, ALGO, MC>;
    }
}
template <int G, bool MC>
static const void *layered_ptr_g(int waves, bool f16, int algo) {
    if (algo == 0) return f16 ? layered_ptr_w<G, false, _Float16, 0, MC>(waves) : layered_ptr_w<G, false, float, 0, MC>(waves);
    return f16 ? layered_ptr_w<G, false, _Float16, 1, MC>(waves) : layered_ptr_w<G, false, float, 1, MC>(waves);
}

// algo: 0 sum-product, 1 min-sum.  qc_arith: positions computed in the hot loop instead of read from the table (G = 20, fp32
// min-sum decode only: a measured alternative, 14 % slower — DESIGN §3d); f16: messages stored in half precision; mc: the
// Monte-Carlo instance (noise generated and words classified in the kernel)
const void *bp_layered_kernel_ptr(int G, int waves, bool qc_arith, bool f16, int algo, bool mc) {
    if (qc_arith) return (G == 20 && !f16 && algo == 1 && !mc) ? layered_ptr_w<20, true, float, 1, false>(waves) : nullptr;
    switch (G) {
        case 16: return mc ? layered_ptr_g<16, true>(waves, f16, algo) : layered_ptr_g<16, false>(waves, f16, algo);
        case 20: return mc ? layered_ptr_g<20, true>(waves, f16, algo) : layered_ptr_g<20, false>(waves, f16, algo);
        case 32: return mc ? layered_ptr_g<32, true>(waves, f16, algo) : layered_ptr_g<32, false>(waves, f16, algo);
        case 64: return mc ? layered_ptr_g<64, true>(waves, f16, algo) : layered_ptr_g<64, false>(waves, f16, algo);
        default: return nullptr;
    }
}

hipError_t bp_layered_launch(const void *kernel, const LayerTables &t, const DecodeArgs &a, int grid, int block, size_t lds, hipStream_t s) {
    LayerTables tt = t;
    DecodeArgs aa = a;
    void *args[2] = {&tt, &aa};
    return hipLaunchKernel(kernel, dim3(grid), dim3(block), args, lds, s);
}

}  // namespace acg
