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
// Source layout: bp_core.inc holds the kernel; bp_inst_{spa,ms}_{f32,f64}.hip instantiate it (parallel
// compilation); this file holds the dispatcher, the launcher and the stand-alone debug/AWGN kernels.
//
// A wavefront is a persistent worker: each L-lane group walks frames g, g+G, g+2G, ... and
// restarts on a new frame the moment its current one reaches a zero syndrome (the reference's
// early exit, bp.h:195-196), independently of the other groups in the wave.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"

namespace acg {
#include "bp_core.inc"

const void *bp_kernel_ptr_spa_f32(int maxd, int L, bool mc, int variant);
const void *bp_kernel_ptr_spa_f64(int maxd, int L, bool mc, int variant);
const void *bp_kernel_ptr_ms_f32(int maxd, int L, bool mc, int variant);
const void *bp_kernel_ptr_ms_f64(int maxd, int L, bool mc, int variant);

const void *bp_kernel_ptr_spa_f32_dbg(int L);
const void *bp_kernel_ptr_spa_f64_dbg(int L);
const void *bp_kernel_ptr_dbg(int f64, int L) {
#ifdef ACG_FAST_BUILD
    return f64 ? nullptr : bp_kernel_ptr_spa_f32_dbg(L);
#else
    return f64 ? bp_kernel_ptr_spa_f64_dbg(L) : bp_kernel_ptr_spa_f32_dbg(L);
#endif
}

// algo: 0 sum-product, 1 min-sum; f64: 0/1
const void *bp_kernel_ptr(int algo, int f64, int maxd, int L, bool mc, int variant) {
#ifdef ACG_FAST_BUILD
    if (f64 || algo) return nullptr;
    return bp_kernel_ptr_spa_f32(maxd, L, mc, variant);
#else
    if (algo == 0) return f64 ? bp_kernel_ptr_spa_f64(maxd, L, mc, variant) : bp_kernel_ptr_spa_f32(maxd, L, mc, variant);
    return f64 ? bp_kernel_ptr_ms_f64(maxd, L, mc, variant) : bp_kernel_ptr_ms_f32(maxd, L, mc, variant);
#endif
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
    // the fp32 kernels evaluate phi in the log2(e)-scaled domain: report it back in natural units
    if (i < n) out[i] = (float) ((double) Dom<float>::phi((float) ((double) x[i] * Dom<float>::scale)) / Dom<float>::scale);
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
                y[(size_t) f * n + v] = __builtin_fmaf(sigma, z[e], bit ? -1.0f : 1.0f);  // explicit fma: same symbol in every kernel
            }
        }
    }
}

// Per-frame classification of exp() (experiment.h:109-120) for engines without an in-kernel generator:
// one wavefront per frame; correct <=> ok and bits == sent word; raw-channel Hamming count from y.
__global__ void classify_kernel(const float *y, const uint32_t *bits, const uint8_t *ok, const int32_t *iters,
                                int64_t frames, int n, int nwords, int64_t first_frame, const uint32_t *cw_packed,
                                int64_t n_cw, unsigned long long *counters, const int32_t *row_ptr,
                                const int32_t *edge_var, int m) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nw = (int64_t) gridDim.x * (blockDim.x >> 6);
    unsigned long long c_ok = 0, c_ps = 0, c_tot = 0, c_h = 0, c_hok = 0, c_hw = 0, c_it = 0;
    for (int64_t f = wid; f < frames; f += nw) {
        const uint32_t *cw = cw_packed ? cw_packed + (size_t) ((first_frame + f) % n_cw) * nwords : nullptr;
        int ham = 0;
        for (int v = lane; v < n; v += 64) {
            const uint32_t bit = cw ? ((cw[v >> 5] >> (v & 31)) & 1u) : 0u;
            const float yv = y[(size_t) f * n + v];
            ham += ((!bit && yv <= 0.0f) || (bit && yv > 0.0f)) ? 1 : 0;
        }
        bool neq = false;
        for (int w = lane; w < nwords; w += 64) neq |= (bits[(size_t) f * nwords + w] != (cw ? cw[w] : 0u));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ham += __shfl_xor(ham, o, 64);
        const bool differ = __ballot(neq) != 0ull;
        bool okf = ok[f] != 0;
        if (row_ptr) {  // decoders that always report ok (QP-ADMM, qp_admm.h:177): IsCodeword here (experiment.h:111)
            bool sbad = false;
            for (int c = lane; c < m; c += 64) {
                uint32_t sy = 0;
                for (int e = row_ptr[c]; e < row_ptr[c + 1]; ++e) {
                    const int v = edge_var[e];
                    sy ^= (bits[(size_t) f * nwords + (v >> 5)] >> (v & 31)) & 1u;
                }
                sbad |= (sy != 0u);
            }
            okf = okf && (__ballot(sbad) == 0ull);
        }
        const bool correct = okf && !differ;
        c_ok += correct;
        c_ps += (okf && differ);
        c_tot += 1;
        c_h += ham;
        c_hok += correct ? ham : 0;
        c_hw += correct ? 0 : ham;
        c_it += iters ? iters[f] : 0;
    }
    if (lane == 0 && c_tot) {
        atomicAdd(&counters[MC_CORRECT], c_ok);
        atomicAdd(&counters[MC_PSEUDO], c_ps);
        atomicAdd(&counters[MC_TOTAL], c_tot);
        atomicAdd(&counters[MC_HAM], c_h);
        atomicAdd(&counters[MC_HAM_OK], c_hok);
        atomicAdd(&counters[MC_HAM_WRONG], c_hw);
        atomicAdd(&counters[MC_ITERS], c_it);
    }
}

hipError_t classify_launch(const float *y, const uint32_t *bits, const uint8_t *ok, const int32_t *iters, int64_t frames,
                           int n, int nwords, int64_t first_frame, const uint32_t *cw_packed, int64_t n_cw,
                           unsigned long long *counters, const int32_t *row_ptr, const int32_t *edge_var, int m,
                           hipStream_t s) {
    int grid = (int) std::min<int64_t>((frames + 3) / 4, 256 * 8);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(classify_kernel, dim3(grid), dim3(256), 0, s, y, bits, ok, iters, frames, n, nwords, first_frame,
                       cw_packed, n_cw, counters, row_ptr, edge_var, m);
    return hipGetLastError();
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
