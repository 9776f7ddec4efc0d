// Micro-benchmark (developer tool): what one VALU wave-instruction costs a SIMD on gfx950, by operation — the fp64
// arithmetic of QP-ADMM, the 32-bit helpers around it (v_and_or_b32, shifts, SDWA adds), compares / selects / min / max,
// packed f16, the transcendentals, and v_fma_f32 as the yardstick.  W wavefronts per SIMD run the same loop of 8
// independent chains (128 instructions per trip); the figure is kernel time x 2.4 GHz / (instructions per wavefront x W),
// i.e. wall time at the nominal clock (in brackets the same from the wavefront's own s_memtime ticks, which run slower
// than the shader clock under load and are only good for comparing rows).  Result (profiles/r02_valu_op_rates.txt):
// two classes — about 2.3 cycles (fp32 add/mul/fma, and/or/xor, add/sub, right shifts, v_mov_b32, v_cndmask with vcc)
// and about 4.15 (every fp64 operation, v_lshlrev_b32, v_cmp_*, v_min/max_f32, v_cndmask_b32 with an SGPR mask, all
// three-operand integer forms, SDWA, DPP, conversions, 24-bit multiplies, packed f16) — and 8.3 for v_exp/v_log_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 8192
#define N_REP 16  // the 8 chains are stepped 16 times per trip: loop control is 1 scalar branch per 128 VALU instructions
#define OPS(X)                                                                                                          \
    X(0, "v_fma_f32 (3 VGPR)", asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0.999f), "v"(1e-3f)))        \
    X(1, "v_add_f32", asm volatile("v_add_f32 %0, %1, %0" : "+v"(u[i]) : "v"(1e-3f)))                                   \
    X(2, "v_and_b32", asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(0xfffffff0u)))                            \
    X(3, "v_xor_b32", asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(0x80000000u)))                            \
    X(4, "v_add_u32", asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(12345u)))                                  \
    X(5, "v_lshlrev_b32 (by 1)", asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i])))                                   \
    X(6, "v_lshrrev_b32 (by 3)", asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i])))                                   \
    X(7, "v_mov_b32", asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7])))                              \
    X(8, "v_cndmask_b32 (SGPR mask)", asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(77u), "s"(0x5555555555555555ull)))  \
    X(9, "v_cmp_gt_f32 (vcc)", asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(u[i]), "v"(0.5f) : "vcc"))               \
    X(25, "v_cmp_gt_f32 (SGPR pair)", asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m[i]) : "v"(u[i]), "v"(0.5f)))       \
    X(26, "v_cmp_gt_i32 (vcc)", asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(u[i]), "v"(5u) : "vcc"))                 \
    X(27, "v_cmp + v_cndmask (vcc)", asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(0.5f) : "vcc")) \
    X(28, "v_max_f32 (plain)", asm volatile("v_max_f32 %0, %1, %0" : "+v"(u[i]) : "v"(0.5f)))                             \
    X(29, "v_min_f32 (plain)", asm volatile("v_min_f32 %0, %1, %0" : "+v"(u[i]) : "v"(0.5f)))                             \
    X(30, "v_mul_f32", asm volatile("v_mul_f32 %0, %1, %0" : "+v"(u[i]) : "v"(0.999f)))                                   \
    X(31, "v_fmaak_f32", asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(u[i]) : "v"(0.999f)))                   \
    X(32, "v_fmac_f32", asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(u[i]) : "v"(0.999f), "v"(1e-3f)))                     \
    X(33, "v_add_f32 (|x|: VOP3)", asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(u[i]) : "v"(1e-3f)))                      \
    X(34, "v_log_f32", asm volatile("v_log_f32 %0, |%0|" : "+v"(u[i])))                                                   \
    X(35, "v_med3_f32", asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0.1f), "v"(0.9f)))                    \
    X(36, "v_min3_f32", asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0.1f), "v"(0.9f)))                    \
    X(37, "v_and_b32 (literal)", asm volatile("v_and_b32 %0, 0x7ffffffe, %0" : "+v"(u[i])))                               \
    X(38, "v_or_b32", asm volatile("v_or_b32 %0, %1, %0" : "+v"(u[i]) : "v"(1u)))                                         \
    X(39, "v_mov_b32 dpp (quad_perm)", asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) & 7]))) \
    X(40, "v_cvt_f32_i32", asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(u[i])))                                             \
    X(41, "v_sub_u32", asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(3u)))                                       \
    X(42, "v_mul_u32_u24", asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(3u)))                               \
    X(43, "v_mad_u32_u24", asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(3u), "v"(7u)))                  \
    X(44, "v_ashrrev_i32 (by 31)", asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(u[i])))                                 \
    X(45, "v_lshlrev_b32 (VGPR amount)", asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(1u)))                 \
    X(46, "v_add_u32 x2 (= shift left 1)", asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[i])))                             \
    X(10, "v_and_or_b32 (SGPR mask)", asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "s"(0x80000000u), "v"(0x3ff00000u))) \
    X(11, "v_and_or_b32 (3 VGPR)", asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0x80000000u), "v"(0x3ff00000u)))    \
    X(12, "v_lshl_add_u32", asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(16u)))                       \
    X(13, "v_bfe_u32", asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i])))                                             \
    X(14, "v_bfi_b32", asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(0x80000000u), "v"(0x3ff00000u)))      \
    X(15, "v_add_u32_sdwa (WORD_0)", asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(u[i]) : "s"(16u))) \
    X(16, "v_max_f32 (-x, 0)", asm volatile("v_max_f32 %0, -%0, 0" : "+v"(u[i])))                                       \
    X(17, "v_fma_f64", asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(0.999), "v"(1e-3)))                   \
    X(18, "v_add_f64", asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(1e-3)))                                   \
    X(19, "v_mul_f64", asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(0.999)))                                  \
    X(20, "v_max_f64 (-x, 0)", asm volatile("v_max_f64 %0, -%0, 0" : "+v"(a[i])))                                       \
    X(21, "v_mov_b64", asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7])))                             \
    X(22, "v_exp_f32", asm volatile("v_exp_f32 %0, -%0" : "+v"(u[i])))                                                  \
    X(23, "v_pk_add_f16", asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(0x3c003c00u)))                      \
    X(24, "v_pk_min_f16", asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(u[i]) : "v"(0x3c003c00u)))                      \
    X(47, "v_pk_fma_f32 (2 x fp32)", asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(p2a), "v"(p2b)))        \
    X(48, "v_pk_mul_f32 (2 x fp32)", asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(p2a)))                      \
    X(49, "v_pk_add_f32 (2 x fp32)", asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(p2b)))

template <int MODE>
__global__ void k(double *out, unsigned long long *stamps) {
    double a[8];
    uint32_t u[8];
    unsigned long long m[8] = {};
    const double p2a = __longlong_as_double(0x3f7fbe773f7fbe77ll), p2b = __longlong_as_double(0x3a83126f3a83126fll);  // (0.999f, 0.999f), (1e-3f, 1e-3f)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-3 + 1.0 + i;
        u[i] = threadIdx.x * 2654435761u + i;
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < N_REP; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(M, NAME, STMT) if (MODE == M) STMT;
                OPS(X)
#undef X
            }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (double) u[i] + (double) m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int MODE>
static void run(const char *name, double *d) {
    static unsigned long long *stamps = nullptr;
    if (!stamps) hipMalloc(&stamps, sizeof(unsigned long long) * 4 * 256 * 8);
    printf("%-28s", name);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w : {1, 2, 8}) {
        const int blocks = 256 * w;  // 256-thread workgroups: one wavefront per SIMD each; w workgroups per CU
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, stamps);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, stamps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(4 * blocks);
        hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 4 * blocks, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (auto x : h) cyc += (double) x;
        cyc /= (double) h.size();
        const double insts = (double) N_ITER * N_REP * 8 * w;
        // wall: kernel time x 2.4 GHz per instruction per SIMD | clk: the wavefront's own s_memtime ticks per instruction / W
        printf("  W=%d: %5.2f (%4.2f)", w, ms * 1e-3 * 2.4e9 / insts, cyc / insts);
    }
    printf("\n");
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}
int main() {
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    printf("per wave-instruction and SIMD, W wavefronts per SIMD: kernel time x 2.4 GHz (in brackets: s_memtime ticks of a wavefront / W)\n");
#define X(M, NAME, STMT) run<M>(NAME, d);
    OPS(X)
#undef X
    return 0;
}
