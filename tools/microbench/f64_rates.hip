// Micro-benchmark (developer tool): issue cost of the fp64 VALU operations the QP-ADMM kernels are made of, per
// wave-instruction per SIMD: v_fma_f64, v_add_f64, v_mul_f64, v_max_f64, and the 32-bit helper v_and_or_b32.
// 8 independent accumulators per lane, 1/2/5 wavefronts per SIMD (the QP-ADMM workgroup kernel runs 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 16384
template <int MODE>
__global__ void k(double *out, unsigned long long *stamps) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    double a[8];
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 1e-3 + 1.0 + i;
        u[i] = threadIdx.x * 2654435761u + i;
    }
    const double b = 0.999, c = 1e-3;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "s"(0x80000000u), "v"(0x3ff00000u));
            if (MODE == 5) {  // the v-update's mix: one v_and_or_b32 + one v_lshlrev_b32 per v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_lshlrev_b32 %1, 1, %1\n\tv_and_or_b32 %1, %1, %4, %5"
                             : "+v"(a[i]), "+v"(u[i]) : "v"(b), "v"(c), "s"(0x80000000u), "v"(0x3ff00000u));
            }
            if (MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0.999f), "v"(1e-3f));
            if (MODE == 8) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(0xfffffff0u));
            if (MODE == 9) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
            if (MODE == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(0x80000000u), "v"(0x3ff00000u));
            if (MODE == 11) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(0x80000000u), "v"(0x3ff00000u));
            if (MODE == 12) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(0x80000000u));
            if (MODE == 13) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(u[i]) : "s"(16u));
            if (MODE == 14) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(u[i]) : : "vcc");
            if (MODE == 6) asm volatile("v_max_f64 %0, -%0, 0" : "+v"(a[i]));  // max(-x, 0) with the negate modifier and an inline constant
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (double) u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
        stamps[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}
template <int MODE>
static void run(const char *name, int waves_per_simd, double *d, int per_iter) {
    const int blocks = 256 * waves_per_simd;
    static unsigned long long *stamps = nullptr;
    if (!stamps) hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, stamps);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double) h[2 * b]; wall += (double) h[2 * b + 1]; }
    cyc /= blocks; wall /= blocks;
    const double groups_per_simd = (double) waves_per_simd * N_ITER * 8;
    printf("%-22s waves/SIMD %d : %.3f ms | shader clock %.0f MHz | %.2f cycles per %s per SIMD at the measured clock, %.2f at a nominal 2.4 GHz\n",
           name, waves_per_simd, ms, cyc / (wall / 100e6) / 1e6, ms * 1e-3 * (cyc / (wall / 100e6)) / groups_per_simd,
           per_iter == 1 ? "wave-instruction" : "group of 3 instructions", ms * 1e-3 * 2.4e9 / groups_per_simd);
}
int main() {
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    for (int w : {1, 2, 5}) run<0>("v_fma_f64", w, d, 1);
    for (int w : {1, 5}) run<1>("v_add_f64", w, d, 1);
    for (int w : {1, 5}) run<2>("v_mul_f64", w, d, 1);
    for (int w : {1, 5}) run<3>("v_max_f64", w, d, 1);
    for (int w : {1, 5}) run<6>("v_max_f64 (-x, 0)", w, d, 1);
    for (int w : {1, 5}) run<4>("v_and_or_b32", w, d, 1);
    for (int w : {1, 5}) run<7>("v_fma_f32", w, d, 1);
    for (int w : {1, 5}) run<8>("v_and_b32", w, d, 1);
    for (int w : {1, 5}) run<9>("v_lshlrev_b32", w, d, 1);
    for (int w : {5}) run<10>("v_and_or_b32 (3 VGPR)", w, d, 1);
    for (int w : {5}) run<11>("v_bfi_b32", w, d, 1);
    for (int w : {5}) run<12>("v_xor_b32", w, d, 1);
    for (int w : {5}) run<13>("v_add_u32_sdwa", w, d, 1);
    for (int w : {5}) run<14>("v_cndmask_b32", w, d, 1);
    for (int w : {1, 2, 5}) run<5>("fma_f64+lshl+and_or", w, d, 3);
    return 0;
}
