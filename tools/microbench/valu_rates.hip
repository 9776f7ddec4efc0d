// Micro-benchmark (developer tool, not part of the product): VALU issue rates on gfx950 that the BP
// kernel's cost model depends on.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
//   1. v_fma_f32 throughput per SIMD at 1..8 waves/SIMD
//   2. v_exp_f32 / v_log_f32 throughput: "exp"/"log" time the pair mul+exp / add+log, "exp-alone"/"log-alone" the
//      transcendental by itself (its price = the pair minus one 2-cycle op if the two do not overlap)
//   3. does a wave64 VALU op with one 32-lane half fully EXEC-masked issue in half the time?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 32768

// stamps[block] = {shader cycles (s_memtime), 100 MHz ticks (s_memrealtime)} spent inside the timed loop by wave 0 of the block
template <int MODE>
__global__ void k(float *out, int mask_upper, unsigned long long *stamps) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 0.999f, c = 1e-3f;
    const bool on = !(mask_upper && (threadIdx.x & 32));
    if (on) {
        for (int i = 0; i < N_ITER; ++i) {
            if (MODE == 0) {
#define F(x) x = __builtin_fmaf(x, b, c)
                F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
#undef F
            } else if (MODE == 1) {
#define F(x) x = __builtin_amdgcn_exp2f(x * -0.001f)
                F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
#undef F
            } else if (MODE == 2) {
#define F(x) x = __builtin_amdgcn_logf(x + 2.0f)
                F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
#undef F
            } else if (MODE == 4) {  // v_exp_f32 ALONE (negation is a free source modifier; the value settles at the fixed point 0.641...)
#define F(x) x = __builtin_amdgcn_exp2f(-x)
                F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
#undef F
            } else if (MODE == 5) {  // v_log_f32 ALONE (|x| is a free source modifier)
#define F(x) x = __builtin_amdgcn_logf(__builtin_fabsf(x))
                F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
#undef F
            } else {  // dependent chain
                a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c);
                a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c); a0 = __builtin_fmaf(a0, b, c);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
        stamps[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

template <int MODE>
static void run(const char *name, int waves_per_simd, int mask_upper, float *d) {
    // one block of 256 threads = 4 waves = 1 wave per SIMD of a CU; launch 256 CUs * waves_per_simd blocks
    const int blocks = 256 * waves_per_simd;
    static unsigned long long *stamps = nullptr;
    if (!stamps) hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, mask_upper, stamps);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, mask_upper, stamps);
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
    const double insts_per_simd = (double) waves_per_simd * N_ITER * 8;
    // wall: 100 MHz ticks.  The shader clock while the loop ran = cyc / (wall / 100e6).
    printf("%-10s waves/SIMD %d mask_upper %d : %.3f ms | in-kernel: %.3f shader cycles per wave-instruction per SIMD, shader clock %.0f MHz "
           "| from the event time at a nominal 2.4 GHz: %.2f\n", name, waves_per_simd, mask_upper, ms, cyc / insts_per_simd,
           cyc / (wall / 100e6) / 1e6, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
    for (int w : {1, 2, 4, 8}) run<0>("fma", w, 0, d);
    for (int w : {1, 2, 4, 8}) run<1>("exp", w, 0, d);
    for (int w : {1, 4, 8}) run<2>("log", w, 0, d);
    for (int w : {1, 2, 4, 8}) run<4>("exp-alone", w, 0, d);
    for (int w : {1, 4, 8}) run<5>("log-alone", w, 0, d);
    for (int w : {1, 2, 4, 8}) run<3>("fma-chain", w, 0, d);
    for (int w : {1, 4, 8}) run<0>("fma", w, 1, d);
    for (int w : {4}) run<1>("exp", w, 1, d);
    return 0;
}
