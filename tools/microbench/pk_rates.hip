// Micro-benchmark (developer tool): does v_pk_fma_f32 / v_pk_mul_f32 give 2 fp32 results per lane at the issue cost of
// one v_fma_f32 on gfx950?   hipcc --offload-arch=gfx950 -O3 pk_rates.hip -o pk_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ITER 4096
template <int MODE>
__global__ void k(float *out) {
    f2 a0 = {threadIdx.x * 1e-3f + 1.0f, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    const f2 b = {0.999f, 0.998f}, c = {1e-3f, 2e-3f};
    for (int i = 0; i < N_ITER; ++i) {
        if (MODE == 0) {
#define F(x) x = __builtin_elementwise_fma(x, b, c)
            F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
#undef F
        } else {
#define F(v) v.x = __builtin_fmaf(v.x, b.x, c.x)
            F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
#undef F
        }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE>
static void run(const char *name, int wps, float *d) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d);
    (void) hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d);
    (void) hipEventRecord(e1);
    (void) hipEventSynchronize(e1);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-12s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, wps, ms,
           ms * 1e-3 * 2.4e9 / ((double) wps * N_ITER * 8));
}
int main() {
    float *d;
    (void) hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
    for (int w : {2, 4, 8}) run<0>("pk_fma_f32", w, d);
    for (int w : {2, 4, 8}) run<1>("fma_f32", w, d);
    return 0;
}
