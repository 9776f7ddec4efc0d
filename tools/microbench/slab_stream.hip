// Microbenchmark behind DESIGN.md §9(1): how fast can a workgroup stream its private [lines][64] slab (read a 256-byte
// line per wave instruction, touch it, write it back) — the access pattern of bp_streamed_kernel — when the lines in
// flight are held (A) in VGPRs, as that kernel does today, or (B) in LDS, landed there by global_load_lds_dword?
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/slab_stream.hip -o slab_stream && ./slab_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int THREADS = 512;  // 8 wavefronts, like bp_streamed_kernel

// (A) G lines per wave in VGPRs at a time
template <int G>
__global__ void __launch_bounds__(THREADS) stream_regs(float *slabs, int lines, int sweeps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    float *S = slabs + (size_t) blockIdx.x * lines * 64;
    for (int s = 0; s < sweeps; ++s) {
        for (int e0 = w * G; e0 < lines; e0 += W * G) {
            float x[G];
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = (e0 + g < lines) ? S[(size_t) (e0 + g) * 64 + lane] : 0.0f;
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = x[g] * 1.0001f + 1.0f;
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (e0 + g < lines) S[(size_t) (e0 + g) * 64 + lane] = x[g];
        }
        __syncthreads();
    }
}

// (B) P lines per wave landed in LDS by DMA while the previous P are processed (double buffer: 2 * P * 256 B per wave)
template <int P>
__global__ void __launch_bounds__(THREADS) stream_lds(float *slabs, int lines, int sweeps) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    float *S = slabs + (size_t) blockIdx.x * lines * 64;
    float *ring = lds + (size_t) w * 2 * P * 64;  // this wave's two buffers
    for (int s = 0; s < sweeps; ++s) {
        int buf = 0;
        // prologue: first batch
        int e0 = w * P;
        if (e0 < lines)
#pragma unroll
            for (int g = 0; g < P; ++g)
                if (e0 + g < lines) __builtin_amdgcn_global_load_lds(S + (size_t) (e0 + g) * 64 + lane, ring + (buf * P + g) * 64, 4, 0, 0);
        for (; e0 < lines; e0 += W * P) {
            __builtin_amdgcn_s_waitcnt(0);  // batch in LDS (and earlier stores retired)
            const int e1 = e0 + W * P;
            if (e1 < lines)  // next batch goes into the other buffer while this one is processed
#pragma unroll
                for (int g = 0; g < P; ++g)
                    if (e1 + g < lines)
                        __builtin_amdgcn_global_load_lds(S + (size_t) (e1 + g) * 64 + lane, ring + ((buf ^ 1) * P + g) * 64, 4, 0, 0);
#pragma unroll
            for (int g = 0; g < P; ++g)
                if (e0 + g < lines) {
                    const float x = ring[(buf * P + g) * 64 + lane] * 1.0001f + 1.0f;
                    S[(size_t) (e0 + g) * 64 + lane] = x;
                }
            buf ^= 1;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
}

// (C) as (B) but reading slab A and writing slab B (ping-pong, swapped every sweep), optionally with non-temporal accesses
template <int P, bool NT>
__global__ void __launch_bounds__(THREADS) stream_lds_pp(float *slabs, float *slabs2, int lines, int sweeps) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    float *S = slabs + (size_t) blockIdx.x * lines * 64, *D = slabs2 + (size_t) blockIdx.x * lines * 64;
    float *ring = lds + (size_t) w * 2 * P * 64;
    for (int s = 0; s < sweeps; ++s) {
        int buf = 0;
        int e0 = w * P;
        if (e0 < lines)
#pragma unroll
            for (int g = 0; g < P; ++g)
                if (e0 + g < lines) __builtin_amdgcn_global_load_lds(S + (size_t) (e0 + g) * 64 + lane, ring + (buf * P + g) * 64, 4, 0, NT ? 2 : 0);
        for (; e0 < lines; e0 += W * P) {
            __builtin_amdgcn_s_waitcnt(0);
            const int e1 = e0 + W * P;
            if (e1 < lines)
#pragma unroll
                for (int g = 0; g < P; ++g)
                    if (e1 + g < lines)
                        __builtin_amdgcn_global_load_lds(S + (size_t) (e1 + g) * 64 + lane, ring + ((buf ^ 1) * P + g) * 64, 4, 0, NT ? 2 : 0);
#pragma unroll
            for (int g = 0; g < P; ++g)
                if (e0 + g < lines) {
                    const float x = ring[(buf * P + g) * 64 + lane] * 1.0001f + 1.0f;
                    if (NT) __builtin_nontemporal_store(x, D + (size_t) (e0 + g) * 64 + lane);
                    else D[(size_t) (e0 + g) * 64 + lane] = x;
                }
            buf ^= 1;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        float *t = S;
        S = D;
        D = t;
    }
}

// (D) wide both ways: 1 KiB per load instruction (global_load_lds_dwordx4: 16 lanes per line), the result written back into
// the LDS copy and stored with global_store_dwordx4 (each lane reads 16 bytes of the lane-linear image back): a quarter of
// the vector-memory instructions of (B) on either side
template <int P>  // P lines per buffer, multiple of 4
__global__ void __launch_bounds__(THREADS) stream_lds_wide(float *slabs, int lines, int sweeps) {
    extern __shared__ float lds[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
    float *S = slabs + (size_t) blockIdx.x * lines * 64;
    float *ring = lds + (size_t) w * 2 * P * 64;
    for (int s = 0; s < sweeps; ++s) {
        int buf = 0;
        int e0 = w * P;
        if (e0 < lines)
#pragma unroll
            for (int g = 0; g < P; g += 4)
                if (e0 + g < lines) __builtin_amdgcn_global_load_lds(S + (size_t) (e0 + g) * 64 + lane * 4, ring + (buf * P + g) * 64, 16, 0, 0);
        for (; e0 < lines; e0 += W * P) {
            __builtin_amdgcn_s_waitcnt(0);
            const int e1 = e0 + W * P;
            if (e1 < lines)
#pragma unroll
                for (int g = 0; g < P; g += 4)
                    if (e1 + g < lines)
                        __builtin_amdgcn_global_load_lds(S + (size_t) (e1 + g) * 64 + lane * 4, ring + ((buf ^ 1) * P + g) * 64, 16, 0, 0);
#pragma unroll
            for (int g = 0; g < P; ++g)
                if (e0 + g < lines) ring[(buf * P + g) * 64 + lane] = ring[(buf * P + g) * 64 + lane] * 1.0001f + 1.0f;
#pragma unroll
            for (int g = 0; g < P; g += 4)
                if (e0 + g < lines) {
                    const f4 v = *reinterpret_cast<const f4 *>(ring + (buf * P + g) * 64 + lane * 4);
                    *reinterpret_cast<f4 *>(S + (size_t) (e0 + g) * 64 + lane * 4) = v;
                }
            buf ^= 1;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

// round 3: where the slabs live decides the rate once they exceed the Infinity Cache: 0 = hipMalloc (fast or slow per
// allocation), 1 = hipDeviceMallocContiguous (the slow mode), 2 = 64 MiB physical chunks mapped into one virtual range (what the
// streamed engine uses now)
static float *alloc_slabs(size_t bytes, int mode) {
    void *p = nullptr;
    if (mode == 1) {
        CHECK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous));
    } else if (mode == 2) {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        const size_t chunk = (size_t) 64 << 20, n = (bytes + chunk - 1) / chunk;
        CHECK(hipMemAddressReserve(&p, n * chunk, 0, nullptr, 0));
        for (size_t i = 0; i < n; i++) {
            hipMemGenericAllocationHandle_t h;
            CHECK(hipMemCreate(&h, chunk, &prop, 0));
            CHECK(hipMemMap((char *) p + i * chunk, chunk, 0, h, 0));
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CHECK(hipMemSetAccess(p, n * chunk, &acc, 1));
    } else {
        CHECK(hipMalloc(&p, bytes));
    }
    return (float *) p;
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, lines = argc > 2 ? atoi(argv[2]) : 1140, sweeps = argc > 3 ? atoi(argv[3]) : 50;
    const int amode = argc > 4 ? atoi(argv[4]) : 0;
    const size_t words = (size_t) blocks * lines * 64;
    float *slabs = alloc_slabs(words * 4, amode);
    printf("allocation: %s\n", amode == 1 ? "hipDeviceMallocContiguous" : (amode == 2 ? "64 MiB chunks mapped into one range (hipMemCreate / hipMemMap)" : "hipMalloc"));
    CHECK(hipMemset(slabs, 0, words * 4));
    const double bytes = 2.0 * words * 4 * sweeps;  // read + write
    printf("%d workgroups x %d lines x 256 B = %.0f MB of slabs, %d sweeps, %.1f GB moved per launch\n", blocks, lines, words * 4 / 1e6,
           sweeps, bytes / 1e9);
#define RUN_REGS(G)                                                                                               \
    {                                                                                                             \
        const double ms = time_ms([&] { hipLaunchKernelGGL(stream_regs<G>, dim3(blocks), dim3(THREADS), 0, 0, slabs, lines, sweeps); }, 3); \
        printf("VGPR staging, %2d lines per wave in flight: %7.2f ms  %6.2f TB/s\n", G, ms, bytes / ms / 1e9);   \
    }
#define RUN_LDS(P)                                                                                                \
    {                                                                                                             \
        const size_t lds = (size_t) (THREADS / 64) * 2 * P * 256;                                                 \
        CHECK(hipFuncSetAttribute((const void *) stream_lds<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
        const double ms = time_ms([&] { hipLaunchKernelGGL(stream_lds<P>, dim3(blocks), dim3(THREADS), lds, 0, slabs, lines, sweeps); }, 3); \
        printf("LDS-DMA staging, %2d lines per wave per buffer (%3zu KB LDS): %7.2f ms  %6.2f TB/s\n", P, lds / 1024, ms, bytes / ms / 1e9); \
    }
    RUN_REGS(8)
    RUN_REGS(16)
    RUN_REGS(24)
    RUN_REGS(32)
    RUN_LDS(8)
    RUN_LDS(16)
    RUN_LDS(24)
    RUN_LDS(32)
#define RUN_WIDE(P)                                                                                               \
    {                                                                                                             \
        const size_t lds = (size_t) (THREADS / 64) * 2 * P * 256;                                                 \
        CHECK(hipFuncSetAttribute((const void *) stream_lds_wide<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
        const double ms = time_ms([&] { hipLaunchKernelGGL(stream_lds_wide<P>, dim3(blocks), dim3(THREADS), lds, 0, slabs, lines, sweeps); }, 3); \
        printf("LDS-DMA dwordx4 in, dwordx4 out, %2d lines per wave per buffer (%3zu KB LDS): %7.2f ms  %6.2f TB/s\n", P, lds / 1024, ms, bytes / ms / 1e9); \
    }
    RUN_WIDE(8)
    RUN_WIDE(16)
    RUN_WIDE(32)
    float *slabs2 = alloc_slabs(words * 4, amode);
    CHECK(hipMemset(slabs2, 0, words * 4));
#define RUN_PP(P, NT, IP)                                                                                         \
    {                                                                                                             \
        const size_t lds = (size_t) (THREADS / 64) * 2 * P * 256;                                                 \
        CHECK(hipFuncSetAttribute((const void *) stream_lds_pp<P, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
        const double ms = time_ms([&] { hipLaunchKernelGGL((stream_lds_pp<P, NT>), dim3(blocks), dim3(THREADS), lds, 0, slabs, IP ? slabs : slabs2, lines, sweeps); }, 3); \
        printf("LDS-DMA staging, %2d lines per buffer, %s, %s: %7.2f ms  %6.2f TB/s\n", P, IP ? "in place" : "ping-pong (read A, write B)", NT ? "non-temporal" : "default policy", ms, bytes / ms / 1e9); \
    }
    RUN_PP(16, false, true)
    RUN_PP(16, false, false)
    RUN_PP(16, true, true)
    RUN_PP(16, true, false)
    return 0;  // (the process ends here: the allocations go with it)
}
