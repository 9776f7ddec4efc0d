// Which SIMD does wavefront w of a 4-wavefront workgroup run on?  (HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8],
// SE [15:13]; XCC_ID separate.)  Persistent-style launch: 1280 workgroups of 256 threads, 30 KB of LDS each, so five are
// resident per CU, as in the QP-ADMM workgroup kernel.  Prints the histogram of SIMD ids per wavefront index and of
// the SIMD of wavefront 0 over the workgroups resident on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ void probe(uint32_t *out, int spin) {
    extern __shared__ unsigned char smem[];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;  // stay resident long enough for the CU to fill
    smem[threadIdx.x] = (unsigned char) x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (xcc & 0xF) | ((uint32_t) smem[5] << 31);
    }
}
int main() {
    const int nb = 1280;
    uint32_t *d;
    hipMalloc(&d, nb * 8 * sizeof(uint32_t));
    hipFuncSetAttribute((const void *) probe, hipFuncAttributeMaxDynamicSharedMemorySize, 30 * 1024);
    probe<<<nb, 256, 30 * 1024>>>(d, 2000000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    std::map<uint32_t, std::vector<int>> per_cu;  // (xcc, se, cu) -> SIMD of wavefront 0 of each resident workgroup
    int rr = 0;
    for (int b = 0; b < nb; ++b) {
        int s0 = -1;
        bool rot = true;
        for (int w = 0; w < 4; ++w) {
            const uint32_t hw = h[(b * 4 + w) * 2];
            const int simd = (hw >> 4) & 3;
            hist[w][simd]++;
            if (w == 0) s0 = simd;
            else if (simd != ((s0 + w) & 3)) rot = false;
        }
        rr += rot;
        const uint32_t hw = h[b * 8], xcc = h[b * 8 + 1] & 0xF;
        per_cu[(xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xF)].push_back(s0);
    }
    printf("workgroups whose wavefront w sits on SIMD (s0 + w) mod 4: %d of %d\n", rr, nb);
    for (int w = 0; w < 4; ++w) printf("wavefront %d on SIMD 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int pat[6] = {};
    int shown = 0;
    for (auto &kv : per_cu) {
        int c[4] = {};
        for (int s : kv.second) c[s]++;
        int mx = 0;
        for (int s = 0; s < 4; ++s) mx = c[s] > mx ? c[s] : mx;
        pat[mx > 5 ? 5 : mx]++;
        if (shown++ < 6) {
            printf("CU %05x: %zu workgroups, SIMD of wavefront 0:", kv.first, kv.second.size());
            for (int s : kv.second) printf(" %d", s);
            printf("\n");
        }
    }
    printf("%zu CUs; most wavefront-0s on one SIMD of a CU = 1: %d  2: %d  3: %d  4: %d  5+: %d\n", per_cu.size(), pat[1], pat[2], pat[3], pat[4], pat[5]);
    return 0;
}
