// valu_peak.hip -- how many wave64 VALU instructions per second a SIMD of this GPU issues.
// Every wave runs N independent v_fma_f32 chains in a loop; the grid puts W waves on every SIMD.
// Prints, per W: wave instructions per second per SIMD, the shader clock held during the loop
// (s_memtime ticks per s_memrealtime tick x 100 MHz) and the resulting cycles per instruction.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/micro/valu_peak tools/micro/valu_peak.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define ITER 8192
#define CHAINS 16

__global__ void k_fma(float* out, unsigned long long* clk, float a, float b) {
    float x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = (float)(threadIdx.x + i);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;  // keeps the chains alive
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, 4);
    const int max_blocks = cus * 8;
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * max_blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("CUs %d\n", cus);
    for (int w : {1, 2, 4, 8}) {
        // 256-thread blocks = 4 waves = one per SIMD; w blocks per CU
        const int blocks = cus * w;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, clk, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, clk, 1.0001f, 0.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        std::vector<double> ghz, cyc;
        for (int b = 0; b < blocks; ++b) {
            if (h[2 * b + 1] == 0) continue;
            ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
            cyc.push_back((double)h[2 * b] / ((double)ITER * CHAINS));
        }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cyc.begin(), cyc.end());
        const double insts = (double)blocks * 4.0 * ITER * CHAINS * reps;
        const double per_simd = insts / (ms * 1e-3) / (cus * 4.0);
        printf("waves/SIMD %d: %.3f G wave-instructions/s per SIMD (wall clock, incl. launch gaps); in-kernel clock %.3f GHz; "
               "cycles per instruction of one wave %.2f -> cycles per instruction of the SIMD %.2f\n",
               w, per_simd / 1e9, ghz[ghz.size() / 2], cyc[cyc.size() / 2], cyc[cyc.size() / 2] / w);
    }
    return 0;
}
