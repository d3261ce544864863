// Micro-benchmark of the vector memory pipeline for scattered node-like reads: how the cost of a
// wave-level load depends on its width, on the number of active lanes and on the working set.
// build: hipcc --offload-arch=gfx950 -O3 -o ta_bench ta_bench.hip ; run: ./ta_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Each lane walks a pseudo-random chain of 64-byte records; PARTS = float4 loads per record,
// LANES = active lanes per wave.  The next index depends on the loaded data (like a traversal).
template <int PARTS, int WIDTH>
__global__ __launch_bounds__(256) void k_chase(const uint4* __restrict__ rec, uint32_t mask, int steps, int lanes, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    if ((int)lane < lanes) {
        for (int s = 0; s < steps; ++s) {
            const uint4* p = rec + (size_t)(idx & mask) * 4;
            uint32_t v = 0;
            if (WIDTH == 4) {
#pragma unroll
                for (int k = 0; k < PARTS; ++k) { const uint4 q = p[k]; v += q.x ^ q.y ^ q.z ^ q.w; }
            } else if (WIDTH == 2) {
                const uint2* p2 = reinterpret_cast<const uint2*>(p);
#pragma unroll
                for (int k = 0; k < PARTS; ++k) { const uint2 q = p2[k]; v += q.x ^ q.y; }
            } else {
                const uint32_t* p1 = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
                for (int k = 0; k < PARTS; ++k) v += p1[k];
            }
            acc += v;
            idx = idx * 1664525u + 1013904223u + (v & 1u);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// The same chase with every load issued twice, once for each half of the wave (exec-masked): does a
// load with 32 distinct lines cost less than half of one with 64?
template <int PARTS>
__global__ __launch_bounds__(256) void k_chase_split(const uint4* __restrict__ rec, uint32_t mask, int steps, int groups, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    const uint32_t per = 64u / (uint32_t)groups;
    for (int s = 0; s < steps; ++s) {
        const uint4* p = rec + (size_t)(idx & mask) * 4;
        uint32_t v = 0;
        for (int g = 0; g < groups; ++g) {
            if (lane / per == (uint32_t)g) {
#pragma unroll
                for (int k = 0; k < PARTS; ++k) { const uint4 q = p[k]; v += q.x ^ q.y ^ q.z ^ q.w; }
            }
            asm volatile("" ::: "memory");
        }
        acc += v;
        idx = idx * 1664525u + 1013904223u + (v & 1u);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int PARTS>
static void run_split(const uint4* d_rec, uint32_t n_rec, int groups, int blocks_per_cu, uint32_t* d_out, int cus) {
    const int steps = 2000, blocks = cus * blocks_per_cu;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_chase_split<PARTS>), dim3(blocks), dim3(256), 0, 0, d_rec, n_rec - 1, 200, groups, d_out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_chase_split<PARTS>), dim3(blocks), dim3(256), 0, 0, d_rec, n_rec - 1, steps, groups, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_steps = (double)blocks * 4 * steps;
    printf("%d x dwordx4 in %d lane groups, %2d waves per CU, records %8u: %7.3f ms, %7.2f ns per wave step per CU (%6.3f ns per lane record)\n", PARTS, groups,
           blocks_per_cu * 4, n_rec, ms, ms * 1e6 / (wave_steps / cus), ms * 1e6 / (wave_steps / cus) / 64.0);
}

template <int PARTS, int WIDTH>
static void run(const char* name, const uint4* d_rec, uint32_t n_rec, int lanes, uint32_t* d_out, int cus) {
    const int steps = 2000, blocks = cus * 6;  // 24 waves per CU
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_chase<PARTS, WIDTH>), dim3(blocks), dim3(256), 0, 0, d_rec, n_rec - 1, 200, lanes, d_out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_chase<PARTS, WIDTH>), dim3(blocks), dim3(256), 0, 0, d_rec, n_rec - 1, steps, lanes, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr = (double)blocks * 4 * steps * PARTS;
    const double per_cu_ns = ms * 1e6 / (wave_instr / cus);
    printf("%-28s records %8u (%6.1f MB) lanes %2d: %7.3f ms, %6.2f ns per wave-load per CU, %7.1f GB/s useful\n", name, n_rec, n_rec * 64.0 / 1e6, lanes, ms,
           per_cu_ns, (double)blocks * 256 / 64 * lanes * steps * PARTS * WIDTH * 4 / (ms * 1e6));
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    const uint32_t max_rec = 1u << 22;  // 256 MB
    std::vector<uint4> h(max_rec * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint4((uint32_t)i * 747796405u, (uint32_t)i, 7u, 9u);
    uint4* d_rec;
    uint32_t* d_out;
    CHECK(hipMalloc((void**)&d_rec, h.size() * sizeof(uint4)));
    CHECK(hipMalloc((void**)&d_out, 64));
    CHECK(hipMemcpy(d_rec, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    for (uint32_t n_rec : {128u, 1u << 15, 1u << 17, 1u << 22}) {  // 8 KB (L1), 2 MB (L2), 8 MB, 256 MB
        for (int lanes : {64, 32, 8}) {
            run<1, 4>("1 x dwordx4", d_rec, n_rec, lanes, d_out, cus);
            run<2, 4>("2 x dwordx4", d_rec, n_rec, lanes, d_out, cus);
            run<4, 4>("4 x dwordx4 (one node)", d_rec, n_rec, lanes, d_out, cus);
            run<4, 2>("4 x dwordx2", d_rec, n_rec, lanes, d_out, cus);
            run<4, 1>("4 x dword", d_rec, n_rec, lanes, d_out, cus);
        }
    }
    printf("\nall 64 lanes, the load split over lane groups / fewer waves per CU\n");
    for (uint32_t n_rec : {1u << 15, 1u << 17}) {
        for (int bpc : {6, 3, 2}) {
            for (int groups : {1, 2, 4}) {
                run_split<1>(d_rec, n_rec, groups, bpc, d_out, cus);
                run_split<4>(d_rec, n_rec, groups, bpc, d_out, cus);
            }
        }
    }
    return 0;
}
