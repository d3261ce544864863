// ray_sort.hip -- optional reordering of a bounce's live paths before the extension traversal.
//
// After the first bounce the rays of neighbouring queue entries start all over the scene and point anywhere;
// sorting them by (origin cell, direction octant) makes the rays of a wave visit the same part of the
// hierarchy.  Everything downstream addresses a path by its position in the packed state and its radiance
// by the path number that travels with it, so the permutation changes no result (tested bit for bit).
// The price is a radix sort of (key, position) pairs and one pass that moves 52 bytes per path.
#include <hip/hip_runtime.h>

#include <cstring>  // rocPRIM's headers use memset without including it

#include <rocprim/rocprim.hpp>

#include "wavefront.h"

namespace {

#define RS_BLOCK 256

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 7 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(RS_BLOCK) void k_ray_keys(PathState ps, uint32_t count, float3 lo, float3 scale, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * RS_BLOCK + threadIdx.x;
    if (i >= count) return;
    const float4 o = ps.o_seed[i], d = ps.d_depth[i];
    // NaN and out-of-range origins land in the border cells
    const uint32_t cx = (uint32_t)fminf(fmaxf((o.x - lo.x) * scale.x, 0.f), 127.f);
    const uint32_t cy = (uint32_t)fminf(fmaxf((o.y - lo.y) * scale.y, 0.f), 127.f);
    const uint32_t cz = (uint32_t)fminf(fmaxf((o.z - lo.z) * scale.z, 0.f), 127.f);
    const uint32_t oct = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
    keys[i] = ((spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2)) << 3) | oct;
    vals[i] = i;
}

__global__ __launch_bounds__(RS_BLOCK) void k_ray_permute(PathState ps, uint32_t count, const uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * RS_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t from = order[i];
    ps.o_seed_next[i] = ps.o_seed[from];
    ps.d_depth_next[i] = ps.d_depth[from];
    ps.thr_next[i] = ps.thr[from];  // carries the path number (only camera segments keep it in `pid`, and those are never sorted)
}

}  // namespace

size_t ray_sort_temp_bytes(uint32_t capacity) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, (size_t)capacity, 0, 24, (hipStream_t) nullptr) != hipSuccess) return 0;
    return bytes;
}

hipError_t ray_sort(hipStream_t s, const PathState& ps, uint32_t count, const float lo[3], const float hi[3], uint32_t* k_in, uint32_t* k_out, uint32_t* v_in,
                    uint32_t* v_out, void* temp, size_t temp_bytes) {
    if (count == 0) return hipSuccess;
    float3 l = make_float3(lo[0], lo[1], lo[2]), sc;
    sc.x = hi[0] > lo[0] ? 128.f / (hi[0] - lo[0]) : 0.f;
    sc.y = hi[1] > lo[1] ? 128.f / (hi[1] - lo[1]) : 0.f;
    sc.z = hi[2] > lo[2] ? 128.f / (hi[2] - lo[2]) : 0.f;
    const dim3 g((count + RS_BLOCK - 1) / RS_BLOCK), b(RS_BLOCK);
    hipLaunchKernelGGL(k_ray_keys, g, b, 0, s, ps, count, l, sc, k_in, v_in);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (size_t)count, 0, 24, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ray_permute, g, b, 0, s, ps, count, v_out);
    return hipGetLastError();
}
