// wavefront.h -- host-side launchers of the wavefront kernels (wavefront.hip)
// and of the GPU BVH build (bvh_build.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "device_scene.h"

void wf_generate(hipStream_t s, const RenderParams& rp, const PathState& ps, uint32_t npix_pad, uint32_t n_paths, uint32_t sample_begin,
                 const float* acc, uint32_t* segctr, uint32_t* pixseed /* npix_pad words of scratch */, int seed_L_from_acc);
// closest hits of the live paths (entries [0, count) of the packed path state); `work_counter`
// is a zeroed device word the refilling waves reserve their spans from; `coherent` = camera rays
void wf_extend(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t count, uint32_t* work_counter,
               uint32_t resident_waves, bool coherent, bool count_trav, unsigned long long* trav_counters);
uint32_t wf_resident_waves(int device);
// positions of the paths that hit something (misses end there)
void wf_compact_hits(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t count, uint32_t* hit_queue, uint32_t* hit_count);
// Everything of a bounce's closest-hit programs that does not depend on shadow-ray
// visibility: light samples (terms + shadow rays, their count in `shadow_count`), emission,
// next direction / throughput, survivors -> the `*_next` path state, packed; segments whose radiance needs the
// light terms -> gather_queue.
// surface / shadow / gather read their queue length from device memory; `max_*` sizes the grid
// `queue` == nullptr: no hit queue -- the kernel looks at all `max_count` live paths, ends the ones that missed itself and
// adds the number of hits to *count_ptr (which must be zero before) by way of `block_tally`, a scratch array of one word per
// block of the launch (at least max_count / 256 + 1 words); only where wf_surface_can_skip_queue().
bool wf_surface_can_skip_queue(const RenderParams& rp);
void wf_surface(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, const uint32_t* queue, uint32_t* count_ptr,
                uint32_t max_count, uint32_t npix_pad, uint32_t sample_begin, uint32_t* segctr, uint32_t* next_count,
                uint32_t* gather_queue, uint32_t* gather_count, uint32_t* shadow_count, bool first_bounce /* camera segments: throughput 1 is implied */,
                uint32_t* block_tally);
void wf_shadow(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, const uint32_t* count_ptr, uint64_t max_rays,
               uint32_t* work_counter, uint32_t resident_waves, bool count_trav, unsigned long long* trav_counters, unsigned long long* shadow_total);
void wf_gather(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, const uint32_t* gather_queue, const uint32_t* gather_count,
               uint32_t max_count);
void wf_accumulate(hipStream_t s, const RenderParams& rp, const PathState& ps, uint32_t npix_pad, uint32_t n_samples, float* acc, int replace);
void wf_resolve(hipStream_t s, const float* acc, float* result, uint32_t n_pixels, int spp, float gamma, int frame_id);
void wf_hit_records(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t count, PtcHitRecord* out);

// test hooks: closest hits of entries [0, count) as (global primitive index, t); the shading helpers on raw inputs
void wf_probe_hits(hipStream_t s, const DevScene& sc, const PathState& ps, uint32_t count, int32_t* prim, float* t);
void wf_eval_shading(hipStream_t s, int fn, uint32_t n, const float* in, float* out);

// ---- ray sorting (ray_sort.hip) ------------------------------------------------
// Reorders the live paths (entries [0, count) of o_seed / d_depth / thr / pid -> the same entries of the `*_next`
// arrays, which the caller then swaps in) by a 24-bit key: Morton code of the ray origin's cell on a 128^3 grid
// over [lo, hi] (21 bits) | direction octant (3 bits), so that the rays a wave traces start close together and
// point the same way.  `k_in`, `k_out`, `v_in`, `v_out`: four scratch arrays of `count` words; `temp`: rocPRIM's
// work space of ray_sort_temp_bytes(capacity) bytes.  Returns hipSuccess or the first error.
size_t ray_sort_temp_bytes(uint32_t capacity);
hipError_t ray_sort(hipStream_t s, const PathState& ps, uint32_t count, const float lo[3], const float hi[3], uint32_t* k_in, uint32_t* k_out, uint32_t* v_in,
                    uint32_t* v_out, void* temp, size_t temp_bytes);

// ---- BVH build (bvh_build.hip) ----------------------------------------------
struct BvhBuildInput {
    const PtcTriangle* d_tris;   // device copies of the AoS uploads
    uint32_t n_tris;
    const PtcSphere* d_spheres;
    uint32_t n_spheres;
    const uint32_t* h_tri_mat;   // host: material id | light bit per triangle / sphere (upload order)
    const uint32_t* h_sph_mat;
    float extra_extent;          // |coordinate| bound of ray origins outside the geometry (the eye)
    int builder;                 // 0 = LBVH (Karras radix tree), 1 = PLOC (locally-ordered clustering), 2 = surface-area sweep on the host, 4 = the same sweep on the device
};

struct BvhBuildOutput {  // device allocations owned by the caller afterwards
    float4* tri_q = nullptr;
    float4* tri_nm = nullptr;    // per triangle slot: (normal.xyz, material word): all the surface kernel needs of a triangle, in one 16-byte record
    uint32_t* tri_mat = nullptr;
    int32_t* tri_prim = nullptr;
    DevSphere* spheres = nullptr;
    uint32_t* sph_mat = nullptr;
    int32_t* sph_prim = nullptr;
    BvhNode* nodes = nullptr;
    uint32_t n_nodes = 0;
    uint32_t max_depth = 0;
    WideNode* wnodes = nullptr;  // four-wide version of the hierarchy; null when it could not be built
    uint32_t n_wnodes = 0;
    uint32_t wide_depth = 0;
    // box of the primitives that went into the trees (without the few "large" ones, e.g. a ground plane a
    // thousand times the size of everything else): the grid the ray-sort keys are taken on
    float sort_lo[3] = {0.f, 0.f, 0.f}, sort_hi[3] = {1.f, 1.f, 1.f};
    // A scene with one or two spheres beside its triangles (the Cornell boxes): the reference of the rest of the
    // hierarchy, so that every ray can test the spheres first and then traverse without them; REF_EMPTY_WORD otherwise.
    uint32_t root_without_spheres = REF_EMPTY_WORD;
    // The large triangles of the scene (at most three, in one leaf under the root) can be tested for every ray up front: their
    // slots, and where the traversal starts without them -- in the two-box nodes and in the four-wide nodes (an extra node after
    // the hierarchy's own: the root with that leaf's child slot emptied).  big_count = 0: not available.
    uint32_t big_first = 0, big_count = 0;
    int32_t big_prim[3] = {-1, -1, -1};  // their global triangle indices
    uint32_t root_without_big = REF_EMPTY_WORD, wide_root_without_big = REF_EMPTY_WORD;
    double host_ms = 0.0;        // wall-clock time of the build's passes over primitives / nodes that ran in host code
};

// Returns an empty string on success, else an error message.
std::string bvh_build(hipStream_t s, const BvhBuildInput& in, BvhBuildOutput& out);
void bvh_free(BvhBuildOutput& out);
// Host part of the build on its own: binary nodes -> four-wide nodes.  0 = done, 1 = not representable
// (non-finite box, too deep), 2 = `capacity` too small, 3 = malformed input (a child index outside the
// array, or a node reached twice).
int bvh_collapse_wide_host(const BvhNode* bin, uint32_t n_nodes, WideNode* out, uint32_t capacity, uint32_t* n_wide, uint32_t* depth);
