// wavefront.hip -- the wavefront path-tracing kernels for gfx950.
//
//   k_generate   camera rays for a batch of (pixel, sample) paths
//                (PinholeCamera.cu:37-51), 8x8 pixel tile per 64-lane wave
//   k_extend_static / k_extend
//                closest hit of every active path (rtTrace ray type 0:
//                Triangle.cu / Sphere.cu over the BVH); LDS-resident stack.
//                The static kernel (one ray per lane, tight loop) serves
//                hierarchies of a handful of nodes; in the refilling kernel
//                resident waves reserve spans of the queue from one global
//                counter and lanes that finish pull the next ray of the span
//   k_trace_deferred<false>  refilling waves whose lanes queue leaf primitives in
//                LDS and test them 64 at a time, over two-child or four-wide
//                nodes (the default for every hierarchy that is not tiny)
//   k_surface    one thread per path after the extension: a miss ends the path
//                with the background colour (Common.cu:13-18); at a hit, the part of
//                the chosen integrator's closest-hit program (RayTracer.cu) that
//                does not depend on shadow-ray visibility -- every light sample's
//                shadow ray and f*G term (rays whose term is exactly 0 are dropped,
//                the others queued), emission, the sampled direction, throughput,
//                Russian roulette -- plus the bounce-loop bookkeeping of
//                PinholeCamera.cu:54-66; survivors are compacted into the next
//                queue (wave64 ballot + popcount prefix, one atomic per block)
//   k_shadow_static / k_shadow / k_trace_deferred<true>
//                any-hit traversal of the queued shadow rays (rtTrace ray type 1,
//                Common.cu:30-36); an occluded ray zeroes its term
//   k_gather     radiance of the segments that sampled the lights: folds the terms
//                in the order the reference's loops add them
//   k_accumulate per-pixel sum of the batch's samples, in sample order
//   k_resolve    average, gamma, frame lerp (PinholeCamera.cu:70-80)
//
// All arithmetic that decides a hit (t,u,v) is plain IEEE fp32 in the
// reference's operation order; compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "intersect.h"
#include "shading.h"
#include "wavefront.h"

#include <algorithm>

#define BLOCK TRAV_BLOCK
#ifndef PT_SHADING_FAST
#define PT_SHADING_FAST 1   /* 0: correctly rounded division / square root in every shading kernel (A/B builds) */
#endif
#define GEN_BLOCK 1024   /* generate / shade: one queue atomic per 1024 paths */

// ---------------------------------------------------------------- helpers --

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// Map a padded pixel number to image coordinates: 64 consecutive numbers form
// an 8x8 tile, so one wave traces a compact bundle of camera rays.
// n / d and n % d for a divisor that is the same for the whole launch: one fp64 multiply
// by the host-computed reciprocal (the product is within one of the quotient for any
// 32-bit n) and a fix-up, instead of the ~40-instruction integer division sequence.
__device__ __forceinline__ uint32_t div_by(uint32_t n, uint32_t d, double inv_d, uint32_t& rem) {
    uint32_t q = (uint32_t)((double)n * inv_d);
    int64_t r = (int64_t)n - (int64_t)q * d;
    if (r < 0) {
        --q;
        r += d;
    } else if (r >= (int64_t)d) {
        ++q;
        r -= d;
    }
    rem = (uint32_t)r;
    return q;
}

// Path numbers within a batch: the samples of one 8x8 pixel tile are consecutive
// ((tile * samples + s) * 64 + pixel in tile), so the waves running at any one time trace
// the samples of a small patch of the image and share their part of the hierarchy in the
// caches -- rather than one sample of a fifth of the image.
__device__ __forceinline__ uint32_t path_number(const RenderParams& rp, uint32_t pix, uint32_t s) {
    return (((pix >> 6) * rp.batch_samples + s) << 6) | (pix & 63u);
}
__device__ __forceinline__ void path_split(const RenderParams& rp, uint32_t p, uint32_t& pix, uint32_t& s) {
    const uint32_t tile = div_by(p >> 6, rp.batch_samples, rp.inv_batch_samples, s);
    pix = (tile << 6) | (p & 63u);
}

// Padded pixel number -> image coordinates.  64 consecutive numbers are an 8x8 pixel tile
// (one wave traces a compact bundle of camera rays), 64 consecutive tiles an 8x8 block of
// tiles, and the blocks go row by row: consecutive numbers stay close in both directions.
__device__ __forceinline__ bool pixel_of(const RenderParams& rp, uint32_t pix, uint32_t& x, uint32_t& y) {
    const uint32_t blocks_x = (rp.width + 63u) >> 6;
    const uint32_t in = pix & 63u, tile = (pix >> 6) & 63u, block = pix >> 12;
    uint32_t bx;
    const uint32_t by = div_by(block, blocks_x, rp.inv_tiles_x, bx);
    x = (bx << 6) | ((tile & 7u) << 3) | (in & 7u);
    y = (by << 6) | ((tile >> 3) << 3) | (in >> 3);
    return x < rp.width && y < rp.height;
}

// Number of image pixels among the padded pixel numbers below tile `tile_global` (= pix >> 6), and the width and
// height of that tile's part of the image (0 x 0 ... 8 x 8).  Only the last column and the last row of 64x64 blocks
// are cut by the image border.
__device__ __forceinline__ uint32_t pixels_before_tile(const RenderParams& rp, uint32_t tile_global, uint32_t& tw, uint32_t& th) {
    const uint32_t blocks_x = (rp.width + 63u) >> 6;
    const uint32_t tile = tile_global & 63u, block = tile_global >> 6;
    uint32_t bx;
    const uint32_t by = div_by(block, blocks_x, rp.inv_tiles_x, bx);
    const uint32_t vw = min(64u, rp.width - (bx << 6)), vh = min(64u, rp.height - (by << 6));  // the block's part of the image
    const uint32_t x0 = (tile & 7u) << 3, y0 = (tile >> 3) << 3;                               // the tile's corner in the block
    tw = vw > x0 ? min(8u, vw - x0) : 0u;
    th = vh > y0 ? min(8u, vh - y0) : 0u;
    return (by << 6) * rp.width + (bx << 6) * vh + min(y0, vh) * vw + min(x0, vw) * th;
}

// Reserve one queue position for every thread with keep=true and return it.  Wave64
// ballot + popcount give the position inside the wave, the waves' counts meet in LDS
// and ONE atomic per block reserves the block's span (a per-wave atomic on the single
// counter word saturates the atomic unit at these queue sizes).  Must be reached by
// every thread of the block; may be called repeatedly.
__device__ __forceinline__ uint32_t queue_reserve(bool keep, uint32_t* __restrict__ counter) {
    __shared__ uint32_t s_cnt[16];
    __shared__ uint32_t s_base;
    const unsigned long long mask = __ballot(keep);
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63u) >> 6;
    const uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();  // the previous call's readers are done with s_cnt / s_base
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < nwaves; ++w) total += s_cnt[w];
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
    for (uint32_t w = 0; w < wave; ++w) base += s_cnt[w];
    return base + prefix;
}

// Append `value` for every thread with keep=true.
__device__ __forceinline__ void queue_append(bool keep, uint32_t value, uint32_t* __restrict__ out, uint32_t* __restrict__ counter) {
    const uint32_t at = queue_reserve(keep, counter);
    if (keep) out[at] = value;
}

// N streams on ONE counter with one atomic per block: all of stream 0 of the block, then
// all of stream 1, ...  A block that handles N chunks of consecutive queue entries this
// way keeps the entries in order and issues N times fewer atomics (the same-address
// atomic rate, about 88 per microsecond, is what bounds the cheap kernels).
template <int N>
__device__ __forceinline__ void queue_reserve_streams(const bool keep[N], uint32_t* __restrict__ counter, uint32_t at[N]) {
    __shared__ uint32_t s_cnt[16][N];  // per wave and stream: entries, then their first position
    __shared__ uint32_t s_tot[N];      // per stream: entries of the block, then the stream's first position
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63u) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long m[N];
#pragma unroll
    for (int k = 0; k < N; ++k) m[k] = __ballot(keep[k]);
    __syncthreads();  // the previous call's readers are done
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) s_cnt[wave][k] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)N) {  // one thread per stream: its total
        uint32_t t = 0;
        for (uint32_t w = 0; w < nwaves; ++w) t += s_cnt[w][threadIdx.x];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // the block's span, split among the streams
        uint32_t total = 0;
        for (int k = 0; k < N; ++k) total += s_tot[k];
        uint32_t run = total ? atomicAdd(counter, total) : 0u;
        for (int k = 0; k < N; ++k) {
            const uint32_t t = s_tot[k];
            s_tot[k] = run;
            run += t;
        }
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)N) {  // one thread per stream: where each wave's entries start
        uint32_t run = s_tot[threadIdx.x];
        for (uint32_t w = 0; w < nwaves; ++w) {
            const uint32_t c = s_cnt[w][threadIdx.x];
            s_cnt[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) at[k] = keep[k] ? s_cnt[wave][k] + (uint32_t)__popcll(m[k] & lt) : 0u;
}

// The same for four streams at once, with one round of barriers: streams 0 and 1 go to
// `c01` (all of stream 0 of the block, then all of stream 1), stream 2 to `c2`, stream 3
// to `c3`.  In two halves, so that the caller can put work that does not depend on the positions
// between them: the block's three atomics are in flight meanwhile (a reservation costs a block the round
// trip to the counter word, one to two microseconds with every block of the grid asking the same word).
// Once per kernel (the halves share LDS words that are not guarded against a second call).
struct Reserve4 {
    unsigned long long m0, m1, m2, m3;  // who appends to which stream (wave-uniform)
    uint32_t base, first;               // threads 0..2: the block's first position on their counter; thread 0: entries of stream 0
};
struct Reserve4Lds {
    uint32_t cnt[16][5];
    uint32_t base[4];
};

// `tally` / `c_tally` (optional): a fifth count of which only the block's total is wanted: written to c_tally[blockIdx.x].
__device__ __forceinline__ void reserve4_begin(Reserve4& r, Reserve4Lds& L, bool k0, bool k1, bool k2, bool k3, uint32_t* __restrict__ c01, uint32_t* __restrict__ c2,
                                               uint32_t* __restrict__ c3, bool tally = false, uint32_t* __restrict__ c_tally = nullptr) {
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63u) >> 6;
    r.m0 = __ballot(k0);
    r.m1 = __ballot(k1);
    r.m2 = __ballot(k2);
    r.m3 = __ballot(k3);
    r.base = r.first = 0u;
    if (lane == 0) {
        L.cnt[wave][0] = (uint32_t)__popcll(r.m0);
        L.cnt[wave][1] = (uint32_t)__popcll(r.m1);
        L.cnt[wave][2] = (uint32_t)__popcll(r.m2);
        L.cnt[wave][3] = (uint32_t)__popcll(r.m3);
    }
    if (c_tally) {
        const unsigned long long m4 = __ballot(tally);
        if (lane == 0) L.cnt[wave][4] = (uint32_t)__popcll(m4);
    }
    __syncthreads();
    if (c_tally && threadIdx.x == 3u) {  // a plain store per block (summed by k_sum_words afterwards): a second atomic request per block would cost as much as the first
        uint32_t t = 0;
        for (uint32_t w = 0; w < nwaves; ++w) t += L.cnt[w][4];
        c_tally[blockIdx.x] = t;
    }
    if (threadIdx.x < 3u) {
        uint32_t t0 = 0, t1 = 0;
        const uint32_t k = threadIdx.x == 0u ? 0u : threadIdx.x + 1u;
        for (uint32_t w = 0; w < nwaves; ++w) {
            t0 += L.cnt[w][k];
            if (threadIdx.x == 0u) t1 += L.cnt[w][1];
        }
        uint32_t* c = threadIdx.x == 0u ? c01 : threadIdx.x == 1u ? c2 : c3;
        r.base = (t0 + t1) ? atomicAdd(c, t0 + t1) : 0u;
        r.first = t0;
    }
}

// at[k] receives the position for the streams this thread appends to.
__device__ __forceinline__ void reserve4_end(const Reserve4& r, Reserve4Lds& L, uint32_t at[4]) {
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (threadIdx.x < 3u) {
        const uint32_t k = threadIdx.x == 0u ? 0u : threadIdx.x + 1u;
        L.base[k] = r.base;
        if (threadIdx.x == 0u) L.base[1] = r.base + r.first;
    }
    __syncthreads();
    uint32_t b0 = L.base[0], b1 = L.base[1], b2 = L.base[2], b3 = L.base[3];
    for (uint32_t w = 0; w < wave; ++w) {
        b0 += L.cnt[w][0];
        b1 += L.cnt[w][1];
        b2 += L.cnt[w][2];
        b3 += L.cnt[w][3];
    }
    at[0] = b0 + (uint32_t)__popcll(r.m0 & lt);
    at[1] = b1 + (uint32_t)__popcll(r.m1 & lt);
    at[2] = b2 + (uint32_t)__popcll(r.m2 & lt);
    at[3] = b3 + (uint32_t)__popcll(r.m3 & lt);
}

__device__ __forceinline__ uint32_t segment_seed(const RenderParams& rp, uint32_t key0, uint32_t sample, uint32_t bounce, uint32_t* segctr, uint32_t pix) {
    if (rp.seeding == 0) {  // PinholeCamera.cu:56, i++ over the pixel's whole history
        uint32_t i = segctr[pix];
        segctr[pix] = i + 1u;
        return tea16(key0, i);
    }
    return tea16(key0, (sample << 12) | bounce);
}

// (origin | seed) and (direction | depth) of the segment at position `at` of the path state.  Camera segments all
// start at the eye with depth 0, so k_generate writes one record per path for them, (direction | seed), and not two.
// A ray can carry one bit from the kernel that made it to the kernel that traces it: "none of the scene's large triangles
// (DevScene::big_count) can be hit" -- k_surface has run the first half of the triangle test on them.  Later segments: the top
// bit of the depth word; shadow rays: the sign of the stored tmax (a tmax that is not positive is stored as +0: nothing lies
// in (tmin, tmax) either way).  The loaders below take the bit out, so every other reader sees plain values.
#define SEG_SKIP_BIG 0x80000000u
__device__ __forceinline__ void load_segment(const RenderParams& rp, const PathState& ps, uint32_t at, float4& o4, float4& d4, bool* skip_big = nullptr) {
    d4 = ps.d_depth[at];
    if (rp.camera_rays) {
        o4 = make_float4(rp.cf.eye.x, rp.cf.eye.y, rp.cf.eye.z, d4.w);
        d4.w = u2f(0u);
        if (skip_big) *skip_big = false;
    } else {
        o4 = ps.o_seed[at];
        if (skip_big) *skip_big = (f2u(d4.w) & SEG_SKIP_BIG) != 0u;
        d4.w = u2f(f2u(d4.w) & ~SEG_SKIP_BIG);
    }
}
// Shadow ray idx: o4 = (origin | tmax), d4 = (direction | slot of the ray's term).
__device__ __forceinline__ void load_shadow_ray(const PathState& ps, uint32_t idx, float4& o4, float4& d4, bool* skip_big = nullptr) {
    o4 = ps.sh_o[idx];
    d4 = ps.sh_d[idx];
    if (skip_big) *skip_big = (f2u(o4.w) & 0x80000000u) != 0u;
    o4.w = fabsf(o4.w);
}
__device__ __forceinline__ float shadow_tmax_word(float tmax, bool skip_big) {
    const float t = tmax > 0.f ? tmax : (tmax != tmax ? tmax : 0.f);  // NaN stays NaN (such a ray accepts nothing wherever it starts)
    return skip_big ? -t : t;
}

// ---------------------------------------------------------------- generate --
#define GEN_CHUNKS 4   /* chunks of GEN_BLOCK queue entries per block of k_compact_hits */

// Start of every pixel's jitter stream, tea<16>(index * frameID, 0) (PinholeCamera.cu:23-24): the same
// for all samples of the pixel, so it is computed once per frame and not once per path.
__global__ __launch_bounds__(BLOCK) void k_pixel_seeds(RenderParams rp, uint32_t npix_pad, uint32_t* __restrict__ pixseed) {
    const uint32_t pix = blockIdx.x * BLOCK + threadIdx.x;
    if (pix >= npix_pad) return;
    uint32_t x, y;
    pixseed[pix] = pixel_of(rp, pix, x, y) ? tea16((x * rp.height + y) * rp.frame_id, 0u) : 0u;
}

#ifndef CAM_BLOCK
#define CAM_BLOCK 256   /* no barrier and no atomic in this kernel: small blocks, so that a CU never waits for sixteen wave slots at once */
#endif
__global__ __launch_bounds__(CAM_BLOCK) void k_generate(RenderParams rp, PathState ps, uint32_t npix_pad, uint32_t n_paths, uint32_t sample_begin,
                                                    const float* __restrict__ acc, uint32_t* __restrict__ segctr, const uint32_t* __restrict__ pixseed,
                                                    int seed_L_from_acc) {
    const uint32_t p = blockIdx.x * CAM_BLOCK + threadIdx.x;
    if (p >= n_paths) return;  // whole waves: the number of paths is a multiple of 64
    uint32_t x = 0, y = 0, pix = 0, j = 0;
    path_split(rp, p, pix, j);
    const bool valid = pixel_of(rp, pix, x, y);
    // The 64 paths of a wave are one sample of one 8x8 tile (path_number): telling the compiler so turns the
    // jump-ahead of the jitter stream below, a loop of 32-bit multiplies, and the position below into scalar code.
    const uint32_t js = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);  // sample within the batch
    j = js + sample_begin;
    // Live paths are kept packed, in path-number order: this one's state goes to position `at`, the number of
    // valid paths below it.  Computed, not reserved from a counter: the rate of atomics on one word (about 88
    // blocks per microsecond) would bound this kernel, and the host knows the count without reading it back.
    uint32_t tw, th;
    const uint32_t before = pixels_before_tile(rp, (uint32_t)__builtin_amdgcn_readfirstlane((int)(pix >> 6)), tw, th);
    const uint32_t at = before * rp.batch_samples + js * (tw * th) + ((p >> 3) & 7u) * tw + (p & 7u);
    if (valid) {
        const PtcConfig& cf = rp.cf;
        const uint32_t index = x * rp.height + y;  // PinholeCamera.cu:23
        const uint32_t key0 = index * rp.frame_id;
        float jx = 0.5f, jy = 0.5f;  // :47-48
        if (j != 0) {
            uint32_t st = lcg_skip(pixseed[pix], 2u * (j - 1u));
            jx = rnd(st);
            jy = rnd(st);
        }
        const float px = (float)x + jx, py = (float)y + jy;
        const float abx = cf.tanHFov.x * (px - cf.hSize.x) / cf.hSize.x;  // :49
        const float aby = cf.tanHFov.y * (py - cf.hSize.y) / cf.hSize.y;
        const vec3 dir = normalize(abx * P3v(cf.u) + aby * P3v(cf.v) - P3v(cf.w));  // :51
        const uint32_t seed = segment_seed(rp, key0, j, 0u, segctr, pix);
        ps.d_depth[at] = make_float4(dir.x, dir.y, dir.z, u2f(seed));  // origin = the eye, depth 0: load_segment
        ps.pid[at] = p;
        // the path's radiance sum: seeded with the pixel's running sum (reference order of additions), or left to
        // the first bounce, which writes it (RenderParams::init_L) -- every camera segment ends in exactly one of a
        // miss, a hit that adds its radiance directly, or a hit that goes through the gather kernel
        if (seed_L_from_acc) {
            const float* a = acc + 3 * ((size_t)y * rp.width + x);
            ps.L[p] = make_float4(a[0], a[1], a[2], 0.f);
        }
    }
}

// ------------------------------------------------------------------ extend --

// A light sample's term (the f*G value the reference adds when the shadow ray is unoccluded) lives in three consecutive
// floats at index `slot` = sample * capacity + position of the segment in the bounce's hit queue: one 12-byte store or
// load, and the lanes of a wave touch consecutive records.
struct ContribRecord {
    float x, y, z;
};
__device__ __forceinline__ void contrib_store(const PathState& ps, uint32_t slot, vec3 v) {
    *reinterpret_cast<ContribRecord*>(ps.contrib + 3 * (size_t)slot) = ContribRecord{v.x, v.y, v.z};
}
__device__ __forceinline__ vec3 contrib_load(const PathState& ps, uint32_t slot) {
    const ContribRecord r = *reinterpret_cast<const ContribRecord*>(ps.contrib + 3 * (size_t)slot);
    return mk3(r.x, r.y, r.z);
}

// Work distribution of the two traversal kernels.  Each WAVE owns a span of
// consecutive queue entries [pos, end) that it reserves from one global counter
// (`wave_chunk` entries per atomic, so a few thousand atomics per launch); lanes
// that finished their ray take the next entries of the span.  Everything here is
// wave-uniform except the returned index (0xFFFFFFFF = nothing for this lane).
struct WaveWork {
    uint32_t pos, end;
    bool drained;  // the global counter is past `count`
};

__device__ __forceinline__ uint32_t wave_fetch(WaveWork& w, bool idle, uint32_t* __restrict__ counter, uint32_t wave_chunk, uint32_t count) {
    const uint32_t lane = __lane_id();
    if (w.pos >= w.end && !w.drained) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(counter, wave_chunk);
        // a scalar broadcast (all lanes of the wave are here): the compiler then knows that the span, and with it
        // every branch that depends on it, is wave-uniform
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        w.pos = base;
        w.end = min(base + wave_chunk, count);
        if (base >= count) {
            w.drained = true;
            w.pos = w.end = 0;
        }
    }
    const unsigned long long mask = __ballot(idle);
    const uint32_t take = min((uint32_t)__popcll(mask), w.end - w.pos);
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t idx = (idle && rank < take) ? w.pos + rank : 0xFFFFFFFFu;
    w.pos += take;
    return idx;
}

// ---- small scenes: hierarchy and primitives in LDS ---------------------------------------------
// A scene of a few dozen primitives (the Cornell boxes, scene 2, most of scene 1) is a few kilobytes: every block copies
// the binary nodes, the triangle and sphere records and their primitive numbers into LDS once and traverses that copy.
// The node and primitive fetches of a step then are LDS reads (tens of cycles, no work for the vector memory
// pipeline) instead of L1 hits (hundreds of cycles through the texture addresser, which these kernels keep busy);
// the arithmetic and its results are the same.  Such a hierarchy is shallow (at most 64 nodes are allowed), so these
// kernels keep 8 stack entries per ray in LDS instead of 24 (deeper ones go to scratch as always) and the copy takes the
// 16 KB that frees: the LDS per block, and with it the six resident blocks per CU, stay as they are.  Only the
// refilling kernels do this: their blocks live for the whole launch, whereas a block of the one-ray-per-thread
// kernels would copy the scene for 256 rays (measured: scene 2 sphere.test shadow rays 8.2 -> 9.1 ms with the copy in
// k_shadow_static, 5.8 ms in k_shadow).
#define LDS_SCENE_BYTES 16384u
#define LDS_SCENE_STACK 8
#define LDS_SCENE_NODES 64u
static __host__ __device__ inline uint32_t lds_scene_bytes(const DevScene& sc) {
    return sc.n_nodes * 64u + sc.n_tris * 48u + sc.n_spheres * 128u + (((sc.n_tris + sc.n_spheres) * 4u + 15u) & ~15u);
}
// View of `sc` whose traversal arrays are the block's LDS copy (LDS = false: `sc` itself).  All threads of the
// block must call it; it ends with a barrier.
template <bool LDS>
__device__ __forceinline__ DevScene scene_in_lds(const DevScene& sc, uint4* __restrict__ lds) {
    DevScene v = sc;
    if (LDS) {
        const uint32_t o_tri = sc.n_nodes * 4u, o_sph = o_tri + sc.n_tris * 3u, o_prim = o_sph + sc.n_spheres * 8u;  // in 16-byte units
        const uint4* g_nodes = reinterpret_cast<const uint4*>(sc.nodes);
        const uint4* g_tri = reinterpret_cast<const uint4*>(sc.tri_q);
        const uint4* g_sph = reinterpret_cast<const uint4*>(sc.spheres);
        for (uint32_t k = threadIdx.x; k < o_tri; k += blockDim.x) lds[k] = g_nodes[k];
        for (uint32_t k = threadIdx.x; k < sc.n_tris * 3u; k += blockDim.x) lds[o_tri + k] = g_tri[k];
        for (uint32_t k = threadIdx.x; k < sc.n_spheres * 8u; k += blockDim.x) lds[o_sph + k] = g_sph[k];
        int32_t* prim = reinterpret_cast<int32_t*>(lds + o_prim);
        for (uint32_t k = threadIdx.x; k < sc.n_tris; k += blockDim.x) prim[k] = sc.tri_prim[k];
        for (uint32_t k = threadIdx.x; k < sc.n_spheres; k += blockDim.x) prim[sc.n_tris + k] = sc.sph_prim[k];
        v.nodes = reinterpret_cast<const BvhNode*>(lds);
        v.tri_q = reinterpret_cast<const float4*>(lds + o_tri);
        v.spheres = reinterpret_cast<const DevSphere*>(lds + o_sph);
        v.tri_prim = prim;
        v.sph_prim = prim + sc.n_tris;
        __syncthreads();
    }
    return v;
}
#define LDS_SCENE_WORDS(LDS) ((LDS) ? LDS_SCENE_BYTES / 16u : 1u)

// Closest hit of the live paths, i.e. of entries [0, count) of the path state: hit_t / hit_ref
// at the same position.
template <bool COUNT, bool LDS = false, bool SPHERES_FIRST = false>
__global__ __launch_bounds__(BLOCK) void k_extend(RenderParams rp, DevScene sc_global, PathState ps, uint32_t count,
                                                  uint32_t* __restrict__ work_counter, uint32_t wave_chunk, unsigned long long* __restrict__ trav_counters) {
    constexpr int NS = LDS ? LDS_SCENE_STACK : TRAV_LDS_STACK;
    __shared__ uint32_t stack[NS * BLOCK];
    __shared__ uint4 scene_copy[LDS_SCENE_WORDS(LDS)];
    const DevScene sc = scene_in_lds<LDS>(sc_global, scene_copy);
    uint32_t* my_stack = stack + threadIdx.x;
    uint32_t spill[TRAV_SPILL_STACK];
    TravRay r;
    ClosestHit best;
    TravCounters cnt = {0u, 0u};
    int sp = 0;
    uint32_t cur = REF_DONE, p = 0;
    bool active = false;
    WaveWork w = {0u, 0u, false};
    for (;;) {
        const uint32_t nidle = (uint32_t)__popcll(__ballot(!active));
        if (nidle >= rp.tune_refill && !(w.drained && w.pos >= w.end)) {
            const uint32_t idx = wave_fetch(w, !active, work_counter, wave_chunk, count);
            if (idx != 0xFFFFFFFFu) {
                p = idx;
                float4 o4, d4;
                load_segment(rp, ps, p, o4, d4);
                active = trav_init(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), rp.cf.epsilon, PT_RAY_MAX, r, sp, cur, best);
                if (!active) {  // NaN direction or empty scene: a miss
                    ps.hit_t[p] = best.t;
                    ps.hit_ref[p] = HIT_MISS;
                } else if (SPHERES_FIRST) {
                    // the scene's one or two spheres for every new ray, all lanes together (as a leaf of the hierarchy they are
                    // reached by some lanes of the wave at a time, and the sphere test is the longest code of the loop); the
                    // closest hit is the minimum over all accepted tests, in any order
                    for (uint32_t k = 0; k < sc.n_spheres; ++k) accept_sphere(sc, k, r.o, r.d, r.tmin, r.tmax, best);
                    if (COUNT) cnt.prims += sc.n_spheres;
                    cur = sc.spheres_first_root;
                }
            }
        }
        if (!__any(active)) {
            if (w.drained && w.pos >= w.end) break;
            continue;
        }
        if (active) {
            bool occluded = false;
            const bool fin = rp.tune_mode == 2 ? trav_step<false, COUNT, NS>(sc, r, my_stack, spill, sp, cur, best, cnt, occluded)
                                               : trav_round<false, COUNT, NS>(sc, r, my_stack, spill, sp, cur, best, cnt, occluded);
            if (fin) {
                ps.hit_t[p] = best.t;
                ps.hit_ref[p] = best.ref;
                active = false;
            }
        }
    }
    if (COUNT) {
        atomicAdd(&trav_counters[0], (unsigned long long)cnt.nodes);
        atomicAdd(&trav_counters[1], (unsigned long long)cnt.prims);
    }
}

// One ray per thread, one tight traversal loop.  On this hardware it beats the
// refilling variant above: the per-round bookkeeping of the latter costs more issue
// slots than the idle lanes it recovers.
template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_extend_static(RenderParams rp, DevScene sc, PathState ps, uint32_t count,
                                                         unsigned long long* __restrict__ trav_counters) {
    __shared__ uint32_t stack[TRAV_LDS_STACK * BLOCK];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t p = i;
    float4 o4, d4;
    load_segment(rp, ps, p, o4, d4);
    ClosestHit best;
    TravCounters cnt = {0u, 0u};
    traverse_simple<false, COUNT>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), rp.cf.epsilon, PT_RAY_MAX, stack + threadIdx.x, best, cnt);
    ps.hit_t[p] = best.t;
    ps.hit_ref[p] = best.ref;
    if (COUNT) {
        atomicAdd(&trav_counters[0], (unsigned long long)cnt.nodes);
        atomicAdd(&trav_counters[1], (unsigned long long)cnt.prims);
        atomicMax(&trav_counters[5], (unsigned long long)cnt.nodes);
        atomicMax(&trav_counters[6], (unsigned long long)cnt.prims);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_shadow_static(RenderParams rp, DevScene sc, PathState ps, const uint32_t* __restrict__ count_ptr,
                                                         unsigned long long* __restrict__ trav_counters, unsigned long long* __restrict__ shadow_total) {
    __shared__ uint32_t stack[TRAV_LDS_STACK * BLOCK];
    const uint32_t count = *count_ptr;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) atomicAdd(shadow_total, (unsigned long long)count);
    if (i >= count) return;
    float4 o4, d4;
    load_shadow_ray(ps, i, o4, d4);
    ClosestHit tmp;
    TravCounters cnt = {0u, 0u};
    if (traverse_simple<true, COUNT>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), rp.cf.epsilon, o4.w, stack + threadIdx.x, tmp, cnt)) {
        const uint32_t slot = f2u(d4.w);
        contrib_store(ps, slot, mk3(0.f));
    }
    if (COUNT) {
        atomicAdd(&trav_counters[2], (unsigned long long)cnt.nodes);
        atomicAdd(&trav_counters[3], (unsigned long long)cnt.prims);
    }
}

// ---- traversal with deferred primitive tests --------------------------------
// In the kernels above a lane that reaches a leaf tests its primitives while the
// other lanes of the wave wait (and the other way round), so on scattered rays only
// a quarter of the lanes does useful work in an average instruction.  Here a lane
// never tests a primitive of its own ray: reaching a leaf it appends one item
// (ray lane, primitive) per primitive to a queue of its WAVE in LDS, pops its stack
// and goes on with the next node.  Whenever the queue holds 64 items the whole wave
// tests 64 primitives, one per lane, against rays it reads from LDS.  Both phases
// run with (nearly) all lanes busy.  The price: a ray's closest distance may lag a
// step or two behind the leaves it has passed, which costs a few extra node visits,
// never a wrong result (only exact primitive tests accept, and
// the closest hit is the minimum of (t, primitive index) over all accepted tests, in
// any order -- an LDS 64-bit atomic min).
//   ANY = false: extension rays (closest hit -> hit_t / hit_ref)
//   ANY = true : shadow rays (first accepted primitive zeroes the ray's term)
#ifndef DQ_STACK
#define DQ_STACK 12                         /* LDS stack entries per ray; deeper ones in scratch */
#endif
#define DQ_SPILL (TRAV_LDS_STACK + TRAV_SPILL_STACK - DQ_STACK)
#ifndef DQ_LEAF
#define DQ_LEAF 3u                          /* primitives a lane queues per step (3 = BVH_LEAF_MAX: two ballots give the queue positions; 4 needs three: -2 %) */
#endif
#define DQ_ITEMS (64 + 64 * DQ_LEAF)        /* < 64 left over + DQ_LEAF per lane */
#define DQ_ITEM(lane, sphere, slot) (((lane) << 26) | ((sphere) ? 0x2000000u : 0u) | (slot))

struct DeferredLds {
    uint32_t stack[DQ_STACK * BLOCK];
    float2 ray[3][BLOCK];                   // (o.x,o.y) (o.z,d.x) (d.y,d.z)
    unsigned long long best[BLOCK];         // (bits of closest t) << 32 | global primitive index; ANY: bits of tmax << 32
    int best_ref[BLOCK];                    // HIT_* of the closest hit; ANY: anything but HIT_MISS = occluded
    uint32_t items[BLOCK / 64][DQ_ITEMS];
};

// Make the LDS writes of this wave's lanes visible to its other lanes.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One queued primitive test.  `rt` = block-relative index of the ray's owner.
template <bool ANY, bool SPH>
__device__ __forceinline__ void deferred_item(const DevScene& sc, DeferredLds& L, uint32_t item, uint32_t wave_base, float tmin) {
    const uint32_t rt = wave_base + (item >> 26), slot = item & 0x1FFFFFFu;
    const bool sphere = SPH && (item & 0x2000000u) != 0u;
    const float2 r0 = L.ray[0][rt], r1 = L.ray[1][rt], r2 = L.ray[2][rt];
    const vec3 o = mk3(r0.x, r0.y, r1.x), d = mk3(r1.y, r2.x, r2.y);
    if (ANY) {
        if (L.best_ref[rt] != HIT_MISS) return;  // already occluded
        const float tmax = u2f((uint32_t)(L.best[rt] >> 32));
        const bool hit = sphere ? any_sphere(sc, slot, o, d, tmin, tmax) : any_tri(sc, slot, o, d, tmin, tmax);
        if (hit) L.best_ref[rt] = 0;
        return;
    }
    const float tmax = PT_RAY_MAX;
    const float t_best = u2f((uint32_t)(L.best[rt] >> 32));
    Cand c;
    // the primitive's global index is fetched together with the primitive, not after the test: an accepted
    // candidate would otherwise cost the whole wave a second memory round trip per batch
    uint32_t prim = (uint32_t)(sphere ? sc.sph_prim[slot] : sc.tri_prim[slot]);
    if (sphere) {
        c = sphere_candidate(sc.spheres[slot], o, d, nullptr);
    } else {
        const float4* q = sc.tri_q + 3 * (size_t)slot;
        c = tri_candidate(q[0], q[1], q[2], o, d, tmin, t_best);
    }
    asm volatile("" : "+v"(prim));  // keeps the load above the branch
    if (!c.hit || !(c.t > tmin && c.t < tmax) || c.t > t_best) return;
    const unsigned long long key = ((unsigned long long)f2u(c.t) << 32) | prim;
    atomicMin(&L.best[rt], key);
    if (L.best[rt] == key) L.best_ref[rt] = sphere ? HIT_SPHERE(slot) : (int)slot;
}

#ifndef BIG_FIRST_OFF
#define BIG_FIRST_OFF 0
#endif
#ifndef DQ_MIN_WAVES
#define DQ_MIN_WAVES 6   /* waves per SIMD the register allocation aims at (LDS allows 6 blocks per CU) */
#endif
// SPH = false: a scene without spheres -- the sphere test and its branches are compiled out (extension traversal -1.8 %, shadow -1 % on the dragon)
template <bool ANY, bool COUNT, bool WIDE, bool SPH = true>
__global__ __launch_bounds__(BLOCK, DQ_MIN_WAVES) void k_trace_deferred(RenderParams rp, DevScene sc, PathState ps,
                                                          const uint32_t* __restrict__ count_ptr, uint32_t count_arg,
                                                          uint32_t* __restrict__ work_counter, uint32_t wave_chunk,
                                                          unsigned long long* __restrict__ trav_counters, unsigned long long* __restrict__ shadow_total) {
    __shared__ DeferredLds L;
    const uint32_t count = ANY ? *count_ptr : count_arg;
    if (ANY && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(shadow_total, (unsigned long long)count);
    const uint32_t tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6, wave_base = wave << 6;
    uint32_t* my_stack = L.stack + tid;
    uint32_t* items = L.items[wave];
    uint32_t spill[DQ_SPILL];
    TravRay r;
    ClosestHit unused;
    TravCounters cnt = {0u, 0u};
    int sp = 0;
    uint32_t cur = REF_DONE, p = 0, qn = 0;  // qn: items queued (wave-uniform)
    bool active = false;                     // this lane owns a ray
    WaveWork w = {0u, 0u, false};
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // COUNT only, wave-uniform: loop iterations, lanes that own a ray / visit a node / are at a leaf in them,
    // full and partial batches of primitive tests, items in the partial ones
    uint32_t st_iters = 0, st_owned = 0, st_node = 0, st_leaf = 0, st_full = 0, st_part = 0, st_part_items = 0;
    // COUNT only: how coherent the node visits of the wave are (see ptc_debug_counters)
    uint32_t sh_distinct = 0, sh_one_iters = 0, sh_one_lanes = 0, sh_four_iters = 0, sh_four_lanes = 0, sh_top_lanes = 0;
    for (;;) {
        // ---- retire finished rays, fetch new ones
        const bool fin = active && cur == REF_DONE;
        const uint32_t nfree = (uint32_t)__popcll(__ballot(!active || fin));
        const bool more = !(w.drained && w.pos >= w.end);
        if (nfree == 64u || (more && nfree >= rp.tune_refill)) {
            // every queued test belongs to a ray that is still owned: run them all first
            wave_lds_sync();
            while (qn > 0u) {
                const uint32_t n = qn < 64u ? qn : 64u;
                if (COUNT) {
                    if (n == 64u) st_full++;
                    else {
                        st_part++;
                        st_part_items += n;
                    }
                }
                if (lane < n) {
                    if (COUNT) cnt.prims++;
                    deferred_item<ANY, SPH>(sc, L, items[qn - n + lane], wave_base, rp.cf.epsilon);
                }
                qn -= n;
                wave_lds_sync();
            }
            if (ANY && COUNT) {  // occluded rays among the retired ones
                const unsigned long long occ = __ballot(fin && L.best_ref[tid] != HIT_MISS);
                if (lane == 0 && occ) atomicAdd(&trav_counters[7], (unsigned long long)__popcll(occ));
            }
            if (fin) {
                if (ANY) {
                    if (L.best_ref[tid] != HIT_MISS) {
                        contrib_store(ps, p, mk3(0.f));
                    }
                } else {
                    ps.hit_t[p] = u2f((uint32_t)(L.best[tid] >> 32));
                    ps.hit_ref[p] = L.best_ref[tid];
                }
                active = false;
            }
            if (more) {
                const uint32_t idx = wave_fetch(w, !active, work_counter, wave_chunk, count);
                if (idx != 0xFFFFFFFFu) {
                    float4 o4, d4;
                    bool skip_big;
                    if (ANY) {
                        load_shadow_ray(ps, idx, o4, d4, &skip_big);
                        p = f2u(d4.w);   // slot of the ray's term
                        d4.w = o4.w;     // tmax
                    } else {
                        p = idx;
                        load_segment(rp, ps, p, o4, d4, &skip_big);
                        d4.w = PT_RAY_MAX;
                    }
                    o4.w = rp.cf.epsilon;
                    active = trav_init(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), o4.w, d4.w, r, sp, cur, unused);
                    // The scene's large triangles (a floor under a mesh) sit in a leaf of their own under the root, whose box every
                    // ray that starts on the floor is inside of.  A ray that k_surface has marked -- the first half of the triangle
                    // test fails for all of them: it leaves the floor, or points away from it -- starts at the root without that leaf.
                    const uint32_t big_root = WIDE ? sc.big_root_wide : sc.big_root_bin;
                    if (!BIG_FIRST_OFF && skip_big && big_root != REF_EMPTY_WORD) cur = big_root;
                    if (active) {
                        L.ray[0][tid] = make_float2(o4.x, o4.y);
                        L.ray[1][tid] = make_float2(o4.z, d4.x);
                        L.ray[2][tid] = make_float2(d4.y, d4.z);
                        L.best[tid] = ((unsigned long long)f2u(d4.w) << 32) | 0xFFFFFFFFull;
                        L.best_ref[tid] = HIT_MISS;
                    } else if (!ANY) {  // NaN direction or empty scene: a miss
                        ps.hit_t[p] = d4.w;
                        ps.hit_ref[p] = HIT_MISS;
                    }
                }
            }
            if (!__any(active)) {
                if (w.drained && w.pos >= w.end) break;
                continue;
            }
        }
        // ---- this lane's node: start the fetch, ...
        bool do_node = active && REF_KIND(cur) == REF_INTERNAL;
        if (ANY && do_node && L.best_ref[tid] != HIT_MISS) {  // occluded meanwhile
            cur = REF_DONE;
            do_node = false;
        }
        NodeData nd;
        WideData wd;
        if (do_node) {
            if (WIDE) wd = wide_fetch(sc, cur);
            else nd = node_fetch(sc, cur);
        }
        // ---- ... run the full batches of queued tests while it is in flight, ...
        if (qn >= 64u) {
            wave_lds_sync();
            do {
                if (COUNT) {
                    cnt.prims++;
                    st_full++;
                }
                deferred_item<ANY, SPH>(sc, L, items[qn - 64u + lane], wave_base, rp.cf.epsilon);
                qn -= 64u;
            } while (qn >= 64u);
            wave_lds_sync();
        }
        if (COUNT) {
            st_iters++;
            st_owned += (uint32_t)__popcll(__ballot(active));
            st_node += (uint32_t)__popcll(__ballot(do_node));
            const unsigned long long visiting = __ballot(do_node);
            unsigned long long left = visiting;
            uint32_t distinct = 0, top = 0;
            while (left) {  // one round per distinct node among the visiting lanes
                const int l = __ffsll((long long)left) - 1;
                const uint32_t v = (uint32_t)__shfl((int)cur, l);
                const unsigned long long same = __ballot(do_node && cur == v);
                ++distinct;
                top = max(top, (uint32_t)__popcll(same));
                left &= ~same;
            }
            if (visiting) {
                const uint32_t lanes = (uint32_t)__popcll(visiting);
                sh_distinct += distinct;
                sh_top_lanes += top;
                if (distinct == 1u) {
                    sh_one_iters++;
                    sh_one_lanes += lanes;
                }
                if (distinct <= 4u) {
                    sh_four_iters++;
                    sh_four_lanes += lanes;
                }
            }
        }
        // ---- ... then descend (the closest distance includes the tests just made)
        if (do_node) {
            if (COUNT) cnt.nodes++;
            const float lim = ANY ? r.tmax : u2f((uint32_t)(L.best[tid] >> 32));
            if (WIDE) {
                cur = wide_eval<!ANY, DQ_STACK>(wd, r, lim, my_stack, spill, sp);
            } else {
                cur = node_eval<DQ_STACK>(nd, r, lim, my_stack, spill, sp);
            }
        }
        // ---- leaves: queue their primitives, carry on with the stack
        const uint32_t kind = REF_KIND(cur);
        const bool leaf = active && cur != REF_DONE && (kind == REF_TRI || (SPH && kind == REF_SPHERE));
        if (COUNT) st_leaf += (uint32_t)__popcll(__ballot(leaf));
        if (__any(leaf)) {
            const uint32_t have = leaf ? REF_COUNT(cur) : 0u;
            const uint32_t n = have < DQ_LEAF ? have : DQ_LEAF;  // 0..DQ_LEAF
            uint32_t pre = 0u, tot = 0u;
#pragma unroll
            for (uint32_t b = 0; b < (DQ_LEAF > 3u ? 3u : 2u); ++b) {
                const unsigned long long m = __ballot(((n >> b) & 1u) != 0u);
                pre += (uint32_t)__popcll(m & lt_mask) << b;
                tot += (uint32_t)__popcll(m) << b;
            }
            if (leaf) {
                const uint32_t first = REF_FIRST(cur);
                // predicated stores at fixed offsets (a loop over k becomes a loop with exec-mask bookkeeping per round)
                uint32_t* dst = items + qn + pre;
                const uint32_t item0 = DQ_ITEM(lane, SPH && kind == REF_SPHERE, first);
                dst[0] = item0;  // a leaf holds at least one primitive
#pragma unroll
                for (uint32_t k = 1; k < DQ_LEAF; ++k)
                    if (k < n) dst[k] = item0 + k;
                // a longer leaf keeps its remainder for the next step
                cur = have > n ? MAKE_LEAF(kind, first + n, have - n) : trav_pop<DQ_STACK>(my_stack, spill, sp);
            }
            qn += tot;
        }
    }
    if (COUNT) {
        atomicAdd(&trav_counters[ANY ? 2 : 0], (unsigned long long)cnt.nodes);
        atomicAdd(&trav_counters[ANY ? 3 : 1], (unsigned long long)cnt.prims);
        if (lane == 0) {
            unsigned long long* st = trav_counters + (ANY ? 16 : 8);
            atomicAdd(&st[0], (unsigned long long)st_iters);
            atomicAdd(&st[1], (unsigned long long)st_owned);
            atomicAdd(&st[2], (unsigned long long)st_node);
            atomicAdd(&st[3], (unsigned long long)st_leaf);
            atomicAdd(&st[4], (unsigned long long)st_full);
            atomicAdd(&st[5], (unsigned long long)st_part);
            atomicAdd(&st[6], (unsigned long long)st_part_items);
            unsigned long long* sh = trav_counters + (ANY ? 32 : 24);
            atomicAdd(&sh[0], (unsigned long long)sh_distinct);
            atomicAdd(&sh[1], (unsigned long long)sh_one_iters);
            atomicAdd(&sh[2], (unsigned long long)sh_one_lanes);
            atomicAdd(&sh[3], (unsigned long long)sh_four_iters);
            atomicAdd(&sh[4], (unsigned long long)sh_four_lanes);
            atomicAdd(&sh[5], (unsigned long long)sh_top_lanes);
        }
    }
}

// Brute force over all primitives (parity aid).
__global__ __launch_bounds__(BLOCK) void k_extend_brute(RenderParams rp, DevScene sc, PathState ps, uint32_t count) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t p = i;
    float4 o4, d4;
    load_segment(rp, ps, p, o4, d4);
    ClosestHit best;
    brute_force<false>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), rp.cf.epsilon, PT_RAY_MAX, best);
    ps.hit_t[p] = best.t;
    ps.hit_ref[p] = best.ref;
}

// The positions of the paths that hit something go to `hit_queue` (the only ones the
// surface kernel looks at, so that its waves are full); a miss ends the path right here: radiance = background
// colour, not scaled by the throughput (Common.cu:13-18).
#define HIT_CLASSES 4   /* (triangle | sphere) x (Phong | GGX) */

// GROUP: within the block's span of the queue the hits are ordered by shading class, so
// that (all but a few of) the surface kernel's waves see one primitive kind and one BRDF.
// TABLE (small scenes, at most CLASS_TABLE_PRIMS primitives): the block first writes the class of every primitive
// into LDS, one byte each (triangle slots, then sphere slots), and a hit looks its class up there instead of
// following two dependent loads (material id of the primitive, BRDF of the material) through the vector memory
// pipeline.
#define CLASS_TABLE_PRIMS 2048u
template <bool GROUP, bool TABLE = false>
__global__ __launch_bounds__(GEN_BLOCK) void k_compact_hits(RenderParams rp, DevScene sc, PathState ps, uint32_t count, uint32_t* __restrict__ hit_queue,
                                                        uint32_t* __restrict__ hit_count) {
    constexpr int NS = GROUP ? GEN_CHUNKS * HIT_CLASSES : GEN_CHUNKS;
    bool keep[NS];
    uint32_t at[NS], idx[GEN_CHUNKS], stream[GEN_CHUNKS];
    const bool bg = rp.bg[0] != 0.f || rp.bg[1] != 0.f || rp.bg[2] != 0.f;
    __shared__ unsigned char cls_table[TABLE ? CLASS_TABLE_PRIMS : 1u];
    if (TABLE) {
        for (uint32_t k = threadIdx.x; k < sc.n_tris + sc.n_spheres; k += GEN_BLOCK) {
            const bool tri = k < sc.n_tris;
            const uint32_t mat = tri ? sc.tri_mat[k] : sc.sph_mat[k - sc.n_tris];
            cls_table[k] = (unsigned char)((tri ? 0u : 2u) | (sc.materials[mat & ~MAT_LIGHT_BIT].brdf_type == PTC_GGX ? 1u : 0u));
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) keep[k] = false;
#pragma unroll
    for (int c = 0; c < GEN_CHUNKS; ++c) {
        const uint32_t i = (blockIdx.x * GEN_CHUNKS + c) * GEN_BLOCK + threadIdx.x;
        idx[c] = i;
        stream[c] = NS;  // none
        if (i < count) {
            const int ref = ps.hit_ref[i];
            if (ref != HIT_MISS) {
                uint32_t cls = 0;
                if (GROUP && TABLE) {
                    cls = cls_table[ref >= 0 ? (uint32_t)ref : sc.n_tris + (uint32_t)HIT_SPHERE_SLOT(ref)];
                } else if (GROUP) {
                    const uint32_t mat = ref >= 0 ? sc.tri_mat[ref] : sc.sph_mat[HIT_SPHERE_SLOT(ref)];
                    cls = (ref >= 0 ? 0u : 2u) | (sc.materials[mat & ~MAT_LIGHT_BIT].brdf_type == PTC_GGX ? 1u : 0u);
                }
                stream[c] = cls * GEN_CHUNKS + c;
            } else if (bg || rp.init_L) {
                const uint32_t p = rp.camera_rays ? ps.pid[i] : f2u(ps.thr[i].w);
                float4 L = rp.init_L ? make_float4(0.f, 0.f, 0.f, 0.f) : ps.L[p];
                L.x += rp.bg[0];
                L.y += rp.bg[1];
                L.z += rp.bg[2];
                ps.L[p] = L;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
#pragma unroll
        for (int c = 0; c < GEN_CHUNKS; ++c) keep[k] = keep[k] || stream[c] == (uint32_t)k;
    }
    queue_reserve_streams<NS>(keep, hit_count, at);
#pragma unroll
    for (int c = 0; c < GEN_CHUNKS; ++c) {
        if (stream[c] == NS) continue;
        uint32_t pos = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (stream[c] == (uint32_t)k) pos = at[k];
        hit_queue[pos] = idx[c];
    }
}

// ----------------------------------------------------------- light samples --
// Material access helpers.
__device__ __forceinline__ const PtcMaterialValue& material_of(const DevScene& sc, uint32_t mat) { return sc.materials[mat & ~MAT_LIGHT_BIT]; }
__device__ __forceinline__ bool is_light(uint32_t mat) { return (mat & MAT_LIGHT_BIT) != 0u; }

// Position of light sample q's term for the segment at position `seg` of the bounce's hit
// queue: samples_per_path * capacity records of three floats (contrib_store / contrib_load).  (Everything that lives for
// one segment only -- these terms, the gather records -- is indexed by the hit-queue
// position, which is dense, and not by the path number, whose live entries thin out with
// every bounce.)
__device__ __forceinline__ uint32_t contrib_slot(const PathState& ps, uint32_t seg, uint32_t q) { return q * ps.capacity + seg; }

// Does the integrator sample the lights at this hit?
template <int INTEG>
__device__ __forceinline__ bool samples_lights(const RenderParams& rp, uint32_t mat, uint32_t depth) {
    if (INTEG == INT_ANALYTICDIRECT) return false;
    if (INTEG == INT_DIRECT) return !is_light(mat);                                             // RayTracer.cu:160-165
    if (INTEG == INT_PATHTRACER) return !(rp.cf.next_event_est && is_light(mat) && depth != 0u);  // :528-533
    if (INTEG == INT_PATHTRACER2) return !(depth > 0u && is_light(mat));                         // :436-441
    return true;
}

// Light sample q of a segment: the shadow ray and the term the reference adds when that
// ray is unoccluded (f_brdf*G for the quad-light integrators, the Blinn-Phong term for
// `raytracer`).
template <int INTEG, bool FAST = false>
__device__ __forceinline__ vec3 light_sample(const RenderParams& rp, const DevScene& sc, const HitAttr& at, const PtcMaterialValue& mv, uint32_t seed, uint32_t q,
                                             vec3& sh_o, vec3& sh_d, float& sh_tmax) {
    typedef ShMath<FAST> M;  // FAST only ever with INT_PATHTRACER
    const PtcConfig& cf = rp.cf;
    vec3 term = mk3(0.f);
    if (INTEG == INT_RAYTRACER) {
        // RayTracer.cu:42-86: point lights, then directional lights
        if (q < sc.n_plights) {
            const vec3 loc = P3v(sc.plights[q].location);
            sh_d = normalize(loc - at.P);
            sh_tmax = length(loc - at.P);
        } else {
            sh_d = P3v(sc.dlights[q - sc.n_plights].direction);
            sh_tmax = PT_RAY_MAX;
        }
        sh_o = at.P + sh_d * cf.epsilon;
        const vec3 H = normalize(sh_d + at.wo);
        vec3 I = P3v(mv.diffuse) * fmaxf(dot(at.normal, sh_d), 0);
        I += P3v(mv.specular) * pow_pos(fmaxf(dot(at.normal, H), 0), mv.shininess);
        if (q < sc.n_plights) {
            const float lightDist = sh_tmax;
            const float att = dot(P3v(sc.plights[q].attenuation), mk3(1, lightDist, lightDist * lightDist));
            I *= P3v(sc.plights[q].color) / att;
        } else {
            I *= P3v(sc.dlights[q - sc.n_plights].color);
        }
        return I;
    }
    const int root = (int)rp.root_light_samples;
    const uint32_t per_light = (uint32_t)(root * root);
    const uint32_t k = q / per_light, r = q % per_light;
    const int si = (int)(r / (uint32_t)root), sj = (int)(r % (uint32_t)root);
    const PtcQuadLight& ql = sc.qlights[k];
    QuadFrame qf;
    qf.a = P3v(ql.tri1.v1);
    const vec3 b = P3v(ql.tri1.v2);
    // `c` is tri2.v3 in direct / pathTracer (:172,712) and tri1.v3 in
    // computeDirectLight (:380); the loader stores the same point in both.
    const vec3 c = INTEG == INT_PATHTRACER2 ? P3v(ql.tri1.v3) : P3v(ql.tri2.v3);
    qf.ab = b - qf.a;
    qf.ac = c - qf.a;
    // random numbers of this sample: position in the segment's LCG stream;
    // zeta_0..2 come first (:537-539), a glass hit draws once more (:614)
    const uint32_t lead = INTEG == INT_PATHTRACER ? (mv.matType == PTC_GLASS ? 4u : 3u) : 0u;
    uint32_t st = lcg_skip(seed, lead + 2u * q);
    const float u1 = rnd(st);
    const float u2 = rnd(st);
    const vec3 light_pos = quad_sample(qf, rp.light_stratify != 0u, si, sj, root, u1, u2, INTEG == INT_PATHTRACER2);
    sh_o = at.P;
    if (FAST) {  // one reciprocal square root for the direction and the distance
        const vec3 to_light = light_pos - sh_o;
        const float d2 = dot(to_light, to_light), inv_len = __builtin_amdgcn_rsqf(d2);
        sh_d = to_light * inv_len;
        sh_tmax = d2 * inv_len - cf.epsilon;
    } else {
        sh_d = normalize(light_pos - sh_o);
        sh_tmax = length(light_pos - sh_o) - cf.epsilon;
    }
    if (INTEG == INT_DIRECT) {
        // RayTracer.cu:204-221
        const vec3 wdir = normalize(light_pos - sh_o);
        const vec3 f_brdf = (P3v(mv.diffuse) / PT_PIf) +
                            (P3v(mv.specular) * ((mv.shininess + 2.0f) / (2.0f * PT_PIf)) *
                             pow_pos(fmaxf(dot(normalize(reflect(-at.wo, at.normal)), wdir), .0f), mv.shininess));
        const vec3 n_light = normalize(cross(qf.ab, qf.ac));
        const float R = length(sh_o - light_pos);
        const float G = (1.0f / powf(R, 2.0f)) * fmaxf(dot(at.normal, normalize(light_pos - sh_o)), .0f) *
                        (fmaxf(dot(n_light, normalize(light_pos - sh_o)), .0f));
        term = f_brdf * G;
    } else if (INTEG == INT_PATHTRACER) {
        // RayTracer.cu:745-776
        const vec3 n = normalize(at.normal);  // exact in either flavour: see ggx_brdf
        const vec3 rr = M::normalize(reflect(-at.wo, at.normal));
        // normalize(light_pos - sh_o) is formed four times in the source, with the same value each time
        const vec3 to_light = light_pos - sh_o;
        const vec3 w_i_dir = sh_d;
        vec3 f_brdf = mk3(.0f);
        if (mv.brdf_type == PTC_MOD_PHONG) {
            f_brdf = phong_brdf<FAST>(w_i_dir, rr, mv.shininess, P3v(mv.diffuse), P3v(mv.specular));
        } else {
            const float wi_dot_n_dir = dot(w_i_dir, n);
            const float wo_dot_n_dir = dot(at.wo, n);
            if (wi_dot_n_dir > .0f && wo_dot_n_dir > .0f) {
                const vec3 f_brdf_GGX = ggx_brdf<FAST>(w_i_dir, at.wo, n, mv.roughness, P3v(mv.specular));
                f_brdf = (P3v(mv.diffuse) / PT_PIf) + f_brdf_GGX;
            }
        }
        const vec3 n_geo = at.normal;  // the inner `n` of :767, not normalised
        const vec3 n_light = M::normalize(cross(qf.ab, qf.ac));
        // 1 / R^2 with R = length(sh_o - light_pos): the reciprocal of the squared length, without the root
        const float inv_R2 = FAST ? M::rcp(dot(to_light, to_light)) : 1.0f / powf(length(sh_o - light_pos), 2.0f);
        const float G = inv_R2 * fmaxf(dot(n_geo, w_i_dir), .0f) * (fmaxf(dot(n_light, w_i_dir), .0f));
        term = f_brdf * G;
    } else if (INTEG == INT_PATHTRACER2) {
        // RayTracer.cu:410-420
        const float dist = length(light_pos - at.P);
        const vec3 n_light = normalize(cross(qf.ab, qf.ac));
        const vec3 reflect_dir = reflect(-at.wo, at.normal);
        const float G = (1.0f / (dist * dist)) * fmaxf(dot(at.normal, sh_d), .0f) * fmaxf(dot(n_light, sh_d), .0f);
        const vec3 f_brdf = phong_brdf(sh_d, reflect_dir, mv.shininess, P3v(mv.diffuse), P3v(mv.specular));
        term = f_brdf * G;
    }
    return term;
}

// ------------------------------------------------------------------ shadow --
// Any-hit traversal of the queued shadow rays, same work distribution as k_extend.
template <bool COUNT, bool LDS = false, bool SPHERES_FIRST = false>
__global__ __launch_bounds__(BLOCK) void k_shadow(RenderParams rp, DevScene sc_global, PathState ps, const uint32_t* __restrict__ count_ptr,
                                                  uint32_t* __restrict__ work_counter, uint32_t wave_chunk, unsigned long long* __restrict__ trav_counters,
                                                  unsigned long long* __restrict__ shadow_total) {
    constexpr int NS = LDS ? LDS_SCENE_STACK : TRAV_LDS_STACK;
    __shared__ uint32_t stack[NS * BLOCK];
    __shared__ uint4 scene_copy[LDS_SCENE_WORDS(LDS)];
    const DevScene sc = scene_in_lds<LDS>(sc_global, scene_copy);
    const uint32_t count = *count_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(shadow_total, (unsigned long long)count);
    uint32_t* my_stack = stack + threadIdx.x;
    uint32_t spill[TRAV_SPILL_STACK];
    TravRay r;
    ClosestHit best;
    TravCounters cnt = {0u, 0u};
    int sp = 0;
    uint32_t cur = REF_DONE, slot = 0;
    bool active = false;
    WaveWork w = {0u, 0u, false};
    for (;;) {
        const uint32_t nidle = (uint32_t)__popcll(__ballot(!active));
        if (nidle >= rp.tune_refill && !(w.drained && w.pos >= w.end)) {
            const uint32_t idx = wave_fetch(w, !active, work_counter, wave_chunk, count);
            if (idx != 0xFFFFFFFFu) {
                float4 o4, d4;
                load_shadow_ray(ps, idx, o4, d4);
                slot = f2u(d4.w);
                active = trav_init(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), rp.cf.epsilon, o4.w, r, sp, cur, best);
                if (SPHERES_FIRST && active) {  // see k_extend
                    bool blocked = false;
                    for (uint32_t k = 0; k < sc.n_spheres; ++k) blocked = blocked || any_sphere(sc, k, r.o, r.d, r.tmin, r.tmax);
                    if (COUNT) cnt.prims += sc.n_spheres;
                    if (blocked) {
                        contrib_store(ps, slot, mk3(0.f));
                        active = false;
                    }
                    cur = sc.spheres_first_root;
                }
            }
        }
        if (!__any(active)) {
            if (w.drained && w.pos >= w.end) break;
            continue;
        }
        if (active) {
            bool occluded = false;
            const bool fin = rp.tune_shadow == 2 ? trav_step<true, COUNT, NS>(sc, r, my_stack, spill, sp, cur, best, cnt, occluded)
                                                 : trav_round<true, COUNT, NS>(sc, r, my_stack, spill, sp, cur, best, cnt, occluded);
            if (fin) {
                if (occluded) {
                    contrib_store(ps, slot, mk3(0.f));
                }
                active = false;
            }
        }
    }
    if (COUNT) {
        atomicAdd(&trav_counters[2], (unsigned long long)cnt.nodes);
        atomicAdd(&trav_counters[3], (unsigned long long)cnt.prims);
    }
}

__global__ __launch_bounds__(BLOCK) void k_shadow_brute(RenderParams rp, DevScene sc, PathState ps, const uint32_t* __restrict__ count_ptr,
                                                        unsigned long long* __restrict__ shadow_total) {
    const uint32_t count = *count_ptr;
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) atomicAdd(shadow_total, (unsigned long long)count);
    if (i >= count) return;
    float4 o4, d4;
    load_shadow_ray(ps, i, o4, d4);
    ClosestHit tmp;
    if (brute_force<true>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), rp.cf.epsilon, o4.w, tmp)) {
        const uint32_t slot = f2u(d4.w);
        contrib_store(ps, slot, mk3(0.f));
    }
}

// ----------------------------------------------------------------- surface --
// What happens at a hit is split in two kernels around the shadow rays:
//   k_surface  everything that does not need the visibility of the light samples: the
//              shadow rays + their terms, emission, the sampled direction, the throughput
//              factor, Russian roulette, the next queue;
//   k_gather   (after k_shadow) the radiance of the segment, which folds the terms of the
//              unoccluded light samples in the order the reference's loops add them.
struct Segment {
    vec3 o, d, thr;
    uint32_t depth, seed;
    bool done;
};

// gather = false: `a` is the radiance of the segment.
// gather = true : the segment sampled the lights; `a` is the integrator's constant part
//                 and k_gather computes the radiance from it, the throughput before this
//                 segment and the light terms.
struct SurfaceOut {
    vec3 a;
    bool gather;
};

__device__ __forceinline__ vec3 contrib_at(const RenderParams& rp, const PathState& ps, uint32_t seg, uint32_t q) {
    return contrib_load(ps, contrib_slot(ps, seg, q));
}

// Sum of the light-sample terms of quad light k in loop order.
__device__ __forceinline__ vec3 light_sum(const RenderParams& rp, const PathState& ps, uint32_t seg, uint32_t k, uint32_t per_light) {
    vec3 s = mk3(.0f);
    for (uint32_t r = 0; r < per_light; ++r) s += contrib_at(rp, ps, seg, k * per_light + r);
    return s;
}

__device__ __forceinline__ float quad_area(const PtcQuadLight& ql, bool cdl) {
    const vec3 a = P3v(ql.tri1.v1), b = P3v(ql.tri1.v2), c = cdl ? P3v(ql.tri1.v3) : P3v(ql.tri2.v3);
    return length(cross(b - a, c - a));
}

// RayTracer.cu:34-105
__device__ __forceinline__ SurfaceOut surface_raytracer(const RenderParams& rp, const DevScene& sc, const HitAttr& at, Segment& sg) {
    const PtcMaterialValue& mv = material_of(sc, at.mat);
    const vec3 result = P3v(mv.ambient) + P3v(mv.emission);
    const bool lights = (sc.n_plights + sc.n_dlights) != 0u;
    const vec3 a = lights ? result : result * sg.thr;
    if (length(P3v(mv.specular)) > 0) {
        sg.o = at.P;
        sg.d = reflect(-at.wo, at.normal);
        sg.depth++;
        sg.thr *= P3v(mv.specular);
    } else {
        sg.done = true;
    }
    return SurfaceOut{a, lights};
}
__device__ __forceinline__ vec3 gather_raytracer(const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t seg, vec3 a, vec3 thr) {
    vec3 result = a;
    const uint32_t nl = sc.n_plights + sc.n_dlights;
    for (uint32_t q = 0; q < nl; ++q) result += contrib_at(rp, ps, seg, q);  // occluded lights left 0
    return result * thr;
}

// RayTracer.cu:107-151
__device__ __forceinline__ SurfaceOut surface_analytic(const RenderParams& rp, const DevScene& sc, const HitAttr& at, Segment& sg) {
    const PtcMaterialValue& mv = material_of(sc, at.mat);
    vec3 result = P3v(mv.ambient) + P3v(mv.emission);
    if (rp.light_samples == 0u) {
        for (uint32_t i = 0; i < sc.n_qlights; ++i) {
            const vec3 f_brdf = P3v(mv.diffuse) / PT_PIf;
            const vec3 hitPt = at.P, hitPtNormal = at.normal;
            const PtcQuadLight& q = sc.qlights[i];
            const vec3 p1 = P3v(q.tri1.v1), p2 = P3v(q.tri1.v2), p3 = P3v(q.tri2.v2), p4 = P3v(q.tri1.v3);
            const float theta_1 = acosf(dot(normalize(p1 - hitPt), normalize(p2 - hitPt)));
            const float theta_2 = acosf(dot(normalize(p2 - hitPt), normalize(p3 - hitPt)));
            const float theta_3 = acosf(dot(normalize(p3 - hitPt), normalize(p4 - hitPt)));
            const float theta_4 = acosf(dot(normalize(p4 - hitPt), normalize(p1 - hitPt)));
            const vec3 gamma_1 = normalize(cross((p1 - hitPt), (p2 - hitPt)));
            const vec3 gamma_2 = normalize(cross((p2 - hitPt), (p3 - hitPt)));
            const vec3 gamma_3 = normalize(cross((p3 - hitPt), (p4 - hitPt)));
            const vec3 gamma_4 = normalize(cross((p4 - hitPt), (p1 - hitPt)));
            const vec3 irradiance_vec = 0.5f * ((theta_1 * gamma_1) + (theta_2 * gamma_2) + (theta_3 * gamma_3) + (theta_4 * gamma_4));
            result += f_brdf * P3v(q.color) * dot(irradiance_vec, hitPtNormal);
        }
    }
    sg.done = true;
    return SurfaceOut{result, false};
}

// RayTracer.cu:153-230
__device__ __forceinline__ SurfaceOut surface_direct(const RenderParams& rp, const DevScene& sc, const HitAttr& at, Segment& sg) {
    const PtcMaterialValue& mv = material_of(sc, at.mat);
    const vec3 result = P3v(mv.ambient) + P3v(mv.emission);
    sg.done = true;
    if (is_light(at.mat)) {
        sg.depth = rp.max_depth_eff;
        return SurfaceOut{result, false};
    }
    return SurfaceOut{result, sc.n_qlights != 0u};
}
__device__ __forceinline__ vec3 gather_direct(const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t seg, vec3 a) {
    vec3 result = a;
    const uint32_t per_light = rp.root_light_samples * rp.root_light_samples;
    for (uint32_t k = 0; k < sc.n_qlights; ++k) {
        const float area = fabsf(quad_area(sc.qlights[k], false));
        result += P3v(sc.qlights[k].color) * light_sum(rp, ps, seg, k, per_light) * (area / (float)rp.light_samples);
    }
    return result;
}

// RayTracer.cu:429-494
__device__ __forceinline__ SurfaceOut surface_pathtracer2(const RenderParams& rp, const DevScene& sc, const HitAttr& at, Segment& sg) {
    const PtcMaterialValue& mv = material_of(sc, at.mat);
    const vec3 L_e = P3v(mv.emission);
    if (sg.depth > 0u && is_light(at.mat)) {
        sg.depth = rp.max_depth_eff;
        sg.done = true;
        return SurfaceOut{mk3(.0f), false};
    }
    const uint32_t per_light = rp.root_light_samples * rp.root_light_samples;
    uint32_t st = lcg_skip(sg.seed, 2u * per_light * sc.n_qlights);
    const float zeta1 = rnd(st), zeta2 = rnd(st);
    const PolarAngle theta = angle_acos(zeta1);
    const float phi = zeta2;  // in turns: 2*pi*zeta2 (:447)
    const vec3 w = normalize(at.normal);
    vec3 a = mk3(0, 1, 0);
    if (1.0f - fabsf(dot(a, w)) <= 1.0f) a = mk3(1, 0, 0);
    const vec3 u = normalize(cross(a, w)), v = normalize(cross(w, u));
    const vec3 w_i = normalize(spherical_dir(theta, phi, u, v, w));
    const vec3 norm = normalize(at.normal);
    const vec3 incoming_ray_dir = normalize(-at.wo);
    const vec3 reflected_ray_dir = normalize(reflect(incoming_ray_dir, norm));
    const vec3 f_brdf = phong_brdf(w_i, reflected_ray_dir, mv.shininess, P3v(mv.diffuse), P3v(mv.specular));
    const float pdf = 1.0f / (2.0f * PT_PIf);
    const vec3 attenuation = (f_brdf * fmaxf(dot(norm, w_i), .0f)) * (1.0f / pdf);
    vec3 radiance = mk3(.0f);
    if (sg.depth == 0u) radiance += L_e * sg.thr;
    // "radiance += L_d * throughput" (:471) happens in gather_pathtracer2
    sg.thr *= attenuation;
    const float q = fminf(fmaxf(fmaxf(sg.thr.x, sg.thr.y), sg.thr.z), 1.0f);
    if (rnd(st) > q) {
        sg.depth = rp.max_depth_eff;
        sg.done = true;
        return SurfaceOut{radiance, true};
    }
    sg.thr *= 1.0f / (q);
    sg.o = at.P;
    sg.d = w_i;
    sg.depth++;
    return SurfaceOut{radiance, true};
}
__device__ __forceinline__ vec3 gather_pathtracer2(const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t seg, vec3 a, vec3 thr) {
    const uint32_t per_light = rp.root_light_samples * rp.root_light_samples;
    vec3 L_d = mk3(.0f);
    for (uint32_t k = 0; k < sc.n_qlights; ++k) {
        const float area = fabsf(quad_area(sc.qlights[k], true));
        L_d += P3v(sc.qlights[k].color) * (area / (float)rp.light_samples) * light_sum(rp, ps, seg, k, per_light);
    }
    vec3 radiance = a;
    radiance += L_d * thr;
    return radiance;
}

// RayTracer.cu:327-332
__device__ __forceinline__ float phase_hg(float wi_dot_wo, float g) {
    const float numerator = (1.0f - (g * g));
    const float denom = powf((1.0f + powf(g, 2.0f) + (2.0f * g * wi_dot_wo)), (3.0f / 2.0f));
    return (1.0f / (4.0f * PT_PIf)) * (numerator / denom);
}

// RayTracer.cu:337-360: the sampled direction is written to a by-value argument and
// never leaves the function, so only the phase-function value is computed here.
__device__ __forceinline__ float sample_phase_hg(float g, float zeta) {
    float cos_theta;
    if (fabsf(g) < 0.001f) cos_theta = 1.0f - (2.0f * zeta);
    else cos_theta = (1.0f / (2.0f * g)) * (1.0f + powf(g, 2.0f) - powf((1 - g * g) / (1 - g + (2.0f * g * zeta)), 2.0f));
    return phase_hg(-cos_theta, g);
}

// RayTracer.cu:496-963, all three material types, without the light sum (gather_pathtracer).
template <bool FAST = false>
__device__ __forceinline__ SurfaceOut surface_pathtracer(const RenderParams& rp, const DevScene& sc, const HitAttr& at, Segment& sg) {
    typedef ShMath<FAST> M;
    const PtcMaterialValue& mv = material_of(sc, at.mat);
    const PtcConfig& cf = rp.cf;
    const vec3 n = normalize(at.normal);  // exact in either flavour: see ggx_brdf
    const vec3 L_e = P3v(mv.emission), K_s = P3v(mv.specular), K_d = P3v(mv.diffuse);
    const vec3 r = M::normalize(reflect(-at.wo, at.normal));
    PolarAngle theta = angle_zero();
    float phi = .0f;  // in turns
    vec3 w_i = mk3(.0f);
    vec3 addon_throughput = mk3(1.0f);
    vec3 new_pos = mk3(.0f);  // :517

    if (cf.next_event_est && is_light(at.mat) && sg.depth != 0u) {  // :528-533
        sg.depth = rp.max_depth_eff;
        sg.done = true;
        return SurfaceOut{mk3(.0f), false};
    }
    uint32_t st = sg.seed;
    const float zeta_0 = rnd(st), zeta_1 = rnd(st), zeta_2 = rnd(st);
    const float K_s_avg = FAST ? (K_s.x + K_s.y + K_s.z) * (1.0f / 3.0f) : (K_s.x + K_s.y + K_s.z) / 3.0f;
    const float K_d_avg = FAST ? (K_d.x + K_d.y + K_d.z) * (1.0f / 3.0f) : (K_d.x + K_d.y + K_d.z) / 3.0f;
    float t = .0f;
    if ((K_s_avg == .0f) && (K_d_avg == .0f)) {  // :546-553
        if (mv.brdf_type == PTC_MOD_PHONG) t = .0f;
        else if (mv.brdf_type == PTC_GGX) t = 1.0f;
    } else {
        const float k_val = M::div(K_s_avg, K_s_avg + K_d_avg);
        t = mv.brdf_type == PTC_MOD_PHONG ? k_val : fmaxf(.25f, k_val);
    }
    compute_polar_angles<FAST>(rp.semantics, rp.sampling_method, phi, theta, zeta_0, zeta_1, zeta_2, t, mv.shininess);  // :566

    if (mv.matType == PTC_GLASS) {  // :574-621
        const vec3 inc_ray_dir = -at.wo;
        const vec3 norm = n;
        float n1 = 1.0f, n2 = mv.ior;
        float i_dot_n;
        vec3 chosen_ray_dir = mk3(.0f);
        if (dot(inc_ray_dir, norm) >= .0f) {
            if (rp.semantics == 1) {  // swapValue takes its arguments by value (:296-300)
                const float tmp = n1;
                n1 = n2;
                n2 = tmp;
            }
            i_dot_n = dot(inc_ray_dir, norm);
        } else {
            i_dot_n = dot(inc_ray_dir, -norm);
        }
        const float eta = n2 / n1;
        float reflect_probability = 1.0f;
        if (refract(chosen_ray_dir, inc_ray_dir, norm, eta)) reflect_probability = fresnel_schlick(i_dot_n);
        const float rand_valz = rnd(st);  // :614, one more draw before the light samples
        if (rand_valz < reflect_probability) w_i = normalize(reflect(inc_ray_dir, norm));
        else w_i = normalize(chosen_ray_dir);
    } else if (mv.matType == PTC_VOLUMETRIC) {  // :622-672
        // payload.dir there is this segment's ray direction (PinholeCamera.cu:65; for a
        // camera ray, where the reference leaves it unset, it is defined the same way)
        const vec3 inc_ray_dir = -at.wo;
        w_i = sg.d;
        if (!(dot(inc_ray_dir, n) >= .0f)) {
            const vec3 surface_pt = at.P, curr_pt = sg.o;
            const float ray_distance = length(surface_pt - curr_pt);
            const float param_dist_t = 0.08f;
            const float transmit = expf(-mv.sigma_t * fminf(param_dist_t, PT_RAY_MAX) * ray_distance);  // :362-365
            const float weight = sample_phase_hg(mv.g, zeta_0);
            addon_throughput *= transmit * weight;  // every case of the switch below overwrites it
            new_pos = param_dist_t < ray_distance ? curr_pt + (w_i * param_dist_t) : surface_pt;
        }
    } else if (mv.brdf_type == PTC_MOD_PHONG) {  // :675-684
        const vec3 w = ((zeta_0 <= t) && (rp.sampling_method == PTC_BRDF_SAMPLING)) ? r : n;
        vec3 u = mk3(.0f), v = mk3(.0f);
        gen_coord_frame<FAST>(rp.semantics, w, u, v);
        w_i = spherical_dir<FAST>(theta, phi, u, v, w);
    } else {  // :685-696
        const PolarAngle theta_h_sample = angle_atan<FAST>(M::div(mv.roughness * M::sqrt(zeta_1), M::sqrt(1.0f - zeta_1)), rp.semantics != 1);
        const float phi_h_sample = zeta_2;  // in turns: 2*pi*zeta_2 (:687)
        vec3 u_h = mk3(.0f), v_h = mk3(.0f);
        gen_coord_frame<FAST>(rp.semantics, n, u_h, v_h);
        w_i = M::normalize(reflect(-at.wo, spherical_dir<FAST>(theta_h_sample, phi_h_sample, u_h, v_h, n)));
    }

    if (cf.next_event_est) st = lcg_skip(st, 2u * rp.root_light_samples * rp.root_light_samples * sc.n_qlights);  // the u1,u2 of every light sample (:724-725)

    vec3 f_brdf = mk3(1.0f);
    float pdf = 1.0f;
    switch (rp.sampling_method) {  // :811-897
        case PTC_HEMISPHERE_SAMPLING:
            f_brdf = phong_brdf<FAST>(w_i, r, mv.shininess, P3v(mv.diffuse), P3v(mv.specular));  // the same expression, inline in the source (:813)
            pdf = 1.0f / (2.0f * PT_PIf);
            addon_throughput = (f_brdf * fmaxf(dot(n, w_i), .0f) * (1.0f / pdf));
            break;
        case PTC_COSINE_SAMPLING:
            f_brdf = phong_brdf<FAST>(w_i, r, mv.shininess, P3v(mv.diffuse), P3v(mv.specular));  // (:822)
            pdf = FAST ? fmaxf(dot(n, w_i), .0f) * (1.0f / PT_PIf) : fmaxf(dot(n, w_i), .0f) / (PT_PIf);
            addon_throughput = (f_brdf)*fmaxf(dot(n, w_i), .0f) * M::rcp(pdf);
            break;
        case PTC_BRDF_SAMPLING:
            // The probe shadow rays of :831-860 only feed locals that are never read
            // again; they cannot change the payload and are not traced.
            if (mv.brdf_type == PTC_MOD_PHONG) {
                f_brdf = phong_brdf<FAST>(w_i, r, mv.shininess, K_d, K_s);
                pdf = FAST ? ((1.0f - t) * (dot(n, w_i) * (1.0f / PT_PIf))) + t * ((mv.shininess + 1.0f) * (1.0f / (2.0f * PT_PIf))) * pow_pos(fmaxf(dot(r, w_i), .0f), mv.shininess)
                           : ((1.0f - t) * (dot(n, w_i) / PT_PIf)) + t * ((mv.shininess + 1.0f) / (2.0f * PT_PIf)) * pow_pos(fmaxf(dot(r, w_i), .0f), mv.shininess);
            } else {
                const float wi_dot_n = dot(w_i, n);
                const float wo_dot_n = dot(at.wo, n);
                if (wi_dot_n > .0f && wo_dot_n > .0f) {
                    const vec3 h = normalize(w_i + at.wo);  // exact, as inside ggx_brdf
                    f_brdf = (K_d / PT_PIf) + ggx_brdf<FAST>(w_i, at.wo, n, mv.roughness, K_s);
                    pdf = ggx_pdf<FAST>(w_i, n, h, t, mv.roughness);
                } else {
                    f_brdf = mk3(.0f);
                }
            }
            addon_throughput = (f_brdf * fmaxf(dot(n, w_i), .0f) * M::rcp(pdf));
            break;
        default:
            break;
    }

    // :900-917: result = L_e + L_d at the first hit with NEE, L_d alone later, L_e without
    // NEE; L_d is added by gather_pathtracer
    SurfaceOut out;
    out.gather = cf.next_event_est != 0u;
    const bool with_Le = !cf.next_event_est || sg.depth == 0u;
    out.a = out.gather ? (with_Le ? L_e : mk3(.0f)) : L_e * sg.thr;

    if (cf.russian_roul) {  // :923-933, q from the throughput BEFORE this bounce's factor
        const float q = 1.0f - fminf(fmaxf(fmaxf(sg.thr.x, sg.thr.y), sg.thr.z), 1.0f);
        if (rnd(st) < q) addon_throughput *= mk3(.0f);
        else addon_throughput *= M::rcp(1.0f - q);
    }
    if (mv.matType != PTC_GLASS) sg.thr *= addon_throughput;  // :935-941 (glass: times 1)
    // :943-957; glass never assigns new_pos, so its path continues from (0,0,0)
    sg.o = (mv.matType == PTC_GLASS || mv.matType == PTC_VOLUMETRIC) ? new_pos : at.P;
    sg.d = w_i;
    sg.depth++;  // :959-962 (the condition there is a tautology)
    return out;
}

__device__ __forceinline__ vec3 gather_pathtracer(const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t seg, vec3 a, vec3 thr, bool with_Le) {
    const PtcConfig& cf = rp.cf;
    const uint32_t per_light = rp.root_light_samples * rp.root_light_samples;
    vec3 L_d = mk3(.0f);
    for (uint32_t k = 0; k < sc.n_qlights; ++k) {  // :702-800
        const vec3 sampled_result = light_sum(rp, ps, seg, k, per_light);
        const float area = quad_area(sc.qlights[k], false);
        if (cf.next_event_est == PTC_NEE_MIS) {
            // :782-794 as written: pdf_lights_k, D and h are never assigned, so
            // pdf_NEE = 0 and pdf_brdf carries a 0/0 term.
            const float exp_beta = 2.0f, pdf_NEE = .0f, pdf_brdf = __builtin_nanf("");
            const float pdf_denom_sum = powf(pdf_NEE, exp_beta) + powf(pdf_brdf, exp_beta);
            const float pdf_numerator = powf(pdf_NEE, exp_beta);
            const float weight_i = pdf_numerator / pdf_denom_sum;
            L_d += P3v(sc.qlights[k].color) * sampled_result * (area / (float)rp.light_samples) * (1.0f / pdf_NEE) * weight_i;
        } else {
            L_d += P3v(sc.qlights[k].color) * sampled_result * (area / (float)rp.light_samples);  // :798
        }
    }
    vec3 result = mk3(.0f);
    if (with_Le) result += a + L_d;  // :902
    else result += L_d;              // :908
    return result * thr;             // :918
}

template <int INTEG>
__device__ __forceinline__ vec3 gather_radiance(const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t seg, vec3 a, vec3 thr, bool with_Le) {
    if (INTEG == INT_RAYTRACER) return gather_raytracer(rp, sc, ps, seg, a, thr);
    if (INTEG == INT_DIRECT) return gather_direct(rp, sc, ps, seg, a);
    if (INTEG == INT_PATHTRACER2) return gather_pathtracer2(rp, sc, ps, seg, a, thr);
    if (INTEG == INT_PATHTRACER) return gather_pathtracer(rp, sc, ps, seg, a, thr, with_Le);
    return a;
}

#define GATHER_WITH_LE 0x80000000u   /* bits of a segment's word in PathState::g_p beside the path number */
#define GATHER_A_SET 0x40000000u
// Threads per block of k_surface.  With one or two light samples per hit: 256 for small launches, 512 / 768 for large
// ones (below).  With many samples per hit every sample takes one queue atomic
// per block, and the rate of atomics on one word (about 88 per microsecond) bounds the kernel: 1024.
#ifndef SURF_BLOCK_FEW
#define SURF_BLOCK_FEW 256
#endif
#define SURF_BLOCK_MANY 1024
// Large launches with one or two samples per hit: every block sends ONE request for its queue positions, all to the same
// cache line, which takes about 88 requests per microsecond -- 2.1 M blocks of 256 threads (the first bounce of a 1080p x 256 spp
// frame) are 24 ms of requests, and that was the kernel's whole time on three different workloads.  Blocks of 512 / 768 threads
// (8 / 12 waves: three / two blocks fill the 24 wave slots of a CU; 384, 640 and 1024 threads do not divide them and lose 10-25 %)
// halve / third the requests; the block barriers get dearer, so small launches stay at 256.  Which of the two is faster depends
// on the work per hit: 768 with two light samples per hit (dragon: 95.8 -> 84.5 ms per three frames in this kernel; 512: 88.5),
// 512 with one (Cornell box: 107.7 -> 82.9; 768: 86.5).  The instantiations with the correctly rounded arithmetic need their
// registers (5 waves per SIMD, 20 wave slots per CU: 768 threads would leave 8 of them empty): 512 with one sample (Cornell box,
// literal semantics: -20 % of the kernel's time against 256), 256 with two (dragon, literal: 96.7 ms; 512: 102.7; 768: 147).
#ifndef SURF_BLOCK_BIG_ONE
#define SURF_BLOCK_BIG_ONE 512
#endif
#ifndef SURF_BLOCK_BIG_TWO
#define SURF_BLOCK_BIG_TWO 768
#endif
#ifndef SURF_BLOCK_BIG_TWO_EXACT
#define SURF_BLOCK_BIG_TWO_EXACT 256
#endif
#ifndef SURF_BIG_LAUNCH
#define SURF_BIG_LAUNCH (1u << 21)   /* hits from which on the larger blocks still fill the GPU several times over */
#endif
#ifndef SURF_WAVES
#define SURF_WAVES 6   /* waves per SIMD the register allocation aims at, hardware reciprocal / square root (FAST): 80 registers with 7 of them spilled beats 5 waves without spills by 1-1.5 % of the kernel's time (and beats 80 registers without spills, which the kernel gets when the shadow rays' common origin is not kept per sample: the schedule that comes out of that is 3 % slower); 7 waves (72 registers, 34 spilled) lose 25 %, 8 lose 100 % */
#endif
#ifndef SURF_WAVES_EXACT
#define SURF_WAVES_EXACT 5   /* the same with the correctly rounded operations: at 6 waves the spills cost 30 % (dragon, literal semantics) */
#endif
// One thread per path of the bounce that hit something.
// MATS: the material table (at most MAT_LDS_MAX de-duplicated records: any scene whose primitives share a few materials,
// whatever its size) is copied into LDS by every block, and the material of a hit -- read at the light samples and again in
// the closest-hit program, after stores the compiler must assume to alias it -- comes from there instead of through the
// vector memory pipeline.
#define MAT_LDS_MAX 24u
#define MAT_WORDS (sizeof(PtcMaterialValue) / 4u)
#define QLIGHT_LDS_MAX 8u   /* the quad lights travel with the materials: their corners are read once per light sample */
#define QLIGHT_WORDS (sizeof(PtcQuadLight) / 4u)
// Can the ray hit none of the scene's large triangles?  Their planes -- (normal, first vertex), equal ones once -- come with the
// launch parameters.  The first exits of the triangle test in the test's own arithmetic (same products, same sums, so the
// same signs): parallel to the plane, or the plane behind the origin; and, instead of the division, |num| < 0.0009 |den|,
// which puts the quotient safely below the 0.001 under which the test rejects.  A ray for which every plane says "no" is
// marked (SEG_SKIP_BIG) and the traversal kernels do not look at those triangles for it; any other ray is tested in full as before.
__device__ __forceinline__ bool misses_big(const RenderParams& rp, vec3 o, vec3 d) {
    bool any = false;
    for (uint32_t k = 0; k < rp.n_big_planes; ++k) {
        const vec3 n0 = mk3(rp.big_plane[k][0], rp.big_plane[k][1], rp.big_plane[k][2]);
        const vec3 v1 = mk3(rp.big_plane[k][3], rp.big_plane[k][4], rp.big_plane[k][5]);
        const float dn = dot(d, n0);
        const bool flip = dn >= .0f;
        const float den = flip ? -dn : dn;
        const float num0 = dot(v1 - o, n0);
        const float num = flip ? -num0 : num0;
        const bool no = (den == 0.0f) | (num > 0.f) | (fabsf(num) < 0.0009f * fabsf(den));
        any = any || !no;
    }
    return rp.n_big_planes != 0u && !any;
}
// FAST: the shading arithmetic takes the hardware's reciprocal / square root / reciprocal square root (ShMath); the
// path tracer under `intended` semantics only.
// SPH = false: a scene without spheres (the sphere branch of the hit attributes is compiled out: -3 % on the dragon).
// MARK: the scene has large triangles whose planes come with the launch parameters, and the rays made here are tested against
// them (misses_big).  A template parameter and not a run-time test of the plane count: with the code merely present the kernel is
// 5-14 % slower on scenes that have no such triangles (the Cornell boxes).
template <int INTEG, bool MATS = false, bool FAST = false, bool SPH = true, bool MARK = false>
__global__ __launch_bounds__(SURF_BLOCK_MANY, FAST ? SURF_WAVES : SURF_WAVES_EXACT) void k_surface(RenderParams rp, DevScene sc_global, PathState ps, const uint32_t* __restrict__ queue,
                                                       const uint32_t* __restrict__ count_ptr, uint32_t npix_pad, uint32_t sample_begin, uint32_t* __restrict__ segctr,
                                                       uint32_t* __restrict__ next_count,
                                                       uint32_t* __restrict__ gather_queue, uint32_t* __restrict__ gather_count,
                                                       uint32_t* __restrict__ shadow_count, uint32_t first_bounce, uint32_t dense_count, uint32_t* __restrict__ hit_tally) {
    // queue == nullptr ("dense"): there is no hit queue; thread i looks at live path i of the bounce, which may have missed
    // (then it does what the hit compaction does for a miss) and the block notes its hits in hit_tally[blockIdx.x].  Taken when nearly
    // every path of the bounce is expected to hit: the compaction pass and the queue's 12 bytes per path are saved.
    const bool dense = queue == nullptr;
    const uint32_t count = dense ? dense_count : *count_ptr;  // hits of this bounce, written by k_compact_hits
    if (blockIdx.x * blockDim.x >= count) return;
    __shared__ uint32_t mat_copy[MATS ? MAT_LDS_MAX * MAT_WORDS : 1u];
    __shared__ uint32_t qlight_copy[MATS ? QLIGHT_LDS_MAX * QLIGHT_WORDS : 1u];
    DevScene sc = sc_global;
    if (MATS) {
        const uint32_t* g = reinterpret_cast<const uint32_t*>(sc_global.materials);
        for (uint32_t k = threadIdx.x; k < sc_global.n_materials * MAT_WORDS; k += blockDim.x) mat_copy[k] = g[k];
        sc.materials = reinterpret_cast<const PtcMaterialValue*>(mat_copy);
        const uint32_t* gq = reinterpret_cast<const uint32_t*>(sc_global.qlights);
        for (uint32_t k = threadIdx.x; k < sc_global.n_qlights * QLIGHT_WORDS; k += blockDim.x) qlight_copy[k] = gq[k];
        sc.qlights = reinterpret_cast<const PtcQuadLight*>(qlight_copy);
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = i < count;
    bool sample = false, alive = false, gather = false;
    uint32_t p = 0, bounce = 0;
    Segment sg;
    HitAttr at;
    uint32_t at_state = 0;  // position of the path's state
    int hit_ref = HIT_MISS;
    float4 n_o = make_float4(0.f, 0.f, 0.f, 0.f), n_d = n_o, n_t = n_o;  // state of the next segment
    const bool in_range = hit;
    if (in_range) {
        at_state = dense ? i : queue[i];
        hit_ref = ps.hit_ref[at_state];
        hit = hit_ref != HIT_MISS;  // (a queued path always hit)
    }
    if (hit) {
        // camera segments: throughput 1 (k_generate does not write it), path number in an array of its own; later
        // segments carry the path number in the fourth word of their throughput record
        float4 o4, d4;
        load_segment(rp, ps, at_state, o4, d4);
        float4 t4;
        if (first_bounce) {
            t4 = make_float4(1.f, 1.f, 1.f, 0.f);
            p = ps.pid[at_state];
        } else {
            t4 = ps.thr[at_state];
            p = f2u(t4.w);
        }
        sg.o = mk3(o4.x, o4.y, o4.z);
        sg.d = mk3(d4.x, d4.y, d4.z);
        sg.thr = mk3(t4.x, t4.y, t4.z);
        sg.seed = f2u(o4.w);
        sg.depth = f2u(d4.w);
        sg.done = false;
        bounce = sg.depth;  // a live path has gained one unit of depth per segment (every integrator's survivors do `depth++` once)
        hit_attributes<SPH>(sc, hit_ref, ps.hit_t[at_state], sg.o, sg.d, at);
        sample = rp.samples_per_path != 0u && samples_lights<INTEG>(rp, at.mat, sg.depth);
    }
    if (in_range && !hit) {  // dense only: a miss ends the path: radiance = background colour (Common.cu:13-18), as k_compact_hits does
        if (rp.init_L || rp.bg[0] != 0.f || rp.bg[1] != 0.f || rp.bg[2] != 0.f) {
            const uint32_t pm = first_bounce ? ps.pid[at_state] : f2u(ps.thr[at_state].w);
            float4 L = rp.init_L ? make_float4(0.f, 0.f, 0.f, 0.f) : ps.L[pm];
            L.x += rp.bg[0];
            L.y += rp.bg[1];
            L.z += rp.bg[2];
            ps.L[pm] = L;
        }
    }
    if (MATS) __syncthreads();  // the material table is in place (the hit's own loads above were in flight beside the copy)
    // ---- the light samples: term + shadow ray each; a term that is exactly zero adds
    // nothing whether or not the ray is blocked, so no ray is queued for it; any other value
    // (NaN included) is stored and its ray queued; k_shadow zeroes the term of an occluded ray
    // With one or two samples per segment (the usual next-event estimation) the rays wait in
    // registers and get their queue positions together with the survivor / gather queues,
    // in a single round of block barriers.
    const bool few = rp.samples_per_path <= 2u;
    vec3 f_o[2], f_d[2];
    float f_tmax[2] = {0.f, 0.f};
    uint32_t f_slot[2] = {0u, 0u};
    bool f_need[2] = {false, false};
    if (INTEG != INT_ANALYTICDIRECT) {
        if (few) {
#pragma unroll
            for (uint32_t q = 0; q < 2u; ++q) {
                f_o[q] = f_d[q] = mk3(0.f);
                if (sample && q < rp.samples_per_path) {
                    const vec3 term = light_sample<INTEG, FAST>(rp, sc, at, material_of(sc, at.mat), sg.seed, q, f_o[q], f_d[q], f_tmax[q]);
                    f_need[q] = !(term.x == 0.f && term.y == 0.f && term.z == 0.f);
                    f_slot[q] = contrib_slot(ps, i, q);
                    contrib_store(ps, f_slot[q], term);
                }
            }
        } else {
            // MANY_STEP samples per queue reservation: one atomic and one round of barriers for all of them
            // (four for the integrators that are used with dozens of light samples, two where the
            // registers of the rays in flight would spill)
            constexpr uint32_t MANY_STEP = (INTEG == INT_DIRECT || INTEG == INT_RAYTRACER) ? 4u : 2u;
            for (uint32_t q0 = 0; q0 < rp.samples_per_path; q0 += MANY_STEP) {
                vec3 sh_o[MANY_STEP], sh_d[MANY_STEP];
                float sh_tmax[MANY_STEP];
                bool need_ray[MANY_STEP];
                uint32_t slot[MANY_STEP], at_[MANY_STEP];
#pragma unroll
                for (uint32_t k = 0; k < MANY_STEP; ++k) {
                    const uint32_t q = q0 + k;
                    sh_o[k] = sh_d[k] = mk3(0.f);
                    sh_tmax[k] = 0.f;
                    need_ray[k] = false;
                    slot[k] = 0;
                    if (sample && q < rp.samples_per_path) {
                        const vec3 term = light_sample<INTEG, FAST>(rp, sc, at, material_of(sc, at.mat), sg.seed, q, sh_o[k], sh_d[k], sh_tmax[k]);
                        need_ray[k] = !(term.x == 0.f && term.y == 0.f && term.z == 0.f);
                        slot[k] = contrib_slot(ps, i, q);
                        contrib_store(ps, slot[k], term);
                    }
                }
                queue_reserve_streams<MANY_STEP>(need_ray, shadow_count, at_);
#pragma unroll
                for (uint32_t k = 0; k < MANY_STEP; ++k) {
                    if (need_ray[k]) {
                        ps.sh_o[at_[k]] = make_float4(sh_o[k].x, sh_o[k].y, sh_o[k].z, shadow_tmax_word(sh_tmax[k], false));
                        ps.sh_d[at_[k]] = make_float4(sh_d[k].x, sh_d[k].y, sh_d[k].z, u2f(slot[k]));
                    }
                }
            }
        }
    }
    // ---- the rest of the closest-hit program
    if (hit) {
        const vec3 thr_before = sg.thr;
        const uint32_t depth_before = sg.depth;
        SurfaceOut out;
        if (INTEG == INT_RAYTRACER) out = surface_raytracer(rp, sc, at, sg);
        else if (INTEG == INT_ANALYTICDIRECT) out = surface_analytic(rp, sc, at, sg);
        else if (INTEG == INT_DIRECT) out = surface_direct(rp, sc, at, sg);
        else if (INTEG == INT_PATHTRACER2) out = surface_pathtracer2(rp, sc, at, sg);
        else out = surface_pathtracer<FAST>(rp, sc, at, sg);
        gather = out.gather && sample;
        if (gather) {
            // w: does the first-hit rule of pathTracer include the emission (:900-903)
            // one word per waiting segment: the path number, whether the first-hit rule of pathTracer includes the emission
            // (:900-903), and whether the constant part is anything but +0 -- it rarely is (the emission of the surface at a
            // first hit, nothing later with next-event estimation), and only then is its record written and read
            const bool a_set = !(out.a.x == 0.f && out.a.y == 0.f && out.a.z == 0.f) || __builtin_signbitf(out.a.x) || __builtin_signbitf(out.a.y) ||
                               __builtin_signbitf(out.a.z);
            // The word holds the segment's position in this bounce's packed path state, not the path number: the throughput
            // before the segment and the path number are in that state's record, which stays as it is until the next
            // bounce's surface kernel runs -- and that kernel waits for this bounce's gather.
            ps.g_p[i] = at_state | ((!rp.cf.next_event_est || depth_before == 0u) ? GATHER_WITH_LE : 0u) | (a_set ? GATHER_A_SET : 0u);
            if (a_set) ps.g_a[i] = make_float4(out.a.x, out.a.y, out.a.z, 0.f);
        } else {
            // gather wanted but there is not a single light sample (lightsamples 0): the same
            // formulas over zero terms, here
            vec3 radiance = out.a;
            if (out.gather) radiance = gather_radiance<INTEG>(rp, sc, ps, i, out.a, thr_before, !rp.cf.next_event_est || depth_before == 0u);
            float4 L = rp.init_L ? make_float4(0.f, 0.f, 0.f, 0.f) : ps.L[p];
            L.x += radiance.x;  // PinholeCamera.cu:61
            L.y += radiance.y;
            L.z += radiance.z;
            ps.L[p] = L;
        }
        ++bounce;
        // PinholeCamera.cu:66 plus the segment cap and the zero-throughput cull
        alive = !sg.done && sg.depth != rp.max_depth_eff && bounce < rp.bounce_cap;
        if (alive && rp.cull_zero && sg.thr.x == 0.f && sg.thr.y == 0.f && sg.thr.z == 0.f) alive = false;
    }
    // the queue positions of this block are asked for now and used after the survivor's next seed (tea<16>, the
    // longest dependent chain of the kernel) has been computed.  (Asking for the shadow rays' positions earlier
    // still, before the closest-hit program, costs a barrier in the middle of the kernel: +17 ms per frame.)
    __shared__ Reserve4Lds rs_lds;
    Reserve4 rs;
    if (few) reserve4_begin(rs, rs_lds, f_need[0], f_need[1], gather, alive, shadow_count, gather_count, next_count, hit, dense ? hit_tally : nullptr);
    if (alive) {
        uint32_t pix, x, y;
        uint32_t sample;
        path_split(rp, p, pix, sample);
        sample += sample_begin;
        pixel_of(rp, pix, x, y);
        const uint32_t key0 = (x * rp.height + y) * rp.frame_id;
        const uint32_t seed = segment_seed(rp, key0, sample, bounce, segctr, pix);
        n_o = make_float4(sg.o.x, sg.o.y, sg.o.z, u2f(seed));
        const bool skip_big = MARK && misses_big(rp, sg.o, sg.d);
        n_d = make_float4(sg.d.x, sg.d.y, sg.d.z, u2f(sg.depth | (skip_big ? SEG_SKIP_BIG : 0u)));
        n_t = make_float4(sg.thr.x, sg.thr.y, sg.thr.z, u2f(p));
    }
    uint32_t next_at = 0;  // the survivor's position in the next bounce's (packed) path state
    if (few) {
        uint32_t pos[4];
        reserve4_end(rs, rs_lds, pos);
#pragma unroll
        for (uint32_t q = 0; q < 2u; ++q)
            if (f_need[q]) {
                const bool skip_big = MARK && misses_big(rp, f_o[q], f_d[q]);
                ps.sh_o[pos[q]] = make_float4(f_o[q].x, f_o[q].y, f_o[q].z, shadow_tmax_word(f_tmax[q], skip_big));
                ps.sh_d[pos[q]] = make_float4(f_d[q].x, f_d[q].y, f_d[q].z, u2f(f_slot[q]));
            }
        if (gather) gather_queue[pos[2]] = i;
        next_at = pos[3];
    } else {
        queue_append(gather, i, gather_queue, gather_count);
        next_at = queue_reserve(alive, next_count);
    }
    if (alive) {
        ps.o_seed_next[next_at] = n_o;
        ps.d_depth_next[next_at] = n_d;
        ps.thr_next[next_at] = n_t;
    }
}

// After k_shadow: the radiance of the segments that sampled the lights.
#ifndef GATHER_BLOCK
#define GATHER_BLOCK 256   /* small blocks: they run in the wave slots the traversal kernel of the next bounce leaves free */
#endif
template <int INTEG>
__global__ __launch_bounds__(GATHER_BLOCK) void k_gather(RenderParams rp, DevScene sc, PathState ps, const uint32_t* __restrict__ gather_queue,
                                                      const uint32_t* __restrict__ gather_count) {
    const uint32_t i = blockIdx.x * GATHER_BLOCK + threadIdx.x;
    if (i >= *gather_count) return;
    const uint32_t seg = gather_queue[i];  // position in the hit queue
    const uint32_t word = ps.g_p[seg];
    const float4 a4 = (word & GATHER_A_SET) ? ps.g_a[seg] : make_float4(0.f, 0.f, 0.f, 0.f);
    // throughput before the segment and path number: the segment's record in the bounce's path state (camera segments:
    // throughput 1, path number in an array of its own)
    const uint32_t at_state = word & ~(GATHER_WITH_LE | GATHER_A_SET);
    const float4 t4 = rp.camera_rays ? make_float4(1.f, 1.f, 1.f, 0.f) : ps.thr[at_state];
    const uint32_t p = rp.camera_rays ? ps.pid[at_state] : f2u(t4.w);
    const vec3 a = mk3(a4.x, a4.y, a4.z), thr = mk3(t4.x, t4.y, t4.z);
    const vec3 radiance = gather_radiance<INTEG>(rp, sc, ps, seg, a, thr, (word & GATHER_WITH_LE) != 0u);
    float4 L = rp.init_L ? make_float4(0.f, 0.f, 0.f, 0.f) : ps.L[p];
    L.x += radiance.x;  // PinholeCamera.cu:61
    L.y += radiance.y;
    L.z += radiance.z;
    ps.L[p] = L;
}

// -------------------------------------------------------------- accumulate --
__global__ __launch_bounds__(BLOCK) void k_accumulate(RenderParams rp, PathState ps, uint32_t npix_pad, uint32_t n_samples, float* __restrict__ acc, int replace) {
    const uint32_t pix = blockIdx.x * BLOCK + threadIdx.x;
    if (pix >= npix_pad) return;
    uint32_t x, y;
    if (!pixel_of(rp, pix, x, y)) return;
    float* a = acc + 3 * ((size_t)y * rp.width + x);
    if (replace) {  // the path started from the running sum, see k_generate
        const float4 L = ps.L[path_number(rp, pix, 0u)];
        a[0] = L.x; a[1] = L.y; a[2] = L.z;
        return;
    }
    float r = a[0], g = a[1], b = a[2];
    for (uint32_t s = 0; s < n_samples; ++s) {
        const float4 L = ps.L[path_number(rp, pix, s)];
        r += L.x; g += L.y; b += L.z;
    }
    a[0] = r; a[1] = g; a[2] = b;
}

// ----------------------------------------------------------------- resolve --
__global__ __launch_bounds__(BLOCK) void k_resolve(const float* __restrict__ acc, float* __restrict__ result, uint32_t n_pixels, int spp, float gamma, int frame_id) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pixels) return;
    // PinholeCamera.cu:70-71
    vec3 r = mk3(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]) / (float)spp;
    r = mk3(powf(r.x, 1.0f / gamma), powf(r.y, 1.0f / gamma), powf(r.z, 1.0f / gamma));
    if (frame_id != 1) {  // :75-80
        const float u = 1.0f / (float)frame_id;
        const vec3 old = mk3(result[3 * i], result[3 * i + 1], result[3 * i + 2]);
        r = lerp3(old, r, u);
    }
    result[3 * i] = r.x;
    result[3 * i + 1] = r.y;
    result[3 * i + 2] = r.z;
}

// -------------------------------------------------------------- hit records --
__global__ __launch_bounds__(BLOCK) void k_hit_records(RenderParams rp, DevScene sc, PathState ps, uint32_t count, PtcHitRecord* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;  // position of a camera path of sample 0
    if (i >= count) return;
    uint32_t x, y;
    if (!pixel_of(rp, ps.pid[i], x, y)) return;
    PtcHitRecord rec;
    rec.prim = -1;
    rec.t = rec.u = rec.v = 0.f;
    const int ref = ps.hit_ref[i];
    if (ref != HIT_MISS) {
        rec.t = ps.hit_t[i];
        if (ref >= 0) {
            float4 o4, d4;
            load_segment(rp, ps, i, o4, d4);
            const float4* q = sc.tri_q + 3 * (size_t)ref;
            const Cand c = tri_candidate(q[0], q[1], q[2], mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), 0.f, PT_RAY_MAX);
            rec.u = c.u;
            rec.v = c.v;
            rec.prim = sc.tri_prim[ref];
        } else {
            rec.prim = sc.sph_prim[HIT_SPHERE_SLOT(ref)];
        }
    }
    out[(size_t)y * rp.width + x] = rec;
}

// ------------------------------------------------------------- test hooks --
// Global primitive index and distance of the closest hits of entries [0, count) (probe order).
__global__ __launch_bounds__(BLOCK) void k_probe_hits(DevScene sc, PathState ps, uint32_t count, int32_t* __restrict__ prim, float* __restrict__ t) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const int ref = ps.hit_ref[i];
    prim[i] = ref == HIT_MISS ? -1 : ref >= 0 ? sc.tri_prim[ref] : sc.sph_prim[HIT_SPHERE_SLOT(ref)];
    t[i] = ref == HIT_MISS ? 0.f : ps.hit_t[i];
}

// The shading helpers on caller-made inputs (16 floats in, 4 out per item); see ptc_eval_shading.
__global__ __launch_bounds__(BLOCK) void k_eval_shading(int fn, uint32_t n, const float* __restrict__ in, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* a = in + 16 * (size_t)i;
    const vec3 v0 = mk3(a[0], a[1], a[2]), v1 = mk3(a[3], a[4], a[5]), v2 = mk3(a[6], a[7], a[8]), v3 = mk3(a[9], a[10], a[11]);
    vec3 r = mk3(0.f);
    // the flavour the path tracer runs under `intended` semantics
    constexpr bool F = PT_SHADING_FAST != 0;
    typedef ShMath<F> M;
    if (fn == 0) {
        r = phong_brdf<F>(v0, v1, a[12], v2, v3);
    } else if (fn == 1) {
        r = ggx_brdf<F>(v0, v1, v2, a[12], v3);
    } else if (fn == 2) {
        r = mk3(ggx_pdf<F>(v0, v1, v2, a[12], a[13]), 0.f, 0.f);
    } else if (fn == 3) {
        PolarAngle theta = angle_zero();
        float phi = 0.f;
        compute_polar_angles<F>(1, (uint32_t)a[8], phi, theta, a[3], a[4], a[5], a[6], a[7]);
        vec3 u = mk3(0.f), v = mk3(0.f);
        gen_coord_frame<F>(1, v0, u, v);
        r = spherical_dir<F>(theta, phi, u, v, v0);
    } else if (fn == 4) {
        const float zeta_1 = a[6], zeta_2 = a[7], roughness = a[8];
        const PolarAngle theta_h = angle_atan<F>(M::div(roughness * M::sqrt(zeta_1), M::sqrt(1.0f - zeta_1)), false);
        vec3 u = mk3(0.f), v = mk3(0.f);
        gen_coord_frame<F>(1, v0, u, v);
        r = M::normalize(reflect(-v1, spherical_dir<F>(theta_h, zeta_2, u, v, v0)));
    }
    out[4 * (size_t)i] = r.x;
    out[4 * (size_t)i + 1] = r.y;
    out[4 * (size_t)i + 2] = r.z;
    out[4 * (size_t)i + 3] = 0.f;
}

// --------------------------------------------------------------- launchers --
static inline uint32_t blocks_for(uint64_t n, uint32_t block = BLOCK) { return (uint32_t)((n + block - 1) / block); }

void wf_generate(hipStream_t s, const RenderParams& rp, const PathState& ps, uint32_t npix_pad, uint32_t n_paths, uint32_t sample_begin,
                 const float* acc, uint32_t* segctr, uint32_t* pixseed, int seed_L_from_acc) {
    hipLaunchKernelGGL(k_pixel_seeds, dim3(blocks_for(npix_pad)), dim3(BLOCK), 0, s, rp, npix_pad, pixseed);
    hipLaunchKernelGGL(k_generate, dim3(blocks_for(n_paths, CAM_BLOCK)), dim3(CAM_BLOCK), 0, s, rp, ps, npix_pad, n_paths, sample_begin, acc, segctr, pixseed,
                       seed_L_from_acc);
}

// Grid and per-wave span of a traversal launch over (at most) n rays: enough waves to
// fill the chip a few times over when n is large, one ray per lane when it is small.
static void trav_geometry(uint64_t n, uint32_t resident_waves, uint32_t tune_chunk, uint32_t& blocks, uint32_t& wave_chunk) {
    uint64_t chunk = n / ((uint64_t)resident_waves * 2u);
    chunk = (chunk / 64u) * 64u;
    // 512 rays per reservation; 1024 for launches of at least 16 such rounds per resident wave (the same span for every
    // launch, dragon frame: 256 -11 %, 512 0, 1024 +0.7 %, 2048 0, 4096 -4 %: fewer atomics against a longer tail; 1024
    // for the large launches only: +2 %; 768 / 1536 and 8 / 32 rounds: the same)
    chunk = std::max<uint64_t>(64u, std::min<uint64_t>(chunk, n >= (uint64_t)resident_waves * 16384u ? 1024u : 512u));
    if (tune_chunk) chunk = tune_chunk;
    wave_chunk = (uint32_t)chunk;
    const uint64_t waves = (n + chunk - 1) / chunk;
    blocks = (uint32_t)std::min<uint64_t>((waves + 3) / 4, std::max<uint32_t>(resident_waves / 4u, 1u));
}

// Resident waves of the deferred-test kernels (their LDS footprint differs from k_extend's).
static uint32_t deferred_resident_waves() {
    static uint32_t cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cache[dev] == 0u) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_deferred<false, false, false>, BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 3;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cache[dev] = (uint32_t)per_cu * (uint32_t)cus * (BLOCK / 64);
    }
    return cache[dev];
}

// Automatic kernel choice by hierarchy size (measured on the small scenes, whose trees come from the SAH sweep):
// with a handful of nodes every ray does the same few steps and the one-ray-per-lane loop wins (scene 2
// sphere.test, 6 nodes: static 830, refilling rounds 730 Mrays/s); a few more and the rays' step counts differ
// enough for refilling waves to pay, for the closest-hit rays first (Cornell box + GGX sphere, 8 nodes: extension
// 184 ms refilling / 192 ms static, shadow 62 / 53 ms; cornellRR, 13 nodes: 53 / 60 and 21 / 24 ms); real meshes
// take the deferred tests over the four-wide nodes.
// A scene whose traversal arrays fit the LDS copy of the refilling kernels (Cornell boxes, scene 2) takes those for
// its shadow rays whatever the node count, and for its extension rays above 6 nodes (Cornell box + GGX sphere:
// 433 -> 390 ms per frame, extension 174 -> 106 ms, shadow 51 -> 39 ms; cornellRR 117.5 -> 97 ms; scene 2
// sphere.test 5.7 -> 4.9 ms, its camera rays stay with the static kernel: 0.2 against 0.7 ms).
#define STATIC_EXT_NODES 6u
#define STATIC_SHADOW_NODES 8u
#define TINY_BVH_NODES 64u
// The traversal arrays of the scene fit the LDS copy of the small-scene kernels (and there is a hierarchy to copy).
#ifndef LDS_SCENE_OFF
#define LDS_SCENE_OFF 0
#endif
static bool scene_fits_lds(const DevScene& sc) {
    return !LDS_SCENE_OFF && sc.n_nodes != 0u && sc.n_nodes <= LDS_SCENE_NODES && sc.nodes != nullptr && lds_scene_bytes(sc) <= LDS_SCENE_BYTES;
}

#ifndef SPHERES_FIRST_OFF
#define SPHERES_FIRST_OFF 0
#endif
static bool spheres_first(const DevScene& sc) { return !SPHERES_FIRST_OFF && sc.spheres_first_root != REF_EMPTY_WORD && sc.n_spheres >= 1u && sc.n_spheres <= 2u; }

static uint32_t auto_mode(const DevScene& sc, bool shadow) {
    if (sc.n_nodes > TINY_BVH_NODES) return 6u;
    if (scene_fits_lds(sc)) return (!shadow && sc.n_nodes <= STATIC_EXT_NODES) ? 0u : 1u;
    return sc.n_nodes <= (shadow ? STATIC_SHADOW_NODES : STATIC_EXT_NODES) ? 0u : 1u;
}

// The item word of the deferred-test kernels holds 25 bits of primitive slot.
static bool deferred_fits(const DevScene& sc) { return sc.n_tris < (1u << 25) && sc.n_spheres < (1u << 25); }

void wf_extend(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t count, uint32_t* work_counter, uint32_t resident_waves, bool coherent, bool count_trav,
               unsigned long long* trav_counters) {
    if (count == 0) return;
    uint32_t blocks, wave_chunk;
    // mode 5 (default): see auto_mode; the deferred tests serve the camera rays as well (5 % faster than the
    // static kernel on the dragon)
    uint32_t mode = rp.tune_mode == 3 ? (coherent ? 0u : 1u) : rp.tune_mode == 5 ? auto_mode(sc, false) : rp.tune_mode;
    const bool wide = mode == 6 && sc.wnodes != nullptr;  // mode 6: deferred tests over the four-wide nodes
    if (mode == 6) mode = 4;
    if (mode == 4 && !deferred_fits(sc)) mode = 1;
    trav_geometry(count, mode == 4 ? deferred_resident_waves() : resident_waves, rp.tune_chunk, blocks, wave_chunk);
    // mode 3: static for the camera rays (8x8 pixel bundles that finish together), refilling
    // rounds for the scattered rays, which finish at very different times
    if (rp.brute_force)
        hipLaunchKernelGGL(k_extend_brute, dim3(blocks_for(count)), dim3(BLOCK), 0, s, rp, sc, ps, count);
    else if (mode == 4 && wide && count_trav)
        hipLaunchKernelGGL((k_trace_deferred<false, true, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, nullptr, count, work_counter, wave_chunk, trav_counters, nullptr);
    else if (mode == 4 && wide && sc.n_spheres == 0u)
        hipLaunchKernelGGL((k_trace_deferred<false, false, true, false>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, nullptr, count, work_counter, wave_chunk, trav_counters, nullptr);
    else if (mode == 4 && wide)
        hipLaunchKernelGGL((k_trace_deferred<false, false, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, nullptr, count, work_counter, wave_chunk, trav_counters, nullptr);
    else if (mode == 4 && count_trav)
        hipLaunchKernelGGL((k_trace_deferred<false, true, false>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, nullptr, count, work_counter, wave_chunk, trav_counters, nullptr);
    else if (mode == 4)
        hipLaunchKernelGGL((k_trace_deferred<false, false, false>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, nullptr, count, work_counter, wave_chunk, trav_counters, nullptr);
    else if (mode == 0 && count_trav)
        hipLaunchKernelGGL(k_extend_static<true>, dim3(blocks_for(count)), dim3(BLOCK), 0, s, rp, sc, ps, count, trav_counters);
    else if (mode == 0)
        hipLaunchKernelGGL(k_extend_static<false>, dim3(blocks_for(count)), dim3(BLOCK), 0, s, rp, sc, ps, count, trav_counters);
    else if (count_trav)
        hipLaunchKernelGGL(k_extend<true>, dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count, work_counter, wave_chunk, trav_counters);
    else if (scene_fits_lds(sc) && spheres_first(sc))
        hipLaunchKernelGGL((k_extend<false, true, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count, work_counter, wave_chunk, trav_counters);
    else if (scene_fits_lds(sc))
        hipLaunchKernelGGL((k_extend<false, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count, work_counter, wave_chunk, trav_counters);
    else
        hipLaunchKernelGGL(k_extend<false>, dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count, work_counter, wave_chunk, trav_counters);
}

// Traces the shadow rays k_surface queued (their number is in device memory, at most `max_rays`).
void wf_shadow(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, const uint32_t* count_ptr, uint64_t max_rays,
               uint32_t* work_counter, uint32_t resident_waves, bool count_trav, unsigned long long* trav_counters, unsigned long long* shadow_total) {
    if (max_rays == 0) return;
    uint32_t blocks, wave_chunk;
    uint32_t smode = rp.tune_shadow == 5 ? auto_mode(sc, true) : rp.tune_shadow;
    const bool wide = smode == 6 && sc.wnodes != nullptr;
    if (smode == 6) smode = 4u;
    if (smode == 4 && !deferred_fits(sc)) smode = 1u;
    trav_geometry(max_rays, smode == 4 ? deferred_resident_waves() : resident_waves, rp.tune_chunk, blocks, wave_chunk);
    if (rp.brute_force)
        hipLaunchKernelGGL(k_shadow_brute, dim3(blocks_for(max_rays)), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, shadow_total);
    else if (smode == 4 && wide && count_trav)
        hipLaunchKernelGGL((k_trace_deferred<true, true, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, 0u, work_counter, wave_chunk, trav_counters, shadow_total);
    else if (smode == 4 && wide && sc.n_spheres == 0u)
        hipLaunchKernelGGL((k_trace_deferred<true, false, true, false>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, 0u, work_counter, wave_chunk, trav_counters, shadow_total);
    else if (smode == 4 && wide)
        hipLaunchKernelGGL((k_trace_deferred<true, false, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, 0u, work_counter, wave_chunk, trav_counters, shadow_total);
    else if (smode == 4 && count_trav)
        hipLaunchKernelGGL((k_trace_deferred<true, true, false>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, 0u, work_counter, wave_chunk, trav_counters, shadow_total);
    else if (smode == 4)
        hipLaunchKernelGGL((k_trace_deferred<true, false, false>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, 0u, work_counter, wave_chunk, trav_counters, shadow_total);
    else if (smode == 0 && count_trav)
        hipLaunchKernelGGL(k_shadow_static<true>, dim3(blocks_for(max_rays)), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, trav_counters, shadow_total);
    else if (smode == 0)
        hipLaunchKernelGGL(k_shadow_static<false>, dim3(blocks_for(max_rays)), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, trav_counters, shadow_total);
    else if (count_trav)
        hipLaunchKernelGGL(k_shadow<true>, dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, work_counter, wave_chunk, trav_counters, shadow_total);
    else if (scene_fits_lds(sc) && spheres_first(sc))
        hipLaunchKernelGGL((k_shadow<false, true, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, work_counter, wave_chunk, trav_counters, shadow_total);
    else if (scene_fits_lds(sc))
        hipLaunchKernelGGL((k_shadow<false, true>), dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, work_counter, wave_chunk, trav_counters, shadow_total);
    else
        hipLaunchKernelGGL(k_shadow<false>, dim3(blocks), dim3(BLOCK), 0, s, rp, sc, ps, count_ptr, work_counter, wave_chunk, trav_counters, shadow_total);
}

// Waves of the traversal kernels that the device keeps resident at once.
uint32_t wf_resident_waves(int device) {
    (void)deferred_resident_waves();  // fill its per-device cache now (context creation), not inside a render call
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_extend<false>, BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return (uint32_t)per_cu * (uint32_t)cus * (BLOCK / 64);
}

#ifndef CLASS_TABLE_OFF
#define CLASS_TABLE_OFF 0
#endif
// Queue of the paths whose extension ray hit something; misses end there.
void wf_compact_hits(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t count, uint32_t* hit_queue, uint32_t* hit_count) {
    if (count == 0) return;
    const dim3 g(blocks_for(count, GEN_BLOCK * GEN_CHUNKS)), b(GEN_BLOCK);
    if (rp.group_hits && !CLASS_TABLE_OFF && sc.n_tris + sc.n_spheres <= CLASS_TABLE_PRIMS)
        hipLaunchKernelGGL((k_compact_hits<true, true>), g, b, 0, s, rp, sc, ps, count, hit_queue, hit_count);
    else if (rp.group_hits) hipLaunchKernelGGL(k_compact_hits<true>, g, b, 0, s, rp, sc, ps, count, hit_queue, hit_count);
    else hipLaunchKernelGGL(k_compact_hits<false>, g, b, 0, s, rp, sc, ps, count, hit_queue, hit_count);
}

#ifndef MAT_LDS_OFF
#define MAT_LDS_OFF 0
#endif

// `count_ptr` (device) holds the number of queued paths; `max_count` bounds it and sizes the grid.
bool wf_surface_can_skip_queue(const RenderParams& rp) { return rp.samples_per_path <= 2u; }

// *total += the n words at `words`.
__global__ __launch_bounds__(1024) void k_sum_words(const uint32_t* __restrict__ words, uint32_t n, uint32_t* __restrict__ total) {
    __shared__ uint32_t part[16];
    uint32_t sum = 0;
    for (uint32_t k = blockIdx.x * 1024u + threadIdx.x; k < n; k += gridDim.x * 1024u) sum += words[k];
    for (int off = 32; off > 0; off >>= 1) sum += (uint32_t)__shfl_down((int)sum, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        if (t) atomicAdd(total, t);
    }
}

void wf_surface(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, const uint32_t* queue, uint32_t* count_ptr,
                uint32_t max_count, uint32_t npix_pad, uint32_t sample_begin, uint32_t* segctr, uint32_t* next_count,
                uint32_t* gather_queue, uint32_t* gather_count, uint32_t* shadow_count, bool first_bounce, uint32_t* block_tally) {
    if (max_count == 0) return;
    // hardware reciprocal / square root in the shading arithmetic: the path tracer under `intended` semantics
    // (`literal` keeps the correctly rounded operations: there a last bit decides between 1 and NaN)
    const bool fast = PT_SHADING_FAST && rp.semantics == 1 && rp.integrator == INT_PATHTRACER;
    const uint32_t block = rp.samples_per_path > 2u ? SURF_BLOCK_MANY
                           : max_count < SURF_BIG_LAUNCH ? SURF_BLOCK_FEW
                           : rp.samples_per_path == 2u ? (fast ? SURF_BLOCK_BIG_TWO : SURF_BLOCK_BIG_TWO_EXACT) : SURF_BLOCK_BIG_ONE;
    const dim3 g(blocks_for(max_count, block)), b(block);
    const bool mats = !MAT_LDS_OFF && sc.n_materials != 0u && sc.n_materials <= MAT_LDS_MAX && sc.n_qlights <= QLIGHT_LDS_MAX;
#define PT_SURFACE_LAUNCH(I, M, F, ...)                                                                                                                       \
    hipLaunchKernelGGL((k_surface<I, M, F, ##__VA_ARGS__>), g, b, 0, s, rp, sc, ps, queue, count_ptr, npix_pad, sample_begin, segctr, next_count, gather_queue, gather_count, \
                       shadow_count, first_bounce ? 1u : 0u, max_count, block_tally)
#define PT_SURFACE(I)                           \
    do {                                        \
        if (mats) PT_SURFACE_LAUNCH(I, true, false);  \
        else PT_SURFACE_LAUNCH(I, false, false);      \
    } while (0)
    switch (rp.integrator) {
        case INT_RAYTRACER: PT_SURFACE(INT_RAYTRACER); break;
        case INT_ANALYTICDIRECT: PT_SURFACE(INT_ANALYTICDIRECT); break;
        case INT_DIRECT: PT_SURFACE(INT_DIRECT); break;
        case INT_PATHTRACER:
            // (marked rays: scenes of triangles only, see misses_big)
            if (fast && mats && rp.n_big_planes != 0u && sc.n_spheres == 0u) PT_SURFACE_LAUNCH(INT_PATHTRACER, true, true, false, true);
            else if (fast && mats && sc.n_spheres == 0u) PT_SURFACE_LAUNCH(INT_PATHTRACER, true, true, false);
            else if (fast && mats) PT_SURFACE_LAUNCH(INT_PATHTRACER, true, true);
            else if (fast) PT_SURFACE_LAUNCH(INT_PATHTRACER, false, true);
            else if (mats && rp.n_big_planes != 0u && sc.n_spheres == 0u) PT_SURFACE_LAUNCH(INT_PATHTRACER, true, false, false, true);
            else PT_SURFACE(INT_PATHTRACER);
            break;
        case INT_PATHTRACER2: PT_SURFACE(INT_PATHTRACER2); break;
        default: break;
    }
#undef PT_SURFACE_LAUNCH
#undef PT_SURFACE
    if (!queue) {  // the blocks' hit counts -> *count_ptr
        const uint32_t nb = blocks_for(max_count, block);
        hipLaunchKernelGGL(k_sum_words, dim3(std::min<uint32_t>(blocks_for(nb, 1024u), 256u)), dim3(1024), 0, s, block_tally, nb, count_ptr);
    }
}

// `gather_count` (device) holds the number of queued paths; `max_count` bounds it and sizes the grid.
void wf_gather(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, const uint32_t* gather_queue, const uint32_t* gather_count,
               uint32_t max_count) {
    if (max_count == 0) return;
    const dim3 g(blocks_for(max_count, GATHER_BLOCK)), b(GATHER_BLOCK);
    switch (rp.integrator) {
        case INT_RAYTRACER: hipLaunchKernelGGL(k_gather<INT_RAYTRACER>, g, b, 0, s, rp, sc, ps, gather_queue, gather_count); break;
        case INT_DIRECT: hipLaunchKernelGGL(k_gather<INT_DIRECT>, g, b, 0, s, rp, sc, ps, gather_queue, gather_count); break;
        case INT_PATHTRACER: hipLaunchKernelGGL(k_gather<INT_PATHTRACER>, g, b, 0, s, rp, sc, ps, gather_queue, gather_count); break;
        case INT_PATHTRACER2: hipLaunchKernelGGL(k_gather<INT_PATHTRACER2>, g, b, 0, s, rp, sc, ps, gather_queue, gather_count); break;
        default: break;
    }
}

void wf_accumulate(hipStream_t s, const RenderParams& rp, const PathState& ps, uint32_t npix_pad, uint32_t n_samples, float* acc, int replace) {
    hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(npix_pad)), dim3(BLOCK), 0, s, rp, ps, npix_pad, n_samples, acc, replace);
}

void wf_resolve(hipStream_t s, const float* acc, float* result, uint32_t n_pixels, int spp, float gamma, int frame_id) {
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(n_pixels)), dim3(BLOCK), 0, s, acc, result, n_pixels, spp, gamma, frame_id);
}

void wf_probe_hits(hipStream_t s, const DevScene& sc, const PathState& ps, uint32_t count, int32_t* prim, float* t) {
    if (count == 0) return;
    hipLaunchKernelGGL(k_probe_hits, dim3(blocks_for(count)), dim3(BLOCK), 0, s, sc, ps, count, prim, t);
}

void wf_eval_shading(hipStream_t s, int fn, uint32_t n, const float* in, float* out) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_eval_shading, dim3(blocks_for(n)), dim3(BLOCK), 0, s, fn, n, in, out);
}

void wf_hit_records(hipStream_t s, const RenderParams& rp, const DevScene& sc, const PathState& ps, uint32_t count, PtcHitRecord* out) {
    if (count == 0) return;
    hipLaunchKernelGGL(k_hit_records, dim3(blocks_for(count)), dim3(BLOCK), 0, s, rp, sc, ps, count, out);
}
