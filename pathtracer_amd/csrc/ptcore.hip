// ptcore.hip -- the C ABI of include/ptcore.h: context, uploads, wavefront
// scheduling of the kernels in wavefront.hip and the BVH build of bvh_build.hip.
//
// The control flow of one render is the reference's Renderer::run ->
// context->launch (Renderer.cpp:27-37) unrolled into batches: a batch holds
// `n` samples of every pixel; its paths are generated, then bounce by bounce
// extended, processed at their hits (light samples, next direction), shadow-tested
// and gathered while the packed set of live paths shrinks, and finally summed into
// the per-pixel radiance sums in sample order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ptcore.h"
#include "wavefront.h"

#ifndef BIG_FIRST_OFF
#define BIG_FIRST_OFF 0   /* 1: no ray is marked as unable to hit the large triangles (A/B builds) */
#endif

// Device memory of ptc_render_range_static: the per-path arrays of a batch and the two per-pixel seed
// arrays live here instead of in allocations.  Every GPU the library runs on holds its own copy of the
// array at its own address; host code only ever uses the address hipGetSymbolAddress returned for the
// device at hand (g_scratch_ptr), never the array's name.
#define PTC_STATIC_SCRATCH_BYTES ((size_t)PTC_STATIC_SCRATCH_MIB << 20)
static __device__ __align__(256) unsigned char g_static_scratch[PTC_STATIC_SCRATCH_BYTES];

namespace {
std::string g_create_error;
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};  // per device: address of g_static_scratch there
std::mutex g_scratch_mutex;
// One ptc_render_range_static call at a time per device: the array is shared by every context on that GPU, and two
// calls that overlapped would write each other's path state.  Taken on entry, released after the call's last
// kernel has finished.
std::atomic<bool> g_scratch_busy[kMaxDevices];
struct DeviceBuffer {  // a temporary device allocation that every return path releases
    void* p = nullptr;
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
};
const char* kIntegrators[] = {"raytracer", "analyticdirect", "direct", "pathtracer", "pathtracer2"};
}  // namespace

struct ptc_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t aux_stream = nullptr;           // k_gather of a bounce runs here, beside the next bounce's extension traversal
    hipEvent_t ev_shadow = nullptr, ev_gather = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // build timing
    bool overlap_gather = true;
    int sort_rays = 0;                          // PTC_OPT_SORT_RAYS
    void* sort_temp = nullptr;                  // rocPRIM work space of the ray sort, for sort_temp_capacity paths
    size_t sort_temp_bytes = 0;
    uint32_t sort_temp_capacity = 0;
    std::string error;

    // scene (host copies of what was uploaded)
    PtcConfig cfg{};
    bool have_cfg = false;
    int integrator = -1;
    uint32_t light_samples = 1, light_stratify = 0, next_event_est = 0, sampling_method = 0;
    int32_t spp = 1;
    float bg[3] = {0.f, 0.f, 0.f};
    std::vector<PtcTriangle> tris;
    std::vector<PtcSphere> spheres;
    std::vector<PtcPointLight> plights;
    std::vector<PtcDirectionalLight> dlights;
    std::vector<PtcQuadLight> qlights;
    bool non_diffuse = false;
    bool group_hits = false;                    // PTC_OPT_GROUP_HITS
    bool dense_surface = true;                  // PTC_OPT_DENSE_SURFACE
    bool mark_rays = true;                      // PTC_OPT_MARK_RAYS
    float hit_ratio[16] = {0.f};                // hits per live path in the last batch: [0] camera rays, [b + 1] bounce b of later segments (0 = not known yet)
    bool mixed_hits = false;                    // hits fall into more than one shading class (primitive kind x BRDF)

    // device scene
    PtcTriangle* d_tris = nullptr;
    PtcSphere* d_spheres = nullptr;
    PtcMaterialValue* d_materials = nullptr;
    uint32_t n_materials = 0;
    PtcPointLight* d_plights = nullptr;
    PtcDirectionalLight* d_dlights = nullptr;
    PtcQuadLight* d_qlights = nullptr;
    BvhBuildOutput bvh;
    BvhBuildOutput bvh_any;   // builder 3 only: the hierarchy (and primitive order) the any-hit rays traverse
    bool two_trees = false;
    bool built = false;

    // render targets
    uint32_t width = 0, height = 0, npix_pad = 0;
    float* acc_own = nullptr;
    float* acc = nullptr;  // acc_own or the caller's buffer
    void* acc_ext = nullptr;
    size_t acc_ext_bytes = 0;
    float* result = nullptr;
    uint32_t* segctr = nullptr;
    uint32_t* pixseed = nullptr;                // per padded pixel: start of its jitter stream in the current frame
    int32_t ref_next_sample = 0;  // reference seeding: the sample the per-pixel history is at
    PtcHitRecord* d_hits = nullptr;

    // wavefront state: every per-path array is a slice of one pool (one allocation, or the
    // library's static scratch array while ptc_render_range_static runs)
    PathState ps{};
    void* path_pool = nullptr;
    size_t path_pool_bytes = 0;
    bool no_alloc = false;                      // inside ptc_render_range_static: nothing may allocate
    uint32_t contrib_samples = 0;
    uint32_t* hitq = nullptr;                   // paths of this bounce that hit something
    uint32_t* gatherq = nullptr;                // segments of this bounce whose radiance waits for the light terms
    uint32_t* qcount = nullptr;                 // two groups {hit, next, shadow, extend work, shadow work, gather} used alternately
    uint32_t resident_waves = 6144;
    unsigned long long* counters = nullptr;     // [0..3] traversal nodes/prims (extension, shadow), [4] shadow rays
    uint32_t* h_count = nullptr;                // pinned

    // options
    int semantics = 0, seeding = 0, cull_zero = 1, profile = 0, count_trav = 0, brute_force = 0;
    uint32_t profile_min_items = 0;             // PTC_OPT_PROFILE > 1: only bounces with at least this many live paths are timed
    uint32_t bounce_cap = 1024;
    uint64_t batch_paths = 512u << 20;
    int bvh_builder = 3;
    uint32_t tune_refill = 24, tune_mode = 5, tune_shadow = 5, tune_chunk = 0;

    PtcStats stats{};
    std::vector<hipEvent_t> ev_pool;
    struct Span { int kind; hipEvent_t a, b; };
    std::vector<Span> spans;
    size_t ev_used = 0;
};

namespace {

int fail(ptc_context* c, int code, const std::string& msg) {
    if (c) c->error = msg;
    return code;
}

#define CHK(call)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return fail(ctx, PTC_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
hipError_t upload(T*& dst, const std::vector<T>& src, hipStream_t s) {
    if (dst) {
        (void)hipFree(dst);
        dst = nullptr;
    }
    hipError_t e = hipMalloc((void**)&dst, std::max<size_t>(src.size(), 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (!src.empty()) e = hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
}

void freeTargets(ptc_context* c) {
    for (void* p : {(void*)c->acc_own, (void*)c->result, (void*)c->segctr, (void*)c->pixseed, (void*)c->d_hits})
        if (p) (void)hipFree(p);
    c->acc_own = c->result = nullptr;
    c->segctr = nullptr;
    c->pixseed = nullptr;
    c->d_hits = nullptr;
    c->width = c->height = c->npix_pad = 0;
}

void freePaths(ptc_context* c) {
    if (c->path_pool) (void)hipFree(c->path_pool);
    c->path_pool = nullptr;
    c->path_pool_bytes = 0;
    c->ps = PathState{};
    c->hitq = c->gatherq = nullptr;
    c->contrib_samples = 0;
}

// Pixels of the image padded to whole 64x64-pixel blocks of 8x8 tiles (the path numbering's unit).
uint64_t paddedPixels(uint32_t w, uint32_t h) { return (uint64_t)((w + 63u) / 64u) * ((h + 63u) / 64u) * 4096u; }

int ensureTargets(ptc_context* ctx, uint32_t w, uint32_t h) {
    if (w == 0 || h == 0) return fail(ctx, PTC_ERR_INVALID, "image size must be non-zero");
    const size_t n = (size_t)w * h;
    // a caller-owned sum buffer must hold this image, whenever it was bound
    if (ctx->acc_ext && ctx->acc_ext_bytes < n * 3 * sizeof(float))
        return fail(ctx, PTC_ERR_INVALID, "the bound sum buffer is too small for this image size");
    if (ctx->width == w && ctx->height == h) {
        ctx->acc = ctx->acc_ext ? (float*)ctx->acc_ext : ctx->acc_own;
        return PTC_OK;
    }
    freeTargets(ctx);
    if (paddedPixels(w, h) > 0xFFFFF000ull) return fail(ctx, PTC_ERR_INVALID, "image too large");
    const uint32_t tiles = (uint32_t)(paddedPixels(w, h) / 64u);
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    step(hipMalloc((void**)&ctx->acc_own, n * 3 * sizeof(float)));
    step(hipMalloc((void**)&ctx->result, n * 3 * sizeof(float)));
    step(hipMalloc((void**)&ctx->segctr, (size_t)tiles * 64 * sizeof(uint32_t)));
    step(hipMalloc((void**)&ctx->pixseed, (size_t)tiles * 64 * sizeof(uint32_t)));
    step(hipMalloc((void**)&ctx->d_hits, n * sizeof(PtcHitRecord)));
    if (e == hipSuccess) step(hipMemsetAsync(ctx->acc_own, 0, n * 3 * sizeof(float), ctx->stream));
    if (e == hipSuccess) step(hipMemsetAsync(ctx->result, 0, n * 3 * sizeof(float), ctx->stream));
    if (e == hipSuccess) step(hipMemsetAsync(ctx->segctr, 0, (size_t)tiles * 64 * sizeof(uint32_t), ctx->stream));
    if (e != hipSuccess) {
        freeTargets(ctx);  // nothing half allocated stays behind
        return fail(ctx, PTC_ERR_DEVICE, std::string("allocating the render targets: ") + hipGetErrorString(e));
    }
    ctx->width = w;
    ctx->height = h;
    ctx->npix_pad = tiles * 64u;
    ctx->ref_next_sample = 0;
    ctx->acc = ctx->acc_ext ? (float*)ctx->acc_ext : ctx->acc_own;
    return PTC_OK;
}

// Slices of one pool for the per-path arrays of a batch of `capacity` paths with `S` light samples per
// segment; every slice starts on a 256-byte boundary.  base == nullptr: only the size is computed.
size_t carvePaths(char* base, uint32_t capacity, uint32_t S, PathState& ps, uint32_t*& hitq, uint32_t*& gatherq) {
    size_t at = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + at : nullptr;
        at += (std::max<size_t>(bytes, 1) + 255u) & ~(size_t)255u;
        return (void*)p;
    };
    const size_t n = capacity, ns = (size_t)capacity * S;
    ps.o_seed = (float4*)take(n * sizeof(float4));
    ps.d_depth = (float4*)take(n * sizeof(float4));
    ps.thr = (float4*)take(n * sizeof(float4));
    ps.L = (float4*)take(n * sizeof(float4));
    ps.g_p = (uint32_t*)take(n * sizeof(uint32_t));
    ps.g_a = (float4*)take(n * sizeof(float4));
    ps.hit_t = (float*)take(n * sizeof(float));
    ps.hit_ref = (int32_t*)take(n * sizeof(int32_t));
    ps.contrib = (float*)take(ns * 3 * sizeof(float));
    ps.sh_o = (float4*)take(ns * sizeof(float4));
    ps.sh_d = (float4*)take(ns * sizeof(float4));
    ps.pid = (uint32_t*)take(n * sizeof(uint32_t));
    ps.o_seed_next = (float4*)take(n * sizeof(float4));
    ps.d_depth_next = (float4*)take(n * sizeof(float4));
    ps.thr_next = (float4*)take(n * sizeof(float4));
    hitq = (uint32_t*)take(n * sizeof(uint32_t));
    gatherq = (uint32_t*)take(n * sizeof(uint32_t));
    ps.capacity = capacity;
    return at;
}

// Device memory one path of a batch takes (see carvePaths): the packed state twice, L, hit, gather
// record, two queue entries, and per light sample its term and its shadow-ray record.
uint64_t bytesPerPath(uint32_t S) { return 2u * 48u + 4u + 16u + 8u + 4u + 16u + 8u + (uint64_t)S * (12u + 32u); }

int ensurePaths(ptc_context* ctx, uint32_t capacity, uint32_t contrib_samples) {
    if (ctx->ps.capacity >= capacity && ctx->contrib_samples >= contrib_samples) return PTC_OK;
    if (ctx->no_alloc) return fail(ctx, PTC_ERR_INVALID, "the path arrays would have to grow inside a call that may not allocate");
    capacity = std::max(capacity, ctx->ps.capacity);
    contrib_samples = std::max(contrib_samples, ctx->contrib_samples);
    freePaths(ctx);
    PathState ps{};
    uint32_t *hq = nullptr, *gq = nullptr;
    const size_t bytes = carvePaths(nullptr, capacity, contrib_samples, ps, hq, gq);
    const hipError_t e = hipMalloc(&ctx->path_pool, bytes);  // the only allocation: nothing is left half done when it fails
    if (e != hipSuccess) {
        ctx->path_pool = nullptr;
        (void)hipGetLastError();
        return fail(ctx, e == hipErrorOutOfMemory ? PTC_ERR_NOMEM : PTC_ERR_DEVICE, std::string("allocating the path arrays (") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    }
    ctx->path_pool_bytes = bytes;
    carvePaths((char*)ctx->path_pool, capacity, contrib_samples, ctx->ps, ctx->hitq, ctx->gatherq);
    ctx->contrib_samples = contrib_samples;
    return PTC_OK;
}

uint32_t samplesPerSegment(const ptc_context* c, uint32_t root) {
    switch (c->integrator) {
        case INT_RAYTRACER: return (uint32_t)(c->plights.size() + c->dlights.size());
        case INT_ANALYTICDIRECT: return 0;
        case INT_DIRECT: return (uint32_t)c->qlights.size() * root * root;
        case INT_PATHTRACER: return c->cfg.next_event_est ? (uint32_t)c->qlights.size() * root * root : 0u;
        case INT_PATHTRACER2: return (uint32_t)c->qlights.size() * root * root;
    }
    return 0;
}

hipEvent_t takeEvent(ptc_context* c) {
    if (c->ev_used == c->ev_pool.size()) {
        if (c->no_alloc) return nullptr;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

struct Timed {  // brackets a launch with events when profiling is on
    ptc_context* c;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    Timed(ptc_context* c_, int kind_, hipStream_t s_ = nullptr, bool on = true) : c(c_), kind(kind_), s(s_ ? s_ : c_->stream) {
        if (c->profile && on) {
            a = takeEvent(c);
            b = a ? takeEvent(c) : nullptr;
            if (a && b) (void)hipEventRecord(a, s);
        }
    }
    ~Timed() {
        if (a && b) {
            (void)hipEventRecord(b, s);
            c->spans.push_back({kind, a, b});
        }
    }
};

void collectSpans(ptc_context* c) {  // stream must be idle
    for (auto& sp : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            double* dst = sp.kind == 0 ? &c->stats.extend_ms : sp.kind == 1 ? &c->stats.connect_ms : (sp.kind == 2 || sp.kind == 4) ? &c->stats.shade_ms : &c->stats.other_ms;
            *dst += ms;
            if (sp.kind == 2) c->stats.surface_ms += ms;
            if (sp.kind == 4) c->stats.gather_ms += ms;
        }
    }
    c->spans.clear();
    c->ev_used = 0;
}

// Which root the deferred-test kernels start at when they test the large triangles up front (see DevScene::big_count).
void setBigFirst(DevScene& sc, const BvhBuildOutput& b) {
    sc.big_count = b.big_count;
    sc.big_first = b.big_first;
    sc.big_root_bin = b.root_without_big;
    sc.big_root_wide = b.wnodes ? b.wide_root_without_big : REF_EMPTY_WORD;
}

DevScene deviceScene(const ptc_context* c) {
    DevScene sc{};
    sc.tri_q = c->bvh.tri_q;
    sc.tri_nm = c->bvh.tri_nm;
    sc.tri_mat = c->bvh.tri_mat;
    sc.tri_prim = c->bvh.tri_prim;
    sc.spheres = c->bvh.spheres;
    sc.sph_mat = c->bvh.sph_mat;
    sc.sph_prim = c->bvh.sph_prim;
    sc.materials = c->d_materials;
    sc.n_materials = c->n_materials;
    sc.nodes = c->bvh.nodes;
    sc.wnodes = c->bvh.wnodes;
    sc.n_wnodes = c->bvh.n_wnodes;
    sc.n_tris = (uint32_t)c->tris.size();
    sc.n_spheres = (uint32_t)c->spheres.size();
    sc.n_nodes = c->bvh.n_nodes;
    sc.plights = c->d_plights;
    sc.dlights = c->d_dlights;
    sc.qlights = c->d_qlights;
    sc.n_plights = (uint32_t)c->plights.size();
    sc.n_dlights = (uint32_t)c->dlights.size();
    sc.n_qlights = (uint32_t)c->qlights.size();
    sc.spheres_first_root = c->bvh.root_without_spheres;
    setBigFirst(sc, c->bvh);
    return sc;
}

// The scene as the any-hit (shadow) rays see it: with PTC_OPT_BVH_BUILDER 3 a hierarchy of their own, with the
// primitives in that hierarchy's leaf order (nothing but occlusion is read from it).
DevScene deviceSceneAny(const ptc_context* c) {
    DevScene sc = deviceScene(c);
    if (!c->two_trees) return sc;
    sc.tri_q = c->bvh_any.tri_q;
    sc.tri_nm = c->bvh_any.tri_nm;
    sc.tri_mat = c->bvh_any.tri_mat;
    sc.tri_prim = c->bvh_any.tri_prim;
    sc.spheres = c->bvh_any.spheres;
    sc.sph_mat = c->bvh_any.sph_mat;
    sc.sph_prim = c->bvh_any.sph_prim;
    sc.nodes = c->bvh_any.nodes;
    sc.wnodes = c->bvh_any.wnodes;
    sc.n_wnodes = c->bvh_any.n_wnodes;
    sc.n_nodes = c->bvh_any.n_nodes;
    sc.spheres_first_root = c->bvh_any.root_without_spheres;
    setBigFirst(sc, c->bvh_any);
    return sc;
}

int makeParams(ptc_context* ctx, int frame_id, uint32_t w, uint32_t h, RenderParams& rp) {
    if (!ctx->have_cfg) return fail(ctx, PTC_ERR_INVALID, "ptc_set_config has not been called");
    if (ctx->integrator < 0) return fail(ctx, PTC_ERR_INVALID, "ptc_set_integrator has not been called");
    if (!ctx->built) return fail(ctx, PTC_ERR_NOT_BUILT, "ptc_build has not been called since the last geometry upload");
    std::memset(&rp, 0, sizeof(rp));
    rp.cf = ctx->cfg;
    {
        rp.inv_tiles_x = 1.0 / (double)std::max((w + 63u) / 64u, 1u);  // 64x64-pixel blocks per row
        rp.batch_samples = 1u;  // set per batch by renderRange
        rp.inv_batch_samples = 1.0;
    }
    rp.cf.next_event_est = ctx->next_event_est;
    rp.light_samples = ctx->light_samples;
    rp.light_stratify = ctx->light_stratify;
    rp.sampling_method = ctx->sampling_method;
    rp.integrator = ctx->integrator;
    std::memcpy(rp.bg, ctx->bg, sizeof(rp.bg));
    rp.width = w;
    rp.height = h;
    rp.frame_id = (uint32_t)frame_id;
    rp.bounce_cap = ctx->bounce_cap;
    rp.max_depth_eff = std::min<uint32_t>(ctx->cfg.maxDepth, ctx->bounce_cap);
    rp.semantics = ctx->semantics;
    rp.seeding = ctx->seeding;
    // reference seeding: every segment the reference traces advances the pixel's counter (PinholeCamera.cu:56),
    // which seeds all later samples of the pixel, so a dead path has to be traced to its end
    rp.cull_zero = ctx->cull_zero && ctx->seeding == 1;
    rp.brute_force = ctx->brute_force;
    rp.root_light_samples = (uint32_t)(int)sqrtf((float)ctx->light_samples);
    rp.samples_per_path = samplesPerSegment(ctx, rp.root_light_samples);
    rp.group_hits = ctx->mixed_hits && ctx->group_hits ? 1 : 0;
    // rays marked "cannot hit the large triangles": only when both hierarchies keep the same large triangles apart
    rp.n_big_planes = 0;
    std::memset(rp.big_plane, 0, sizeof(rp.big_plane));
    if (!BIG_FIRST_OFF && ctx->mark_rays && ctx->bvh.big_count != 0u && (!ctx->two_trees || ctx->bvh_any.big_count == ctx->bvh.big_count)) {
        for (uint32_t k = 0; k < ctx->bvh.big_count && k < 3u; ++k) {
            const int32_t g = ctx->bvh.big_prim[k];
            if (g < 0 || (size_t)g >= ctx->tris.size()) {  // (cannot happen: then nothing is marked)
                rp.n_big_planes = 0;
                break;
            }
            const PtcTriangle& t = ctx->tris[(size_t)g];
            const float pl[6] = {t.normal.x, t.normal.y, t.normal.z, t.v1.x, t.v1.y, t.v1.z};
            bool seen = false;
            for (uint32_t j = 0; j < rp.n_big_planes; ++j) seen = seen || std::memcmp(rp.big_plane[j], pl, sizeof(pl)) == 0;
            if (!seen) std::memcpy(rp.big_plane[rp.n_big_planes++], pl, sizeof(pl));
        }
    }
    rp.tune_refill = ctx->tune_refill;
    rp.tune_mode = ctx->tune_mode;
    // shadow rays end at very different times (any hit stops them): refilling waves pay off
    // once the tree is deep enough for that to matter
    rp.tune_shadow = ctx->tune_shadow == 3 ? 1u : ctx->tune_shadow;
    rp.tune_chunk = ctx->tune_chunk;
    return PTC_OK;
}

int uploadLights(ptc_context* ctx) {
    CHK(upload(ctx->d_plights, ctx->plights, ctx->stream));
    CHK(upload(ctx->d_dlights, ctx->dlights, ctx->stream));
    CHK(upload(ctx->d_qlights, ctx->qlights, ctx->stream));
    return PTC_OK;
}

#define TWO_TREES_MAX_PRIMS (1u << 19)  /* builder 3 above this size: one PLOC hierarchy */
#define N_COUNTERS 40  /* [0..3] nodes / primitives (extension, shadow), [4] shadow rays, [5..6] maxima, [7] shadow rays found occluded (deferred kernel), [8..14] / [16..22] schedule statistics of the deferred kernel, [24..29] / [32..37] how many lanes of a wave visit the same node */
#define N_QCOUNT 20
#ifndef DENSE_HIT_RATIO
#define DENSE_HIT_RATIO 0.95f  /* expected hits per live path from which on the hit compaction is skipped */
#endif

// Trace every bounce of the count0 paths k_generate left in the path state.  Per bounce:
// extend, hit compaction, surface (light samples + shadow rays, next direction,
// survivors -> next queue), shadow, gather (radiance of the segments that sampled the lights).  Queue
// lengths stay on the device; the grids are sized by the bounce's path count; the next
// queue's length is the one word read back.
int runBounces(ptc_context* ctx, const RenderParams& rp_frame, const DevScene& sc, uint32_t count0, uint32_t sample_begin, uint32_t max_iters = 0xFFFFFFFFu,
               uint32_t* survivors = nullptr, bool first_is_camera = true) {
    RenderParams rp = rp_frame;
    uint32_t count = count0;
    uint32_t bounce = 0;
    const uint32_t S = rp.samples_per_path;
    const DevScene sc_any = deviceSceneAny(ctx);
    // k_gather only adds to L; hit compaction does too when there is a background colour
    const bool overlap = ctx->overlap_gather && !(rp.bg[0] != 0.f || rp.bg[1] != 0.f || rp.bg[2] != 0.f);
    bool gather_pending = false;
    while (count > 0 && bounce < max_iters) {
        // camera segments of a batch whose sums k_generate did not seed: this bounce writes them
        rp.init_L = (bounce == 0 && first_is_camera && rp.seeding != 0) ? 1u : 0u;
        rp.camera_rays = (bounce == 0 && first_is_camera) ? 1u : 0u;
        uint32_t* grp = ctx->qcount + 8 * (bounce & 1u);  // {hit count, next count, shadow-ray count, extend work, shadow work, gather count}
        const bool timed = ctx->profile && count >= ctx->profile_min_items;
        CHK(hipMemsetAsync(grp, 0, 6 * sizeof(uint32_t), ctx->stream));
        if (ctx->sort_rays && bounce >= 1u && count >= 65536u && !ctx->no_alloc) {
            // scattered rays, ordered by origin cell and direction octant.  The four queue / hit arrays are free at
            // this point of a bounce (the last gather, which reads the gather queue, has to be through first).
            if (gather_pending) {
                CHK(hipStreamWaitEvent(ctx->stream, ctx->ev_gather, 0));
                gather_pending = false;
            }
            if (ctx->sort_temp_capacity < ctx->ps.capacity) {
                if (ctx->sort_temp) (void)hipFree(ctx->sort_temp);
                ctx->sort_temp = nullptr;
                ctx->sort_temp_capacity = 0;
                ctx->sort_temp_bytes = ray_sort_temp_bytes(ctx->ps.capacity);
                CHK(hipMalloc(&ctx->sort_temp, std::max<size_t>(ctx->sort_temp_bytes, 1)));
                ctx->sort_temp_capacity = ctx->ps.capacity;
            }
            Timed t(ctx, 3);
            CHK(ray_sort(ctx->stream, ctx->ps, count, ctx->bvh.sort_lo, ctx->bvh.sort_hi, ctx->hitq, (uint32_t*)ctx->ps.hit_t, ctx->gatherq, (uint32_t*)ctx->ps.hit_ref,
                         ctx->sort_temp, ctx->sort_temp_bytes));
            std::swap(ctx->ps.o_seed, ctx->ps.o_seed_next);
            std::swap(ctx->ps.d_depth, ctx->ps.d_depth_next);
            std::swap(ctx->ps.thr, ctx->ps.thr_next);
        }
        {
            Timed t(ctx, 0, nullptr, timed);
            wf_extend(ctx->stream, rp, sc, ctx->ps, count, grp + 3, ctx->resident_waves, bounce == 0 && first_is_camera, ctx->count_trav != 0, ctx->counters);
        }
        ctx->stats.extension_rays += count;
        ctx->stats.extend_launches++;
        if (timed) {
            ctx->stats.extend_timed_rays += count;
            ctx->stats.extend_timed_launches++;
        }
        // Nearly every path of this bounce is expected to hit (the ratio of the bounce before it, or of the last batch's
        // camera rays): no compaction pass, the surface kernel looks at all live paths and ends the misses itself.
        // (the ratio this bounce had in the last batch: the camera rays of the dragon nearly all hit, the rays they spawn do not)
        const bool camera = bounce == 0 && first_is_camera;
        const uint32_t slot = camera ? 0u : std::min<uint32_t>(bounce + 1u, 15u);
        const float expect = ctx->hit_ratio[slot];
        const bool dense = ctx->dense_surface && !rp.group_hits && !rp.brute_force && wf_surface_can_skip_queue(rp) && expect >= DENSE_HIT_RATIO;
        if (!dense) {
            Timed t(ctx, 3, nullptr, timed);
            wf_compact_hits(ctx->stream, rp, sc, ctx->ps, count, ctx->hitq, grp);
        }
        if (gather_pending) {  // the last bounce's gather reads what this surface kernel overwrites
            CHK(hipStreamWaitEvent(ctx->stream, ctx->ev_gather, 0));
            gather_pending = false;
        }
        {
            Timed t(ctx, 2, nullptr, timed);
            wf_surface(ctx->stream, rp, sc, ctx->ps, dense ? nullptr : ctx->hitq, grp, count, ctx->npix_pad, sample_begin, ctx->segctr, grp + 1, ctx->gatherq,
                       grp + 5, grp + 2, bounce == 0 && first_is_camera, ctx->hitq /* unused as a queue then */);
        }
        if (S) {
            {
                Timed t(ctx, 1, nullptr, timed);
                wf_shadow(ctx->stream, rp, sc_any, ctx->ps, grp + 2, (uint64_t)count * S, grp + 4, ctx->resident_waves, ctx->count_trav != 0, ctx->counters, ctx->counters + 4);
            }
            if (overlap) {
                CHK(hipEventRecord(ctx->ev_shadow, ctx->stream));
                CHK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_shadow, 0));
                {
                    Timed t(ctx, 4, ctx->aux_stream, timed);
                    wf_gather(ctx->aux_stream, rp, sc, ctx->ps, ctx->gatherq, grp + 5, count);
                }
                CHK(hipEventRecord(ctx->ev_gather, ctx->aux_stream));
                gather_pending = true;
            } else {
                Timed t(ctx, 4, nullptr, timed);
                wf_gather(ctx->stream, rp, sc, ctx->ps, ctx->gatherq, grp + 5, count);
            }
            ctx->stats.connect_launches++;
        }
        ctx->stats.shade_launches++;
        CHK(hipMemcpyAsync(ctx->h_count, grp, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));  // {hits, survivors}
        CHK(hipStreamSynchronize(ctx->stream));
        CHK(hipGetLastError());
        ctx->stats.surface_hits += ctx->h_count[0];
        if (max_iters == 0xFFFFFFFFu) ctx->hit_ratio[slot] = (float)ctx->h_count[0] / (float)count;
        if (timed) {
            ctx->stats.surface_timed_hits += ctx->h_count[0];
            ctx->stats.surface_timed_launches++;
        }
        count = ctx->h_count[1];
        // the survivors' packed state becomes the current one
        std::swap(ctx->ps.o_seed, ctx->ps.o_seed_next);
        std::swap(ctx->ps.d_depth, ctx->ps.d_depth_next);
        std::swap(ctx->ps.thr, ctx->ps.thr_next);
        ++bounce;
    }
    if (gather_pending) CHK(hipStreamWaitEvent(ctx->stream, ctx->ev_gather, 0));
    if (survivors) *survivors = count;
    return PTC_OK;
}

// Samples [j0, j1) of every pixel, `per_batch` samples per wavefront batch, into ctx->acc.  The render
// targets and the path arrays are in place; nothing is allocated here.
int renderBatches(ptc_context* ctx, RenderParams& rp, int32_t j0, int32_t j1, uint64_t per_batch) {
    const DevScene sc = deviceScene(ctx);
    for (int32_t ja = j0; ja < j1; ja += (int32_t)per_batch) {
        const uint32_t ns = (uint32_t)std::min<int64_t>((int64_t)per_batch, j1 - ja);
        rp.batch_samples = ns;
        rp.inv_batch_samples = 1.0 / (double)ns;
        const uint32_t n_paths = ns * ctx->npix_pad;
        const int seedL = rp.seeding == 0 ? 1 : 0;  // reference order: result += radiance per segment
        {
            Timed t(ctx, 3);
            wf_generate(ctx->stream, rp, ctx->ps, ctx->npix_pad, n_paths, (uint32_t)ja, ctx->acc, ctx->segctr, ctx->pixseed, seedL);
        }
        const uint32_t count0 = ns * rp.width * rp.height;  // the padding pixels' paths are not in the packed state
        ctx->stats.paths += count0;
        const int rc = runBounces(ctx, rp, sc, count0, (uint32_t)ja);
        if (rc) return rc;
        {
            Timed t(ctx, 3);
            wf_accumulate(ctx->stream, rp, ctx->ps, ctx->npix_pad, ns, ctx->acc, seedL);
        }
        if (ctx->profile) {
            CHK(hipStreamSynchronize(ctx->stream));
            collectSpans(ctx);
        }
    }
    CHK(hipGetLastError());
    return PTC_OK;
}

int checkRange(ptc_context* ctx, int32_t j0, int32_t j1) {
    if (j0 < 0 || j1 < j0) return fail(ctx, PTC_ERR_INVALID, "bad sample range");
    if ((uint32_t)j1 > (1u << 20)) return fail(ctx, PTC_ERR_INVALID, "at most 2^20 samples per pixel");
    return PTC_OK;
}

int renderRange(ptc_context* ctx, int frame_id, uint32_t w, uint32_t h, int32_t j0, int32_t j1, int clear) {
    RenderParams rp;
    int rc = makeParams(ctx, frame_id, w, h, rp);
    if (rc) return rc;
    rc = checkRange(ctx, j0, j1);
    if (rc) return rc;
    rc = ensureTargets(ctx, w, h);
    if (rc) return rc;
    const size_t npix = (size_t)w * h;
    if (clear) {
        CHK(hipMemsetAsync(ctx->acc, 0, npix * 3 * sizeof(float), ctx->stream));
        if (j0 == 0) {
            CHK(hipMemsetAsync(ctx->segctr, 0, (size_t)ctx->npix_pad * sizeof(uint32_t), ctx->stream));
            ctx->ref_next_sample = 0;
        }
    }
    if (rp.seeding == 0 && j1 > j0 && j0 != ctx->ref_next_sample)
        return fail(ctx, PTC_ERR_INVALID, "reference seeding depends on the pixel's earlier samples: ranges must start at 0 (with clear) or continue the previous range");
    const uint32_t S = rp.samples_per_path;
    // samples of every pixel per batch
    uint64_t per_batch = rp.seeding == 0 ? 1u : std::max<uint64_t>(1u, ctx->batch_paths / ctx->npix_pad);
    {
        // a batch may take half of what the device has free (counting what this context already holds for paths)
        const uint64_t per_path = bytesPerPath(S);
        size_t free_b = 0, total_b = 0;
        CHK(hipMemGetInfo(&free_b, &total_b));
        const uint64_t fit = ((uint64_t)free_b + ctx->path_pool_bytes) / 2u / per_path;
        per_batch = std::max<uint64_t>(1u, std::min<uint64_t>(per_batch, fit / ctx->npix_pad));
        // 32-bit indices into the per-sample arrays
        if (S) per_batch = std::max<uint64_t>(1u, std::min<uint64_t>(per_batch, 0xFFFFFF00ull / ((uint64_t)ctx->npix_pad * S)));
    }
    per_batch = std::min<uint64_t>(per_batch, (uint64_t)std::max(1, j1 - j0));
    per_batch = std::max<uint64_t>(1u, std::min<uint64_t>(per_batch, 0x3FFFFFF0ull / ctx->npix_pad));
    if ((uint64_t)ctx->npix_pad * per_batch > 0x3FFFFFF0ull) return fail(ctx, PTC_ERR_INVALID, "batch too large");  // path numbers keep their top two bits free (gather records)
    // Several contexts on one GPU (ranks or host threads sharing it) size their batches from the same free-memory
    // figure at the same moment: when the allocation fails for lack of memory, try again with half the batch.
    while ((rc = ensurePaths(ctx, (uint32_t)(ctx->npix_pad * per_batch), S)) == PTC_ERR_NOMEM && per_batch > 1u) per_batch = (per_batch + 1u) / 2u;
    if (rc) return rc;
    ctx->stats.batch_paths = std::max<uint64_t>(ctx->stats.batch_paths, ctx->npix_pad * per_batch);
    ctx->stats.path_state_bytes = ctx->path_pool_bytes;
    rc = renderBatches(ctx, rp, j0, j1, per_batch);
    if (rc) return rc;
    if (rp.seeding == 0) ctx->ref_next_sample = j1;
    return PTC_OK;
}

int enter(ptc_context* ctx) {
    if (!ctx) return PTC_ERR_INVALID;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, PTC_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return PTC_OK;
}
// An exception (std::bad_alloc from a host container, a thread that could not start ...) must not leave through the
// C ABI: every entry point is a function-try-block that ends here.
int caught(ptc_context* ctx, const char* what) noexcept {
    try {
        const std::string msg = std::string("host-side failure: ") + what;
        if (ctx) ctx->error = msg;
        else g_create_error = msg;
    } catch (...) {
    }
    return PTC_ERR_HOST;
}
#define ENTER()                  \
    do {                         \
        int rc_ = enter(ctx);    \
        if (rc_) return rc_;     \
    } while (0)

}  // namespace

extern "C" {

int ptc_create(int device_ordinal, ptc_context** out) try {
    if (!out) {
        g_create_error = "ptc_create: null output pointer";
        return PTC_ERR_INVALID;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_create_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return PTC_ERR_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        g_create_error = "ptc_create: device ordinal out of range";
        return PTC_ERR_INVALID;
    }
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return PTC_ERR_DEVICE;
    }
    ptc_context* ctx = new ptc_context;
    ctx->device = device_ordinal;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        (void)ptc_destroy(ctx);  // releases whatever exists so far
        return PTC_ERR_DEVICE;
    };
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    ctx->stream = ctx->own_stream;
    if ((e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_shadow, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_gather, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev_t0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev_t1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc((void**)&ctx->qcount, N_QCOUNT * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&ctx->counters, N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(ctx->counters, 0, N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipHostMalloc((void**)&ctx->h_count, 2 * sizeof(uint32_t))) != hipSuccess) return bail("hipHostMalloc", e);
    ctx->resident_waves = wf_resident_waves(device_ordinal);
    if (device_ordinal < kMaxDevices) {  // once per device, with that device current
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_ordinal]) {
            void* p = nullptr;
            if ((e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch))) != hipSuccess) return bail("hipGetSymbolAddress", e);
            g_scratch_ptr[device_ordinal] = (unsigned char*)p;
        }
    }
    *out = ctx;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(nullptr, e.what());
} catch (...) {
    return caught(nullptr, "unknown exception");
}

int ptc_destroy(ptc_context* ctx) {
    if (!ctx) return PTC_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    freeTargets(ctx);
    freePaths(ctx);
    bvh_free(ctx->bvh);
    bvh_free(ctx->bvh_any);
    for (void* p : {(void*)ctx->d_tris, (void*)ctx->d_spheres, (void*)ctx->d_materials, (void*)ctx->d_plights, (void*)ctx->d_dlights, (void*)ctx->d_qlights,
                    (void*)ctx->qcount, (void*)ctx->counters})
        if (p) (void)hipFree(p);
    if (ctx->sort_temp) (void)hipFree(ctx->sort_temp);
    if (ctx->h_count) (void)hipHostFree(ctx->h_count);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->ev_shadow) (void)hipEventDestroy(ctx->ev_shadow);
    if (ctx->ev_gather) (void)hipEventDestroy(ctx->ev_gather);
    if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
    if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
    delete ctx;
    return PTC_OK;
}

const char* ptc_last_error(const ptc_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int ptc_set_config(ptc_context* ctx, const PtcConfig* cfg) try {
    ENTER();
    if (!cfg) return fail(ctx, PTC_ERR_INVALID, "null config");
    ctx->cfg = *cfg;
    ctx->next_event_est = cfg->next_event_est;
    ctx->have_cfg = true;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_integrator(ptc_context* ctx, const char* name) try {
    ENTER();
    if (!name) return fail(ctx, PTC_ERR_INVALID, "null integrator name");
    for (int i = 0; i < 5; ++i)
        if (std::strcmp(name, kIntegrators[i]) == 0) {
            ctx->integrator = i;
            return PTC_OK;
        }
    return fail(ctx, PTC_ERR_INTEGRATOR, std::string("Doesn't support the integrator ") + name);
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_params(ptc_context* ctx, uint32_t light_samples, uint32_t light_stratify, uint32_t next_event_est, uint32_t sampling_method, int32_t spp) try {
    ENTER();
    if (spp < 0) return fail(ctx, PTC_ERR_INVALID, "samples_per_pixel must be >= 0");
    if (light_samples > (1u << 20)) return fail(ctx, PTC_ERR_INVALID, "light_samples too large");
    ctx->light_samples = light_samples;
    ctx->light_stratify = light_stratify;
    ctx->next_event_est = next_event_est;
    ctx->sampling_method = sampling_method;
    ctx->spp = spp;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_triangles(ptc_context* ctx, const PtcTriangle* tris, size_t n) try {
    ENTER();
    if (n && !tris) return fail(ctx, PTC_ERR_INVALID, "null triangle array");
    if (n > (1u << 27) - 1) return fail(ctx, PTC_ERR_INVALID, "too many triangles");
    ctx->tris.assign(tris, tris + n);
    ctx->built = false;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_spheres(ptc_context* ctx, const PtcSphere* sph, size_t n) try {
    ENTER();
    if (n && !sph) return fail(ctx, PTC_ERR_INVALID, "null sphere array");
    if (n > (1u << 27) - 1) return fail(ctx, PTC_ERR_INVALID, "too many spheres");
    ctx->spheres.assign(sph, sph + n);
    ctx->built = false;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_lights(ptc_context* ctx, const PtcPointLight* pl, size_t np, const PtcDirectionalLight* dl, size_t nd, const PtcQuadLight* ql, size_t nq) try {
    ENTER();
    if ((np && !pl) || (nd && !dl) || (nq && !ql)) return fail(ctx, PTC_ERR_INVALID, "null light array");
    if (np + nd > 4096 || nq > 4096) return fail(ctx, PTC_ERR_INVALID, "too many lights");
    ctx->plights.assign(pl, pl + np);
    ctx->dlights.assign(dl, dl + nd);
    ctx->qlights.assign(ql, ql + nq);
    return uploadLights(ctx);  // device copies are made here, so that no render call has to allocate
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_background(ptc_context* ctx, float r, float g, float b) try {
    ENTER();
    ctx->bg[0] = r;
    ctx->bg[1] = g;
    ctx->bg[2] = b;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_build(ptc_context* ctx) try {
    ENTER();
    CHK(hipStreamSynchronize(ctx->stream));
    const hipEvent_t e0 = ctx->ev_t0, e1 = ctx->ev_t1;  // owned by the context: no error path can leak them
    CHK(hipEventRecord(e0, ctx->stream));
    // de-duplicate the per-primitive material copies
    std::map<std::string, uint32_t> ids;
    std::vector<PtcMaterialValue> mats;
    ctx->non_diffuse = false;
    const PtcMaterialValue* last_mv = nullptr;  // a mesh's primitives come in runs of one material: compare with the previous one first
    uint32_t last_id = 0;
    auto matId = [&](const PtcMaterialValue& mv, int objType) {
        uint32_t id;
        if (last_mv && memcmp(last_mv, &mv, sizeof(mv)) == 0) {
            id = last_id;
        } else {
            std::string key((const char*)&mv, sizeof(mv));
            auto it = ids.find(key);
            if (it == ids.end()) {
                id = (uint32_t)mats.size();
                ids.emplace(key, id);
                mats.push_back(mv);
            } else {
                id = it->second;
            }
            last_mv = &mv;
            last_id = id;
        }
        if (mv.matType != PTC_DIFFUSE) ctx->non_diffuse = true;
        return id | (objType == PTC_LIGHT ? MAT_LIGHT_BIT : 0u);
    };
    std::vector<uint32_t> tmat(ctx->tris.size()), smat(ctx->spheres.size());
    for (size_t i = 0; i < ctx->tris.size(); ++i) tmat[i] = matId(ctx->tris[i].mv, ctx->tris[i].objType);
    // spheres always report objectType OBJECT (Sphere.cu:55)
    for (size_t i = 0; i < ctx->spheres.size(); ++i) smat[i] = matId(ctx->spheres[i].mv, PTC_OBJECT);
    {
        // shading classes present: (triangle | sphere) x (Phong | GGX); with more than one, the
        // hit queue is grouped by class so that the lanes of a wave run the same branches
        bool seen[4] = {false, false, false, false};
        for (uint32_t m : tmat) seen[mats[m & ~MAT_LIGHT_BIT].brdf_type == PTC_GGX ? 1 : 0] = true;
        for (uint32_t m : smat) seen[2 + (mats[m & ~MAT_LIGHT_BIT].brdf_type == PTC_GGX ? 1 : 0)] = true;
        ctx->mixed_hits = (int)seen[0] + seen[1] + seen[2] + seen[3] > 1;
    }
    CHK(upload(ctx->d_materials, mats, ctx->stream));
    ctx->n_materials = (uint32_t)mats.size();
    CHK(upload(ctx->d_tris, ctx->tris, ctx->stream));
    CHK(upload(ctx->d_spheres, ctx->spheres, ctx->stream));
    bvh_free(ctx->bvh);
    bvh_free(ctx->bvh_any);
    ctx->two_trees = false;
    BvhBuildInput in{};
    in.d_tris = ctx->d_tris;
    in.n_tris = (uint32_t)ctx->tris.size();
    in.d_spheres = ctx->d_spheres;
    in.n_spheres = (uint32_t)ctx->spheres.size();
    in.h_tri_mat = tmat.data();
    in.h_sph_mat = smat.data();
    // two hierarchies up to the size the sweep builder takes; clustering alone beyond that
    const bool two = ctx->bvh_builder == 3 && ctx->tris.size() + ctx->spheres.size() <= TWO_TREES_MAX_PRIMS;
    in.builder = ctx->bvh_builder == 3 ? (two ? 4 : 1) : ctx->bvh_builder;
    in.extra_extent = ctx->have_cfg ? std::max(fabsf(ctx->cfg.eye.x), std::max(fabsf(ctx->cfg.eye.y), fabsf(ctx->cfg.eye.z))) : 0.f;
    // A hierarchy the traversal stack cannot hold (clustering and sweep follow the geometry, and some geometry
    // makes chains) is built again as a radix tree over the Morton codes, whose depth is bounded by the key length.
    auto build = [&in](int builder, BvhBuildOutput& out, hipStream_t s) {
        BvhBuildInput mine = in;
        mine.builder = builder;
        std::string e = bvh_build(s, mine, out);
        if (!e.empty() && builder != 0 && e.find("too deep") != std::string::npos) {
            bvh_free(out);
            mine.builder = 0;
            e = bvh_build(s, mine, out);
        }
        return e;
    };
    std::string err;
    if (two) {
        // closest-hit rays: surface-area sweep; any-hit rays: clustering.  A build is a chain of some 1500 short launches and
        // a dozen read-backs of a few words, bound by launch and round-trip latency, not by the GPU: the two hierarchies are
        // built side by side, the second from a host thread of its own on the context's second stream (which starts behind
        // the uploads above).
        CHK(hipEventRecord(ctx->ev_shadow, ctx->stream));
        CHK(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_shadow, 0));
        std::string err_any;
        std::thread side([&] {
            try {
                if (hipSetDevice(ctx->device) != hipSuccess) err_any = "BVH build: hipSetDevice failed in the second build thread";
                else err_any = build(1, ctx->bvh_any, ctx->aux_stream);
            } catch (const std::exception& e) {
                err_any = std::string("BVH build: ") + e.what();
            }
        });
        try {
            err = build(in.builder, ctx->bvh, ctx->stream);
        } catch (...) {
            side.join();
            throw;
        }
        side.join();
        (void)hipStreamSynchronize(ctx->aux_stream);
        if (err.empty()) err = err_any;
        ctx->two_trees = err.empty();
    } else {
        err = build(in.builder, ctx->bvh, ctx->stream);
    }
    if (!err.empty()) {
        bvh_free(ctx->bvh);
        bvh_free(ctx->bvh_any);
        ctx->two_trees = false;
        return fail(ctx, PTC_ERR_DEVICE, err);
    }
    CHK(hipEventRecord(e1, ctx->stream));
    CHK(hipEventSynchronize(e1));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ctx->stats.build_ms = ms;
    ctx->stats.build_host_ms = ctx->bvh.host_ms + ctx->bvh_any.host_ms;
    ctx->stats.bvh_nodes = ctx->bvh.n_nodes;
    ctx->stats.bvh_max_depth = ctx->bvh.max_depth;
    for (float& r : ctx->hit_ratio) r = 0.f;  // a new scene: nothing is known about its rays
    ctx->built = true;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_render_range(ptc_context* ctx, int frame_id, uint32_t width, uint32_t height, int32_t spp_begin, int32_t spp_end, int clear) try {
    ENTER();
    return renderRange(ctx, frame_id, width, height, spp_begin, spp_end, clear);
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_render_range_static(ptc_context* ctx, int frame_id, uint32_t width, uint32_t height, int32_t spp_begin, int32_t spp_end, int clear,
                            void* sums_device, size_t sums_bytes) try {
    ENTER();
    RenderParams rp;
    int rc = makeParams(ctx, frame_id, width, height, rp);
    if (rc) return rc;
    rc = checkRange(ctx, spp_begin, spp_end);
    if (rc) return rc;
    if (width == 0 || height == 0) return fail(ctx, PTC_ERR_INVALID, "image size must be non-zero");
    const size_t npix = (size_t)width * height;
    if (!sums_device || sums_bytes < npix * 3 * sizeof(float)) return fail(ctx, PTC_ERR_INVALID, "sum buffer missing or too small for this image size");
    unsigned char* const scratch = ctx->device < kMaxDevices ? g_scratch_ptr[ctx->device] : nullptr;
    if (!scratch) return fail(ctx, PTC_ERR_UNSUPPORTED, "no static scratch on this device");
    if (rp.seeding == 0 && spp_end > spp_begin && !(spp_begin == 0 && clear))
        return fail(ctx, PTC_ERR_INVALID, "reference seeding: the static scratch keeps no history between calls, ranges must start at 0 with clear");
    // layout of the scratch array: per-pixel segment counters, per-pixel jitter seeds, per-path arrays
    const uint64_t npix_pad = paddedPixels(width, height);
    const uint32_t S = rp.samples_per_path;
    const size_t head = 2 * (((size_t)npix_pad * sizeof(uint32_t) + 255u) & ~(size_t)255u);
    PathState ps{};
    uint32_t *hq = nullptr, *gq = nullptr;
    uint64_t per_batch = rp.seeding == 0 ? 1u : std::max<uint64_t>(1u, std::min<uint64_t>(ctx->batch_paths / npix_pad, (uint64_t)std::max(1, spp_end - spp_begin)));
    const uint64_t room = PTC_STATIC_SCRATCH_BYTES > head ? (PTC_STATIC_SCRATCH_BYTES - head) / (bytesPerPath(S) + 1u) : 0u;  // paths, roughly
    per_batch = std::min<uint64_t>(per_batch, std::max<uint64_t>(1u, room / npix_pad));
    while (per_batch > 1u && (npix_pad * per_batch > 0x3FFFFFF0ull || head + carvePaths(nullptr, (uint32_t)(npix_pad * per_batch), S, ps, hq, gq) > PTC_STATIC_SCRATCH_BYTES)) --per_batch;
    if (npix_pad > 0xFFFFF000ull || (S && npix_pad * S > 0xFFFFFF00ull) ||
        head + carvePaths(nullptr, (uint32_t)npix_pad, S, ps, hq, gq) > PTC_STATIC_SCRATCH_BYTES)
        return fail(ctx, PTC_ERR_INVALID, "image too large for the static scratch (" + std::to_string(PTC_STATIC_SCRATCH_MIB) + " MiB): one sample of every pixel of the image, padded to 64x64 blocks, needs " +
                                              std::to_string(bytesPerPath(S) + 8u) + " bytes per pixel with " + std::to_string(S) + " light samples per segment");
    // one call at a time per device; released below, after this call's last kernel has finished
    if (g_scratch_busy[ctx->device].exchange(true, std::memory_order_acquire))
        return fail(ctx, PTC_ERR_BUSY, "the static scratch of this GPU is in use by another ptc_render_range_static call");
    // the context's own arrays step aside for the duration of the call
    struct Saved {
        PathState ps;
        uint32_t *hitq, *gatherq, *segctr, *pixseed, contrib_samples, width, height, npix_pad;
        float* acc;
        bool no_alloc;
    } const saved = {ctx->ps, ctx->hitq, ctx->gatherq, ctx->segctr, ctx->pixseed, ctx->contrib_samples, ctx->width, ctx->height, ctx->npix_pad, ctx->acc, ctx->no_alloc};
    ctx->segctr = (uint32_t*)scratch;
    ctx->pixseed = (uint32_t*)(scratch + head / 2);
    carvePaths((char*)scratch + head, (uint32_t)(npix_pad * per_batch), S, ctx->ps, ctx->hitq, ctx->gatherq);
    ctx->contrib_samples = S;
    ctx->npix_pad = (uint32_t)npix_pad;
    ctx->acc = (float*)sums_device;
    ctx->no_alloc = true;
    auto run = [&]() -> int {
        // the array keeps the last call's contents: what must start at zero is cleared here
        if (rp.seeding == 0) CHK(hipMemsetAsync(ctx->segctr, 0, (size_t)npix_pad * sizeof(uint32_t), ctx->stream));
        if (clear) CHK(hipMemsetAsync(ctx->acc, 0, npix * 3 * sizeof(float), ctx->stream));
        const int r = renderBatches(ctx, rp, spp_begin, spp_end, per_batch);
        if (r) return r;
        CHK(hipStreamSynchronize(ctx->stream));  // the scratch is free for the next call when this one returns
        return PTC_OK;
    };
    rc = run();
    if (rc) {  // kernels of the failed call may still be writing the scratch: wait for both streams before letting go of it
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipGetLastError();
    }
    g_scratch_busy[ctx->device].store(false, std::memory_order_release);
    ctx->ps = saved.ps;
    ctx->hitq = saved.hitq;
    ctx->gatherq = saved.gatherq;
    ctx->segctr = saved.segctr;
    ctx->pixseed = saved.pixseed;
    ctx->contrib_samples = saved.contrib_samples;
    ctx->width = saved.width;
    ctx->height = saved.height;
    ctx->npix_pad = saved.npix_pad;
    ctx->acc = saved.acc;
    ctx->no_alloc = saved.no_alloc;
    return rc;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_resolve(ptc_context* ctx, int frame_id, int32_t total_spp) try {
    ENTER();
    if (!ctx->width) return fail(ctx, PTC_ERR_INVALID, "nothing rendered yet");
    if (!ctx->have_cfg) return fail(ctx, PTC_ERR_INVALID, "ptc_set_config has not been called");
    wf_resolve(ctx->stream, ctx->acc, ctx->result, ctx->width * ctx->height, total_spp, ctx->cfg.gamma, frame_id);
    CHK(hipGetLastError());
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_render(ptc_context* ctx, int frame_id, uint32_t width, uint32_t height) try {
    ENTER();
    int rc = renderRange(ctx, frame_id, width, height, 0, ctx->spp, 1);
    if (rc) return rc;
    rc = ptc_resolve(ctx, frame_id, ctx->spp);
    if (rc) return rc;
    CHK(hipStreamSynchronize(ctx->stream));
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_primary_hits(ptc_context* ctx, uint32_t width, uint32_t height, PtcHitRecord* out) try {
    ENTER();
    if (!out) return fail(ctx, PTC_ERR_INVALID, "null output");
    RenderParams rp;
    const int saved = ctx->integrator;
    if (ctx->integrator < 0) ctx->integrator = INT_RAYTRACER;
    const bool nd = ctx->non_diffuse;
    ctx->non_diffuse = false;  // only intersection runs here
    int rc = makeParams(ctx, 1, width, height, rp);
    ctx->integrator = saved;
    ctx->non_diffuse = nd;
    if (rc) return rc;
    rp.seeding = 1;  // leaves the per-pixel segment counters untouched
    rc = ensureTargets(ctx, width, height);
    if (rc) return rc;
    rc = ensurePaths(ctx, ctx->npix_pad, 0);
    if (rc) return rc;
    const DevScene sc = deviceScene(ctx);
    CHK(hipMemsetAsync(ctx->qcount, 0, N_QCOUNT * sizeof(uint32_t), ctx->stream));
    CHK(hipMemsetAsync(ctx->ps.hit_ref, 0xFF, (size_t)ctx->npix_pad * sizeof(int32_t), ctx->stream));  // HIT_MISS
    wf_generate(ctx->stream, rp, ctx->ps, ctx->npix_pad, ctx->npix_pad, 0u, ctx->acc, ctx->segctr, ctx->pixseed, 0);
    const uint32_t n_rays = width * height;
    rp.camera_rays = 1u;
    wf_extend(ctx->stream, rp, sc, ctx->ps, n_rays, ctx->qcount + 3, ctx->resident_waves, true, false, ctx->counters);
    wf_hit_records(ctx->stream, rp, sc, ctx->ps, n_rays, ctx->d_hits);
    CHK(hipMemcpyAsync(out, ctx->d_hits, (size_t)width * height * sizeof(PtcHitRecord), hipMemcpyDeviceToHost, ctx->stream));
    CHK(hipStreamSynchronize(ctx->stream));
    CHK(hipGetLastError());
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_read_result(ptc_context* ctx, float* rgb_out) try {
    ENTER();
    if (!rgb_out || !ctx->width) return fail(ctx, PTC_ERR_INVALID, "nothing to read");
    CHK(hipMemcpyAsync(rgb_out, ctx->result, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    CHK(hipStreamSynchronize(ctx->stream));
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_read_sum(ptc_context* ctx, float* rgb_out) try {
    ENTER();
    if (!rgb_out || !ctx->width) return fail(ctx, PTC_ERR_INVALID, "nothing to read");
    CHK(hipMemcpyAsync(rgb_out, ctx->acc, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    CHK(hipStreamSynchronize(ctx->stream));
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_bind_sum_buffer(ptc_context* ctx, void* device_ptr, size_t bytes) try {
    ENTER();
    if (device_ptr && ctx->width && bytes < (size_t)ctx->width * ctx->height * 3 * sizeof(float))
        return fail(ctx, PTC_ERR_INVALID, "sum buffer too small for the current image");
    ctx->acc_ext = device_ptr;
    ctx->acc_ext_bytes = bytes;
    if (ctx->width) ctx->acc = device_ptr ? (float*)device_ptr : ctx->acc_own;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_stream(ptc_context* ctx, void* hip_stream) try {
    ENTER();
    CHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_synchronize(ptc_context* ctx) try {
    ENTER();
    CHK(hipStreamSynchronize(ctx->stream));
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_probe_segments(ptc_context* ctx, uint32_t n, const PtcSegmentProbe* in, PtcSegmentResult* out) try {
    ENTER();
    if (n == 0) return PTC_OK;
    if (!in || !out) return fail(ctx, PTC_ERR_INVALID, "null probe array");
    if (n > (1u << 24)) return fail(ctx, PTC_ERR_INVALID, "at most 2^24 probes");
    RenderParams rp;
    int rc = makeParams(ctx, 1, 64, 64, rp);  // the image size only enters the NEXT segment's seed, which is not reported
    if (rc) return rc;
    rp.seeding = 1;  // leaves the per-pixel segment counters alone
    rp.cull_zero = 0;
    rp.bounce_cap = 0xFFFFFFFFu;
    rp.n_big_planes = 0;  // (the depth word of a survivor is reported as it is stored)
    const uint32_t S = rp.samples_per_path;
    const uint32_t cap = (n + 4095u) & ~4095u;
    rc = ensureTargets(ctx, ctx->width ? ctx->width : 64, ctx->height ? ctx->height : 64);  // segctr / acc exist
    if (rc) return rc;
    rc = ensurePaths(ctx, cap, S);
    if (rc) return rc;
    std::vector<float4> o(n), d(n), t(n), L(n, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<uint32_t> pid(n);
    auto bits = [](uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    for (uint32_t i = 0; i < n; ++i) {
        o[i] = make_float4(in[i].o[0], in[i].o[1], in[i].o[2], bits(in[i].seed));
        d[i] = make_float4(in[i].d[0], in[i].d[1], in[i].d[2], bits(in[i].depth));
        t[i] = make_float4(in[i].thr[0], in[i].thr[1], in[i].thr[2], bits(i));  // path number = probe number
        pid[i] = i;
    }
    hipStream_t s = ctx->stream;
    CHK(hipMemcpyAsync(ctx->ps.o_seed, o.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
    CHK(hipMemcpyAsync(ctx->ps.d_depth, d.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
    CHK(hipMemcpyAsync(ctx->ps.thr, t.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
    CHK(hipMemcpyAsync(ctx->ps.pid, pid.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    CHK(hipMemcpyAsync(ctx->ps.L, L.data(), n * sizeof(float4), hipMemcpyHostToDevice, s));
    CHK(hipStreamSynchronize(s));
    const DevScene sc = deviceScene(ctx);
    DeviceBuffer tmp_guard;  // freed on every return path
    CHK(hipMalloc(&tmp_guard.p, (size_t)n * 8));
    void* const tmp = tmp_guard.p;
    uint32_t alive_n = 0;
    const PtcStats saved = ctx->stats;
    rc = runBounces(ctx, rp, sc, n, 0u, 1u, &alive_n, false);
    ctx->stats = saved;  // probes are not part of the render statistics
    if (rc) return rc;
    // hit_t / hit_ref are indexed by the position of the segment, which is the probe number here
    wf_probe_hits(s, sc, ctx->ps, n, (int32_t*)tmp, (float*)tmp + n);
    std::vector<int32_t> prim(n);
    std::vector<float> ht(n);
    CHK(hipMemcpyAsync(prim.data(), tmp, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    CHK(hipMemcpyAsync(ht.data(), (float*)tmp + n, n * sizeof(float), hipMemcpyDeviceToHost, s));
    CHK(hipMemcpyAsync(L.data(), ctx->ps.L, n * sizeof(float4), hipMemcpyDeviceToHost, s));
    if (alive_n) {  // runBounces swapped the arrays: the survivors are the current state
        CHK(hipMemcpyAsync(o.data(), ctx->ps.o_seed, alive_n * sizeof(float4), hipMemcpyDeviceToHost, s));
        CHK(hipMemcpyAsync(d.data(), ctx->ps.d_depth, alive_n * sizeof(float4), hipMemcpyDeviceToHost, s));
        CHK(hipMemcpyAsync(t.data(), ctx->ps.thr, alive_n * sizeof(float4), hipMemcpyDeviceToHost, s));
    }
    CHK(hipStreamSynchronize(s));
    CHK(hipGetLastError());
    for (uint32_t i = 0; i < n; ++i) {
        std::memset(&out[i], 0, sizeof(out[i]));
        out[i].radiance[0] = L[i].x;
        out[i].radiance[1] = L[i].y;
        out[i].radiance[2] = L[i].z;
        out[i].prim = prim[i];
        out[i].t = ht[i];
    }
    for (uint32_t k = 0; k < alive_n; ++k) {
        uint32_t path = 0;
        std::memcpy(&path, &t[k].w, 4);
        if (path >= n) return fail(ctx, PTC_ERR_DEVICE, "probe: survivor with a bad path number");
        PtcSegmentResult& r = out[path];
        r.alive = 1;
        r.o[0] = o[k].x; r.o[1] = o[k].y; r.o[2] = o[k].z;
        r.d[0] = d[k].x; r.d[1] = d[k].y; r.d[2] = d[k].z;
        r.thr[0] = t[k].x; r.thr[1] = t[k].y; r.thr[2] = t[k].z;
        std::memcpy(&r.depth, &d[k].w, 4);
    }
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_eval_shading(ptc_context* ctx, int fn, uint32_t n, const float* in16, float* out4) try {
    ENTER();
    if (n == 0) return PTC_OK;
    if (!in16 || !out4 || fn < 0 || fn > 4) return fail(ctx, PTC_ERR_INVALID, "bad argument");
    float *d_in = nullptr, *d_out = nullptr;
    CHK(hipMalloc((void**)&d_in, (size_t)n * 16 * sizeof(float)));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n * 4 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in16, (size_t)n * 16 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        wf_eval_shading(ctx->stream, fn, n, d_in, d_out);
        e = hipMemcpyAsync(out4, d_out, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, PTC_ERR_DEVICE, std::string("ptc_eval_shading: ") + hipGetErrorString(e));
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_set_option(ptc_context* ctx, int option, int64_t value) try {
    ENTER();
    switch (option) {
        case PTC_OPT_SEMANTICS:
            if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "semantics must be 0 or 1");
            ctx->semantics = (int)value;
            break;
        case PTC_OPT_SEEDING:
            if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "seeding must be 0 or 1");
            ctx->seeding = (int)value;
            break;
        case PTC_OPT_BOUNCE_CAP:
            if (value < 1 || value > 4096) return fail(ctx, PTC_ERR_INVALID, "bounce cap must be 1..4096");
            ctx->bounce_cap = (uint32_t)value;
            break;
        case PTC_OPT_CULL_ZERO: ctx->cull_zero = value != 0; break;
        case PTC_OPT_BATCH_PATHS:
            if (value < 1) return fail(ctx, PTC_ERR_INVALID, "batch size must be positive");
            ctx->batch_paths = (uint64_t)value;
            break;
        case PTC_OPT_PROFILE:
            if (value < 0 || value > 0x7FFFFFFF) return fail(ctx, PTC_ERR_INVALID, "bad profile option");
            ctx->profile = value != 0;
            ctx->profile_min_items = value > 1 ? (uint32_t)value : 0u;
            break;
        case PTC_OPT_COUNT_TRAVERSAL: ctx->count_trav = value != 0; break;
        case PTC_OPT_BRUTE_FORCE: ctx->brute_force = value != 0; break;
        case PTC_OPT_GROUP_HITS: ctx->group_hits = value != 0; break;
        case PTC_OPT_OVERLAP_GATHER: ctx->overlap_gather = value != 0; break;
        case PTC_OPT_SORT_RAYS: ctx->sort_rays = value != 0; break;
        case PTC_OPT_DENSE_SURFACE: ctx->dense_surface = value != 0; break;
        case PTC_OPT_MARK_RAYS: ctx->mark_rays = value != 0; break;
        case PTC_OPT_BVH_BUILDER:
            if (value < 0 || value > 4) return fail(ctx, PTC_ERR_INVALID, "BVH builder must be 0 (LBVH), 1 (PLOC), 2 (surface-area sweep on the host), 4 (the same sweep on the GPU) or 3 (4 for closest-hit rays, 1 for any-hit rays)");
            ctx->bvh_builder = (int)value;
            ctx->built = false;
            break;
        case PTC_OPT_TUNE_TRAVERSAL:
            ctx->tune_refill = std::max<uint32_t>(1u, std::min<uint32_t>(64u, (uint32_t)(value & 0xFF)));
            ctx->tune_mode = (uint32_t)((value >> 8) & 0xF);
            ctx->tune_shadow = (uint32_t)((value >> 12) & 0xF);
            ctx->tune_chunk = (uint32_t)((value >> 16) & 0xFFFF) / 64u * 64u;
            break;
        default: return fail(ctx, PTC_ERR_INVALID, "unknown option");
    }
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_collapse_wide(const void* binary_nodes, uint32_t n_nodes, void* wide_nodes_out, uint32_t capacity, uint32_t* n_wide_out, uint32_t* depth_out) try {
    if (!binary_nodes || !wide_nodes_out || !n_wide_out || !depth_out || n_nodes == 0) return PTC_ERR_INVALID;
    const int rc = bvh_collapse_wide_host((const BvhNode*)binary_nodes, n_nodes, (WideNode*)wide_nodes_out, capacity, n_wide_out, depth_out);
    return rc == 0 ? PTC_OK : rc == 1 ? PTC_ERR_UNSUPPORTED : PTC_ERR_INVALID;  // 2 = capacity, 3 = malformed nodes
} catch (const std::exception& e) {
    return caught(nullptr, e.what());
} catch (...) {
    return caught(nullptr, "unknown exception");
}

int ptc_debug_counters(ptc_context* ctx, uint64_t* out, uint32_t n) try {
    ENTER();
    if (!out || n > N_COUNTERS) return fail(ctx, PTC_ERR_INVALID, "bad argument");
    CHK(hipStreamSynchronize(ctx->stream));
    CHK(hipMemcpy(out, ctx->counters, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_debug_hierarchy(ptc_context* ctx, int which, void* binary_out, uint32_t binary_capacity, uint32_t* n_binary, void* wide_out, uint32_t wide_capacity,
                        uint32_t* n_wide) try {
    ENTER();
    if (!ctx->built) return fail(ctx, PTC_ERR_NOT_BUILT, "ptc_build has not been called since the last geometry upload");
    if ((which != 0 && which != 1) || (which == 1 && !ctx->two_trees)) return fail(ctx, PTC_ERR_INVALID, "no such hierarchy");
    const BvhBuildOutput& b = which ? ctx->bvh_any : ctx->bvh;
    if (n_binary) *n_binary = b.n_nodes;
    if (n_wide) *n_wide = b.wnodes ? b.n_wnodes : 0u;
    if ((binary_out && binary_capacity < b.n_nodes) || (wide_out && b.wnodes && wide_capacity < b.n_wnodes)) return fail(ctx, PTC_ERR_INVALID, "output too small");
    CHK(hipStreamSynchronize(ctx->stream));
    if (binary_out && b.n_nodes) CHK(hipMemcpy(binary_out, b.nodes, (size_t)b.n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
    if (wide_out && b.wnodes && b.n_wnodes) CHK(hipMemcpy(wide_out, b.wnodes, (size_t)b.n_wnodes * sizeof(WideNode), hipMemcpyDeviceToHost));
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_get_stats(ptc_context* ctx, PtcStats* out) try {
    ENTER();
    if (!out) return fail(ctx, PTC_ERR_INVALID, "null output");
    CHK(hipStreamSynchronize(ctx->stream));
    collectSpans(ctx);
    std::vector<unsigned long long> h(N_COUNTERS);
    CHK(hipMemcpy(h.data(), ctx->counters, N_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    ctx->stats.nodes_visited = h[0];
    ctx->stats.prims_tested = h[1];
    ctx->stats.shadow_nodes_visited = h[2];
    ctx->stats.shadow_prims_tested = h[3];
    ctx->stats.shadow_rays = h[4];
    *out = ctx->stats;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

int ptc_reset_stats(ptc_context* ctx) try {
    ENTER();
    CHK(hipStreamSynchronize(ctx->stream));
    collectSpans(ctx);
    CHK(hipMemset(ctx->counters, 0, N_COUNTERS * sizeof(unsigned long long)));
    const double build_ms = ctx->stats.build_ms, build_host_ms = ctx->stats.build_host_ms;
    const uint32_t nodes = ctx->stats.bvh_nodes, depth = ctx->stats.bvh_max_depth;
    const uint64_t pool = ctx->stats.path_state_bytes, bpaths = ctx->stats.batch_paths;
    ctx->stats = PtcStats{};
    ctx->stats.path_state_bytes = pool;
    ctx->stats.batch_paths = bpaths;
    ctx->stats.build_ms = build_ms;
    ctx->stats.build_host_ms = build_host_ms;
    ctx->stats.bvh_nodes = nodes;
    ctx->stats.bvh_max_depth = depth;
    return PTC_OK;
} catch (const std::exception& e) {
    return caught(ctx, e.what());
} catch (...) {
    return caught(ctx, "unknown exception");
}

}  // extern "C"
