/*
 * ptcore.h -- C ABI of the MI355X (gfx950) path-tracing core.
 *
 * This is the drop-in boundary for the per-pixel hot path of the reference
 * renderer: everything OptiXRenderer/Renderer.cpp does to its OptiX context
 * (create, upload Config / Triangle[] / Sphere[] / lights, pick the integrator,
 * build the acceleration structure, launch, map the float3 result) has one
 * entry point here.  All signatures are plain C: pointers, sizes and PODs.
 *
 * Conventions
 *   - every call returns PTC_OK (0) or a negative PTC_ERR_* code;
 *     ptc_last_error(ctx) returns a NUL-terminated message owned by the context
 *     (ptc_last_error(NULL) returns the message of the last failed ptc_create);
 *   - host arrays passed in are copied during the call, the caller keeps
 *     ownership; output buffers are allocated by the caller;
 *   - one context = one GPU = one caller thread at a time.  Multi-GPU runs use
 *     one context per device (one per process/rank, or one per host thread);
 *   - the POD layouts below are the wire format of the reference's own structs
 *     (OptiXRenderer/Geometries.h, Light.h, Config.h) and are size/offset
 *     checked at compile time in C++ translation units.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ PODs -- */

typedef struct PtcFloat3 { float x, y, z; } PtcFloat3;

/* 8-byte aligned like CUDA's float2, so that PtcConfig has the same size (88)
 * and offsets as the reference's Config under the OptiX/CUDA headers. */
#if defined(__cplusplus)
typedef struct alignas(8) PtcFloat2 { float x, y; } PtcFloat2;
#else
typedef struct PtcFloat2 { _Alignas(8) float x; float y; } PtcFloat2;
#endif

/* Geometries.h:10-14 */
enum { PTC_OBJECT = 0, PTC_LIGHT = 1 };                     /* objectType   */
enum { PTC_MOD_PHONG = 0, PTC_GGX = 1 };                    /* brdfType     */
enum { PTC_DIFFUSE = 0, PTC_GLASS = 1, PTC_VOLUMETRIC = 2 };/* materialType */
/* Config.h:5-7 */
enum { PTC_HEMISPHERE_SAMPLING = 0, PTC_COSINE_SAMPLING = 1, PTC_BRDF_SAMPLING = 2 };
enum { PTC_NEE_OFF = 0, PTC_NEE_ON = 1, PTC_NEE_MIS = 2 };

/* Geometries.h:16-24 -- 84 bytes */
typedef struct PtcMaterialValue {
    PtcFloat3 ambient, diffuse, specular, emission;
    float     shininess, roughness;
    int32_t   brdf_type;
    float     ior;
    int32_t   matType;
    float     sigma_a, sigma_s, sigma_t, g;
} PtcMaterialValue;

/* Geometries.h:26-31 -- 136 bytes */
typedef struct PtcTriangle {
    PtcFloat3        v1, v2, v3, normal;
    PtcMaterialValue mv;
    int32_t          objType;
} PtcTriangle;

/* Geometries.h:33-38 -- 152 bytes; trans is row-major, translation in
 * elements [3], [7], [11] (Sphere.cu:67-69). */
typedef struct PtcSphere {
    float            trans[16];
    PtcMaterialValue mv;
    int32_t          objType;
} PtcSphere;

/* Light.h:10-26 */
typedef struct PtcPointLight       { PtcFloat3 color, attenuation, location; } PtcPointLight;
typedef struct PtcDirectionalLight { PtcFloat3 color, direction; }             PtcDirectionalLight;
typedef struct PtcQuadLight        { PtcTriangle tri1, tri2; PtcFloat3 color; } PtcQuadLight;

/* Config.h:9-36 -- 88 bytes */
typedef struct PtcConfig {
    PtcFloat3 w, u, v, eye;
    PtcFloat2 hSize, tanHFov;
    float     gamma;
    uint32_t  maxDepth;
    uint32_t  next_event_est;
    float     epsilon;
    uint32_t  russian_roul;
} PtcConfig;

/* Primary-hit record used by the parity checks: the closest hit of the
 * pixel-centre ray (sample 0) of each pixel.  prim is the global primitive
 * index (triangles 0..T-1 in upload order, then spheres T..T+S-1), -1 = miss.
 * u,v are the barycentric weights of Triangle.cu:43-44 (u weighs v3, v weighs
 * v2); 0 for spheres. */
typedef struct PtcHitRecord { int32_t prim; float t, u, v; } PtcHitRecord;

/* Counters and timings of the work done since the last ptc_reset_stats. */
typedef struct PtcStats {
    uint64_t paths;            /* camera paths started                         */
    uint64_t extension_rays;   /* radiance segments traced (paths x bounces)   */
    uint64_t shadow_rays;      /* shadow rays traced                           */
    uint64_t extend_launches, connect_launches, shade_launches; /* extension / shadow / surface launches */
    /* HIP-event time, only with PTC_OPT_PROFILE: extension-ray traversal, shadow-ray
     * traversal, surface + gather kernels, everything else */
    double   extend_ms, connect_ms, shade_ms, other_ms;
    uint64_t nodes_visited, prims_tested;               /* extension rays, only with PTC_OPT_COUNT_TRAVERSAL */
    uint64_t shadow_nodes_visited, shadow_prims_tested; /* shadow rays, same option                          */
    uint32_t bvh_nodes, bvh_max_depth;
    double   build_ms;
    uint64_t path_state_bytes; /* device memory of the per-path arrays of a batch    */
    uint64_t batch_paths;      /* camera paths of the largest batch rendered so far  */
    double   surface_ms, gather_ms; /* the two parts of shade_ms (PTC_OPT_PROFILE)   */
    uint64_t surface_hits;     /* segments that hit something = work items of the surface kernel */
    double   build_host_ms;    /* part of build_ms spent in host code (wall clock): 0 = the hierarchy was built on the GPU */
    /* the launches PTC_OPT_PROFILE timed (all of them with value 1, those of at least `value` work items otherwise):
     * extension rays and launches behind extend_ms, hits and launches behind surface_ms */
    uint64_t extend_timed_rays, extend_timed_launches, surface_timed_hits, surface_timed_launches;
} PtcStats;

typedef struct ptc_context ptc_context;

/* ----------------------------------------------------------- error codes -- */
enum {
    PTC_OK = 0,
    PTC_ERR_INVALID      = -1,  /* bad argument / call order                   */
    PTC_ERR_INTEGRATOR   = -2,  /* unknown integrator name                     */
    PTC_ERR_DEVICE       = -3,  /* HIP error (message has the HIP string)      */
    PTC_ERR_UNSUPPORTED  = -4,  /* scene uses a feature outside the built path */
    PTC_ERR_NOT_BUILT    = -5,  /* render before ptc_build                     */
    PTC_ERR_HOST         = -6,  /* host-side failure (out of memory, a thread
                                   that could not start): no exception leaves
                                   the library                                 */
    PTC_ERR_NOMEM        = -7,  /* the device is out of memory                 */
    PTC_ERR_BUSY         = -8   /* ptc_render_range_static: another call is
                                   using this GPU's static scratch             */
};

/* ---------------------------------------------------------------- options -- */
enum {
    /* 0 = literal: what the committed RayTracer.cu computes (computePolarAngles,
     *     genCoordFrame, swapValue, samplePhaseHG take their outputs by value,
     *     RayTracer.cu:232,296,302,337, so the callers' theta/phi/u/v stay 0).
     * 1 = intended: the same code with those four outputs taken by reference. */
    PTC_OPT_SEMANTICS = 1,
    /* 0 = reference seeding: per-segment seed tea<16>(index*frameID, i++) with i
     *     running over all segments of all earlier samples of the pixel
     *     (PinholeCamera.cu:34,56); samples of a pixel are traced one at a time.
     * 1 = counter seeding: seed(pixel,sample j,bounce b) =
     *     tea<16>(index*frameID, (j<<12)|b); independent of how samples are
     *     partitioned over batches or GPUs.  Identical to 0 for sample 0. */
    PTC_OPT_SEEDING = 2,
    /* Upper bound on segments per path, applied as min(Config.maxDepth, cap);
     * 1..4096.  `maxdepth -1` parses to 0xFFFFFFFF (SceneLoader.cpp:98), which
     * the reference would iterate literally. Default 1024. */
    PTC_OPT_BOUNCE_CAP = 3,
    /* 1 = stop a path once its throughput is exactly (0,0,0) (every later
     * contribution is radiance*throughput, RayTracer.cu:918). Default 1. Takes
     * effect with counter seeding only: under reference seeding the segments of a
     * dead path still advance the pixel's seed counter (PinholeCamera.cu:56) and
     * with it the seeds of every later sample, so they are traced. */
    PTC_OPT_CULL_ZERO = 4,
    /* Maximum camera paths in flight per wavefront batch (rounded to whole samples over
     * the image, and limited to what fits in half of the free device memory at about
     * 152 + 44 x light samples bytes per path). Default 536870912 (about 140 GB with two
     * light samples per segment): fewer, larger launches per frame.  A batch holds fewer
     * than 2^30 paths whatever is asked for. */
    PTC_OPT_BATCH_PATHS = 5,
    /* 1 = bracket every kernel with HIP events and sum the times per kernel class; N > 1 = only the
     * kernels of bounces with at least N live paths (the event records between the sub-millisecond
     * launches of a long tail of bounces cost more than the kernels they time).  PtcStats says how
     * many rays / hits and launches the times cover. */
    PTC_OPT_PROFILE = 6,
    /* 1 = count BVH nodes visited / primitives tested per ray (slower). */
    PTC_OPT_COUNT_TRAVERSAL = 7,
    /* 0 = BVH traversal, 1 = brute force over all primitives (parity aid). */
    PTC_OPT_BRUTE_FORCE = 8,
    /* Traversal scheduling knobs for experiments: bits 0..7 idle lanes of a wave that
     * trigger a refill (1..64); bits 8..11 extension rays, bits 12..15 shadow rays:
     * 0 = one ray per thread, 1 = refilling waves + while-while rounds, 2 = refilling
     * waves + single steps, 3 = 0 for camera rays and 1 for later bounces (shadow: 1),
     * 4 = refilling waves with deferred primitive tests (leaf primitives go to a
     * per-wave queue in LDS and are tested 64 at a time) over the two-child nodes,
     * 6 = the same over the four-wide nodes with 8-bit child boxes, 5 = automatic: 0 for
     * hierarchies of at most 6 (shadow rays: 8) nodes, 1 up to 64 nodes, 6 otherwise;
     * bits 16..31 queue entries a wave reserves per atomic (multiple of 64, 0 = auto).
     * Default 0x5518. */
    PTC_OPT_TUNE_TRAVERSAL = 9,
    /* Hierarchy builder used by the next ptc_build (the reference's is the device builder "Trbvh",
     * Renderer.cpp:203-213).  0 = LBVH (Karras radix tree over Morton codes), 1 = PLOC (locally-ordered
     * clustering over Morton order), 4 = top-down surface-area sweep (every node split where the sweep over
     * its primitives sorted by centroid, best of the three axes, is cheapest), all nodes of a level at once
     * with segmented scans -- each followed by tree rotations, leaf collapse, node emission and the collapse
     * into four-wide nodes, all on the GPU; 2 = the same sweep in host code (the reference implementation
     * the GPU sweep is tested against; ten times the build time); 3 (default) = two hierarchies, the
     * sweep's (4) for the closest-hit rays and PLOC's for the any-hit rays (both build times, both memory
     * footprints; same images; scenes of more than 2^19 primitives get 1).  Scenes of at most 1024
     * primitives of a kind always get 2 (a handful of nodes: host code is quicker than the launches), and
     * no "large primitive" leaf: in a Cornell box every wall is as large as the scene. */
    PTC_OPT_BVH_BUILDER = 10,
    /* 1 = when a scene mixes shading classes (triangle / sphere hits, Phong / GGX materials) the hits of a
     * bounce are grouped by class before the surface kernel runs, so that the lanes of a wave take the same
     * branches; 0 (default) = plain path order.  Results do not depend on it.  (Grouping paid while the surface
     * kernel was bound by its arithmetic; now that it is bound by memory traffic on the small scenes, the scattered
     * reads through the grouped queue cost more than the divergence: Cornell box with a GGX sphere 368 -> 336 ms
     * per frame without it, the dragon and the Phong Cornell box +0.6 % / +1.3 %.) */
    PTC_OPT_GROUP_HITS = 11,
    /* 1 (default) = the kernel that adds a bounce's light terms to the path radiance runs
     * on a second stream of the context, beside the next bounce's extension traversal
     * (one is bound by HBM, the other by the vector memory pipeline); the context's stream
     * waits for it before the results are used, so callers see ordinary stream order.
     * 0 = everything on the one stream. Results do not depend on it. */
    PTC_OPT_OVERLAP_GATHER = 12,
    /* 1 = before the extension traversal of every bounce after the first, the live paths are reordered by
     * (cell of the ray origin on a 128^3 grid, direction octant): a radix sort of 24-bit keys plus one pass
     * that moves the packed state.  Results do not depend on it.  Default 0: on the dragon the more coherent
     * traversal saves less than the sort costs (extension traversal -13 %, the sort and the move 42 ms per
     * 1080p x 256 spp frame). */
    PTC_OPT_SORT_RAYS = 13,
    /* 1 (default) = a bounce in which at least 95 % of the live paths are expected to hit something (judged by
     * the bounce before it; camera rays: by the last batch's) skips the pass that compacts the hits into a queue:
     * the surface kernel then looks at every live path and ends the ones that missed itself.  Saves a kernel and 12
     * bytes of memory traffic per path in closed scenes.  Only with at most two light samples per hit and without
     * hit grouping.  0 = always compact.  Results do not depend on it. */
    PTC_OPT_DENSE_SURFACE = 14,
    /* 1 (default) = a scene's few very large triangles (a floor under a mesh: they sit in a leaf of their own under the
     * root of the hierarchy, and every ray that starts on them is inside that leaf's box) are looked at once more where a
     * ray is made: the surface kernel runs the first exits of the triangle test against their planes for the shadow rays
     * and the next segment of every hit, and a ray that cannot hit any of them carries a mark with which the traversal
     * kernels start below that leaf.  The exits are the triangle test's own, so the same hits are found; a quarter of
     * the dragon's shadow-ray triangle tests go away.  Path tracer only, scenes of triangles with one to three such
     * triangles.  0 = no ray is marked.  Results do not depend on it. */
    PTC_OPT_MARK_RAYS = 15
};

/* ----------------------------------------------------------- entry points -- */

/* Renderer.cpp:8-14 (Context::create, ray types, entry point). */
int  ptc_create(int device_ordinal, ptc_context** out);
int  ptc_destroy(ptc_context* ctx);
const char* ptc_last_error(const ptc_context* ctx);

/* Renderer.cpp:147-149 (config buffer). */
int  ptc_set_config(ptc_context* ctx, const PtcConfig* cfg);
/* Renderer.cpp:73,152-158: one of raytracer, analyticdirect, direct,
 * pathtracer, pathtracer2; anything else -> PTC_ERR_INTEGRATOR with the message
 * "Doesn't support the integrator <name>". */
int  ptc_set_integrator(ptc_context* ctx, const char* name);
/* Renderer.cpp:160-167 (integrator variables + samples_per_pixel). */
int  ptc_set_params(ptc_context* ctx, uint32_t light_samples, uint32_t light_stratify,
                    uint32_t next_event_est, uint32_t sampling_method, int32_t samples_per_pixel);
/* Renderer.cpp:115-125,174-180 (RT_FORMAT_USER buffers of the AoS PODs). */
int  ptc_set_triangles(ptc_context* ctx, const PtcTriangle* tris, size_t n);
int  ptc_set_spheres(ptc_context* ctx, const PtcSphere* spheres, size_t n);
/* Renderer.cpp:221-223. */
int  ptc_set_lights(ptc_context* ctx,
                    const PtcPointLight* pl, size_t n_point,
                    const PtcDirectionalLight* dl, size_t n_directional,
                    const PtcQuadLight* ql, size_t n_quad);
/* Renderer.cpp:47 (miss program's backgroundColor). */
int  ptc_set_background(ptc_context* ctx, float r, float g, float b);
/* Renderer.cpp:182-217,226: primitive bounds (Triangle.cu:62-72,
 * Sphere.cu:60-73) -> BVH built on the GPU. */
int  ptc_build(ptc_context* ctx);
/* Renderer.cpp:32-33: frameID + launch(0, width, height), synchronous. Renders
 * samples [0, samples_per_pixel), then averages and gamma-corrects
 * (PinholeCamera.cu:70-80). */
int  ptc_render(ptc_context* ctx, int frame_id, uint32_t width, uint32_t height);
/* Build-only: accumulate samples [spp_begin, spp_end) into the pre-gamma sum
 * buffer (cleared first when clear != 0) without resolving. Used for the
 * samples-per-pixel split across GPUs. The work is issued on the context's
 * stream, but the call is NOT asynchronous: the host waits for the stream once
 * per bounce (the number of surviving paths sizes the next launches), so it
 * returns with all but the last few kernels done; the sums are complete after
 * ptc_synchronize (or any later operation on the same stream). */
int  ptc_render_range(ptc_context* ctx, int frame_id, uint32_t width, uint32_t height,
                      int32_t spp_begin, int32_t spp_end, int clear);
/* No counterpart in the reference.  For callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, code that runs between the captures of HIP graphs): ptc_render_range with every
 * per-path and per-pixel work array taken from a fixed array in device memory that the library holds on
 * each GPU it runs on (PTC_STATIC_SCRATCH_MIB MiB, until the process ends) instead of from allocations.
 * The call allocates and frees nothing.  Samples [spp_begin, spp_end) of every pixel are added to the
 * caller's `sums_device` (device memory, width*height*3 floats, same indexing as ptc_read_sum; cleared
 * first when clear != 0); the context's own sum buffer and result are not touched.  The call allocates
 * nothing but it is host-synchronous: the host waits for the stream once per bounce (a 4-byte read-back of the
 * survivor count sizes the next launches) and once more before it returns, so it must NOT be issued on a stream
 * that is being captured into a HIP graph (the waits would fail with hipErrorStreamCaptureUnsupported and
 * invalidate the capture).  One sample of every pixel of the image, padded to whole 64x64
 * pixel blocks, must fit: (160 + 44 x light samples per segment) bytes per padded pixel, i.e. up to
 * 384x448 pixels with two quad lights; larger images -> PTC_ERR_INVALID with a message that
 * says so.  Scene, lights and options must be set up (ptc_set_*, ptc_build) before, with the ordinary
 * calls.  With reference seeding the range must start at 0 with clear (no per-pixel history is kept
 * between calls).  All contexts of a process on one GPU share that GPU's array, so one call at a time may
 * use it: a call that finds the array in use by another context or thread returns PTC_ERR_BUSY at once and has
 * touched nothing (try again later); a call that fails part way waits for its kernels before it lets go of the
 * array. */
#define PTC_STATIC_SCRATCH_MIB 48
int  ptc_render_range_static(ptc_context* ctx, int frame_id, uint32_t width, uint32_t height,
                             int32_t spp_begin, int32_t spp_end, int clear, void* sums_device, size_t sums_bytes);
/* Build-only: result = pow(sum / total_spp, 1/gamma) (frame 1) or the frame
 * lerp of PinholeCamera.cu:75-80. */
int  ptc_resolve(ptc_context* ctx, int frame_id, int32_t total_spp);
/* Build-only parity dump: closest hit of every pixel's sample-0 ray,
 * out[y*width + x]. */
int  ptc_primary_hits(ptc_context* ctx, uint32_t width, uint32_t height, PtcHitRecord* out);
/* Renderer.cpp:85,103 (map/unmap): width*height*3 floats, element (x,y) at
 * (y*width + x)*3, row 0 at the bottom. */
int  ptc_read_result(ptc_context* ctx, float* rgb_out);
/* Build-only: the pre-gamma radiance sums, same indexing. */
int  ptc_read_sum(ptc_context* ctx, float* rgb_out);
/* Build-only: let the caller own the device memory of the radiance sums
 * (width*height*3 floats), e.g. a tensor that a collective then reduces.
 * NULL returns to the context's own buffer.  The context reads, clears and
 * writes the buffer on ITS stream (ptc_set_stream): work of another stream
 * that touches it -- the fill that created it, a copy or a sum read from it --
 * has to be complete before the next ptc_render* call and must not start
 * before ptc_synchronize (or an event wait on the context's stream). */
int  ptc_bind_sum_buffer(ptc_context* ctx, void* device_ptr, size_t bytes);
/* Build-only: run on the caller's HIP stream (hipStream_t as void*). */
int  ptc_set_stream(ptc_context* ctx, void* hip_stream);
int  ptc_synchronize(ptc_context* ctx);

/* Build-only, needs no device: the host step of ptc_build that turns a finished hierarchy of
 * 64-byte two-child nodes (node 0 = root; per node lo0[3] hi0[3] lo1[3] hi1[3] as floats, then the
 * two child words and 8 bytes of padding; child word = kind << 30 | (count-1) << 27 | index with
 * kind 0 node, 1 triangle leaf, 2 sphere leaf, 3 empty) into 64-byte four-wide nodes with 8-bit child
 * boxes (origin[3], scale[3] as floats, then qlo.x[4] qlo.y[4] qlo.z[4] qhi.x[4] qhi.y[4] qhi.z[4]
 * as bytes, then four child words).  PTC_ERR_UNSUPPORTED = the hierarchy cannot be represented that
 * way (a box that is not finite, too deep): ptc_build then keeps the two-child form only.
 * Exposed so that the rounding of the child boxes can be checked without a GPU. */
int  ptc_collapse_wide(const void* binary_nodes, uint32_t n_nodes, void* wide_nodes_out, uint32_t capacity,
                       uint32_t* n_wide_out, uint32_t* depth_out);

/* ---- build-only test hooks: parts of the path on caller-supplied inputs, so that a divergence from the
 * CPU restatement can be localised to a function.  Host pointers; synchronous. */

/* One segment of a path: the ray (tmin = Config.epsilon), the payload it carries into the closest-hit
 * program (throughput, depth, the segment's seed) ... */
typedef struct PtcSegmentProbe { float o[3], d[3], thr[3]; uint32_t depth, seed; } PtcSegmentProbe;
/* ... and what comes out of rtTrace + miss / closest-hit program (PinholeCamera.cu:56-66): the segment's
 * radiance, the payload's throughput / origin / dir / depth afterwards, whether the bounce loop goes on
 * (!done && depth != maxDepth), and the closest hit itself (global primitive index, -1 = miss; t).
 * thr, o, d, depth are only defined where alive != 0. */
typedef struct PtcSegmentResult { float radiance[3], thr[3], o[3], d[3]; uint32_t depth; int32_t alive, prim; float t; } PtcSegmentResult;
/* Runs extension, surface, shadow and gather kernels once on n caller-made segments with the context's
 * scene, integrator and options (counter seeding and no zero-throughput cull are forced; the next
 * segment's seed is not part of the result). */
int  ptc_probe_segments(ptc_context* ctx, uint32_t n, const PtcSegmentProbe* in, PtcSegmentResult* out);
/* The device versions of the reference's shading helpers on n inputs of 16 floats each, 4 floats out:
 *   fn 0 phongBRDF (RayTracer.cu:259-263)   in: wi[3] reflect[3] Kd[3] Ks[3] shininess        out: rgb
 *   fn 1 ggxBRDF   (:265-285)               in: wi[3] wo[3] n[3] Ks[3] roughness              out: rgb
 *   fn 2 ggxPDF    (:287-294)               in: wi[3] n[3] h[3] - - - t roughness             out: pdf
 *   fn 3 Phong-sampled direction (:232-257, 302-322, 675-684, by-reference helpers)
 *                                           in: w[3] zeta0..2 t shininess sampling_method     out: w_i
 *   fn 4 GGX-sampled direction (:685-696, by-reference helpers)
 *                                           in: n[3] wo[3] zeta1 zeta2 roughness              out: w_i */
int  ptc_eval_shading(ptc_context* ctx, int fn, uint32_t n, const float* in16, float* out4);

/* The raw device counters of the PTC_OPT_COUNT_TRAVERSAL instantiations since the last ptc_reset_stats
 * (up to 40 words): [0..3] nodes visited / primitives tested by extension and shadow rays, [4] shadow rays,
 * [8..14] extension and [16..22] shadow rays in the kernel with deferred tests, per wave summed over the
 * launch: loop iterations; lanes owning a ray, visiting a node, standing at a leaf in them; full batches of
 * 64 tests; partial batches and the tests in them.  [24..29] extension and [32..37] shadow rays, same kernel,
 * over the loop iterations in which some lane visits a node: distinct nodes visited by the lanes of the wave;
 * iterations in which all visiting lanes are at ONE node, and the lanes in them; iterations with at most four
 * distinct nodes, and the lanes in them; lanes at the node most lanes of the wave share. */
int  ptc_debug_counters(ptc_context* ctx, uint64_t* out, uint32_t n);
/* Copies of the hierarchy ptc_build made on the device, in the node formats ptc_collapse_wide documents: the
 * two-child nodes and the four-wide nodes (which == 0: the hierarchy of the closest-hit rays, 1: of the any-hit
 * rays when PTC_OPT_BVH_BUILDER 3 built one).  Either output may be null; *n_* receive the node counts (0 wide
 * nodes = that form was not built).  PTC_ERR_INVALID when a non-null output is too small.  An unused child slot
 * of a device-built wide node holds a leaf reference to the all-zero triangle stored after the scene's (slot =
 * number of triangles) and an inside-out byte box, not the empty word. */
int  ptc_debug_hierarchy(ptc_context* ctx, int which, void* binary_out, uint32_t binary_capacity, uint32_t* n_binary,
                         void* wide_out, uint32_t wide_capacity, uint32_t* n_wide);
int  ptc_set_option(ptc_context* ctx, int option, int64_t value);
int  ptc_get_stats(ptc_context* ctx, PtcStats* out);
int  ptc_reset_stats(ptc_context* ctx);

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(PtcFloat3) == 12, "float3");
static_assert(sizeof(PtcMaterialValue) == 84, "MaterialValue is 84 bytes");
static_assert(offsetof(PtcMaterialValue, shininess) == 48, "MaterialValue.shininess");
static_assert(offsetof(PtcMaterialValue, matType) == 64, "MaterialValue.matType");
static_assert(sizeof(PtcTriangle) == 136 && offsetof(PtcTriangle, mv) == 48 &&
              offsetof(PtcTriangle, objType) == 132, "Triangle is 136 bytes");
static_assert(sizeof(PtcSphere) == 152 && offsetof(PtcSphere, mv) == 64 &&
              offsetof(PtcSphere, objType) == 148, "Sphere is 152 bytes");
static_assert(sizeof(PtcQuadLight) == 284 && offsetof(PtcQuadLight, color) == 272, "QuadLight is 284 bytes");
static_assert(sizeof(PtcPointLight) == 36 && sizeof(PtcDirectionalLight) == 24, "lights");
static_assert(sizeof(PtcConfig) == 88 && offsetof(PtcConfig, hSize) == 48 &&
              offsetof(PtcConfig, gamma) == 64 && offsetof(PtcConfig, russian_roul) == 80, "Config is 88 bytes");
static_assert(sizeof(PtcHitRecord) == 16, "HitRecord");
static_assert(sizeof(PtcStats) == 208 && offsetof(PtcStats, build_host_ms) == 168, "PtcStats is 208 bytes");
#endif

#endif /* PTCORE_H */
