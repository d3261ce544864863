/*
 * oracle.h -- CPU oracle for the per-pixel path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (pathtracer_amd/, the
 * C-ABI library) may include, link or call this; only tests/, the smoke check
 * and the CPU-baseline leg of the benchmark do, as the checker.
 *
 * It is a plain C++ restatement of what the reference's OptiX programs compute
 * (PinholeCamera.cu, Triangle.cu, Sphere.cu, Common.cu, RayTracer.cu,
 * random.h), one function per program, each citing the lines it follows.
 *
 * PARITY UNPINNED at the OptiX-SDK level: the reference cannot be compiled
 * here (it needs the OptiX 6.x SDK headers and runtime, which this image
 * lacks) and it ships no tests or golden vectors.  The vector/matrix helpers
 * (normalize, reflect, Matrix4x4::inverse ...) restate the SDK's documented
 * formulas; the closed Trbvh traversal is replaced by "closest hit over all
 * primitives, lowest primitive index wins ties".  The only recorded output of
 * the reference that matches the committed integrators is images/sphere.png
 * (scene 2/sphere.test, `direct`), which the test-suite checks against.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "ptcore.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_RAYTRACER = 0, ORC_ANALYTICDIRECT = 1, ORC_DIRECT = 2, ORC_PATHTRACER = 3, ORC_PATHTRACER2 = 4 };

typedef struct OrcScene {
    PtcConfig config;
    const PtcTriangle* tris;           uint64_t n_tris;
    const PtcSphere* spheres;          uint64_t n_spheres;
    const PtcPointLight* plights;      uint64_t n_plights;
    const PtcDirectionalLight* dlights;uint64_t n_dlights;
    const PtcQuadLight* qlights;       uint64_t n_qlights;
    uint32_t light_samples, light_stratify, next_event_est, sampling_method;
    int32_t  samples_per_pixel;
    int32_t  integrator;               /* ORC_* */
    float    background[3];
    uint32_t width, height;
} OrcScene;

typedef struct OrcOptions {
    int32_t  semantics;   /* 0 literal, 1 intended (see PTC_OPT_SEMANTICS)        */
    int32_t  seeding;     /* 0 reference, 1 counter (see PTC_OPT_SEEDING)         */
    uint32_t bounce_cap;  /* hard limit on segments per path                      */
    int32_t  cull_zero;   /* stop a path whose throughput is exactly 0 (counter seeding only) */
    int32_t  use_bvh;     /* 0 brute force over all primitives, 1 CPU SAH BVH     */
    int32_t  threads;     /* worker threads over pixel rows, <=0: 1               */
    int32_t  frame_id;    /* frameID, 1 for a single-frame render                 */
    int32_t  spp_begin, spp_end; /* sample range; spp_end<=0: [0, samples_per_pixel) */
    /* pixel window [x0,x1) x [y0,y1) and stride; all 0 = whole image, stride 1 */
    uint32_t x0, y0, x1, y1, stride;
} OrcOptions;

typedef struct OrcStats {
    uint64_t paths, extension_rays, shadow_rays;
    double   seconds;
} OrcStats;

/* Closest hit of each pixel's sample-0 camera ray, out[y*width + x]. */
int orc_primary_hits(const OrcScene* s, const OrcOptions* o, PtcHitRecord* out);

/* Full render.  sum_out (may be NULL): pre-gamma radiance sums; result_out (may
 * be NULL): pow(sum/spp, 1/gamma).  Both width*height*3 floats, element (x,y)
 * at (y*width+x)*3, row 0 at the bottom.  Pixels outside the window/stride are
 * left untouched. */
int orc_render(const OrcScene* s, const OrcOptions* o, float* sum_out, float* result_out, OrcStats* stats);

/* One pass of the bounce loop's body for n caller-made segments, and the shading helpers on raw
 * inputs: the CPU side of ptc_probe_segments / ptc_eval_shading (same layouts, include/ptcore.h). */
int orc_probe_segments(const OrcScene* s, const OrcOptions* o, uint32_t n, const PtcSegmentProbe* in, PtcSegmentResult* out);
int orc_eval_shading(int fn, uint32_t n, const float* in16, float* out4);

/* Known-answer helpers for the RNG (random.h:31-67). */
uint32_t orc_tea16(uint32_t v0, uint32_t v1);
uint32_t orc_lcg(uint32_t* state);
float    orc_rnd(uint32_t* state);
/* refract / fresnel_schlick of the OptiX vector header as restated; phaseHG of RayTracer.cu:327 */
int      orc_refract(const float* i3, const float* n3, float ior, float* r3);
float    orc_fresnel_schlick(float cos_theta);
float    orc_phase_hg(float wi_dot_wo, float g);

/* Renderer::getResult's float3 -> RGBA8 conversion (Renderer.cpp:79-101):
 * flip rows, clamp to [0,1], truncate v*255. */
void orc_to_rgba8(const float* result, uint32_t width, uint32_t height, unsigned char* rgba_out);

/* 4x4 inverse used by the sphere programs (exposed for unit tests). */
void orc_mat4_inverse(const float* m16, float* out16);

const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
