/*
 * pthost.h -- C view of the host-side scene loader, so that non-C++ callers
 * (the Python tests and the benchmark) can parse the reference's scene files
 * with the same code the C++ Renderer uses.
 */
#ifndef PTHOST_H
#define PTHOST_H

#include "ptcore.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

/* Borrowed pointers into the loaded scene; valid until pth_scene_free. */
typedef struct PthSceneView {
    PtcConfig config;
    uint32_t width, height;
    uint32_t light_samples, light_stratify, samples_per_pixel, sampling_method;
    const char* integrator_name;
    const char* output_filename;
    const PtcTriangle* triangles;            uint64_t n_triangles;
    const PtcSphere* spheres;                uint64_t n_spheres;
    const PtcPointLight* plights;            uint64_t n_plights;
    const PtcDirectionalLight* dlights;      uint64_t n_dlights;
    const PtcQuadLight* qlights;             uint64_t n_qlights;
    const PtcFloat3* vertices;               uint64_t n_vertices;
} PthSceneView;

/* Parse a scene file / scene text.  Returns 0 or -1 (pth_last_error()). */
int pth_scene_load(const char* filename, pth_scene** out);
int pth_scene_load_text(const char* text, pth_scene** out);
void pth_scene_free(pth_scene* s);
int pth_scene_view(const pth_scene* s, PthSceneView* out);
const char* pth_last_error(void);

/* Renderer::getResult (Renderer.cpp:79-101): float3 result (row 0 at the
 * bottom) -> RGBA8, top row first. */
void pth_result_to_rgba8(const float* result, uint32_t width, uint32_t height, unsigned char* rgba_out);
/* resultToPNG (utils.cpp:3-14): write an RGBA8 image as PNG. Returns 0 or -1. */
int pth_write_png(const char* filename, const unsigned char* rgba, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif
