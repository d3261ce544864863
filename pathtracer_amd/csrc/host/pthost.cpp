// pthost.cpp -- C wrappers over SceneLoader and the PNG writer.
#include "pthost.h"

#include <memory>
#include <sstream>
#include <string>

#include "SceneLoader.h"
#include "png_writer.h"

struct pth_scene {
    std::shared_ptr<Scene> scene;
};

namespace {
thread_local std::string g_error;
}

extern "C" {

const char* pth_last_error(void) { return g_error.c_str(); }

int pth_scene_load(const char* filename, pth_scene** out) {
    if (!filename || !out) { g_error = "null argument"; return -1; }
    try {
        SceneLoader loader;
        std::unique_ptr<pth_scene> h(new pth_scene);
        h->scene = loader.load(filename);
        *out = h.release();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

int pth_scene_load_text(const char* text, pth_scene** out) {
    if (!text || !out) { g_error = "null argument"; return -1; }
    try {
        SceneLoader loader;
        std::istringstream in(text);
        std::unique_ptr<pth_scene> h(new pth_scene);
        h->scene = loader.loadStream(in);
        *out = h.release();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

void pth_scene_free(pth_scene* s) { delete s; }

int pth_scene_view(const pth_scene* s, PthSceneView* v) {
    if (!s || !v) { g_error = "null argument"; return -1; }
    const Scene& sc = *s->scene;
    v->config = sc.config;
    v->width = sc.width;
    v->height = sc.height;
    v->light_samples = sc.light_samples;
    v->light_stratify = sc.light_stratify;
    v->samples_per_pixel = sc.samples_per_pixel;
    v->sampling_method = sc.sampling_method;
    v->integrator_name = sc.integratorName.c_str();
    v->output_filename = sc.outputFilename.c_str();
    v->triangles = sc.triangles.data();  v->n_triangles = sc.triangles.size();
    v->spheres = sc.spheres.data();      v->n_spheres = sc.spheres.size();
    v->plights = sc.plights.data();      v->n_plights = sc.plights.size();
    v->dlights = sc.dlights.data();      v->n_dlights = sc.dlights.size();
    v->qlights = sc.qlights.data();      v->n_qlights = sc.qlights.size();
    v->vertices = sc.vertices.data();    v->n_vertices = sc.vertices.size();
    return 0;
}

void pth_result_to_rgba8(const float* result, uint32_t width, uint32_t height, unsigned char* out) {
    for (uint32_t row = 0; row < height; ++row) {
        const float* src = result + (size_t)(height - 1 - row) * width * 3;  // flip vertically
        unsigned char* dst = out + (size_t)row * width * 4;
        for (uint32_t x = 0; x < width; ++x) {
            for (int ch = 0; ch < 3; ++ch) {
                float v = src[x * 3 + ch];
                v = v > 1.f ? 1.f : v < 0.f ? 0.f : v;  // NaN stays NaN -> cast below
                dst[x * 4 + ch] = static_cast<unsigned char>(v * 255);
            }
            dst[x * 4 + 3] = 255;
        }
    }
}

int pth_write_png(const char* filename, const unsigned char* rgba, uint32_t width, uint32_t height) {
    try {
        std::string err;
        if (!png_write_rgba8(filename, rgba, width, height, &err)) { g_error = err; return -1; }
        return 0;
    } catch (const std::exception& e) {  // the encoder's buffers
        g_error = e.what();
        return -1;
    }
}

}  // extern "C"
