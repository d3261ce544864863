#include "Renderer.h"

#include <stdexcept>

#include "pthost.h"

void Renderer::check(int rc) {
    if (rc != PTC_OK) throw std::runtime_error(ptc_last_error(context));
}

Renderer::Renderer(std::shared_ptr<Scene> scene_, int device) : scene(scene_) {
    if (ptc_create(device, &context) != PTC_OK) throw std::runtime_error(ptc_last_error(nullptr));
    buildScene();
}

Renderer::~Renderer() {
    if (context) ptc_destroy(context);
}

void Renderer::setOption(int option, long long value) { check(ptc_set_option(context, option, value)); }

PtcStats Renderer::getStats() {
    PtcStats st;
    check(ptc_get_stats(context, &st));
    return st;
}

// Same steps as the reference's buildScene (Renderer.cpp:127-227): record the image
// info, upload config, pick the integrator (unknown name -> "Doesn't support the
// integrator <name>"), pass the integrator / ray-generation variables, upload
// triangles, spheres and lights, build the acceleration structure.
void Renderer::buildScene() {
    width = scene->width;
    height = scene->height;
    outputFilename = scene->outputFilename;
    currentFrame = 0;
    numFrames = 1;
    check(ptc_set_config(context, &scene->config));
    check(ptc_set_integrator(context, scene->integratorName.c_str()));
    check(ptc_set_params(context, scene->light_samples, scene->light_stratify, scene->config.next_event_est, scene->sampling_method,
                         (int32_t)scene->samples_per_pixel));
    check(ptc_set_triangles(context, scene->triangles.data(), scene->triangles.size()));
    check(ptc_set_spheres(context, scene->spheres.data(), scene->spheres.size()));
    check(ptc_set_lights(context, scene->plights.data(), scene->plights.size(), scene->dlights.data(), scene->dlights.size(),
                         scene->qlights.data(), scene->qlights.size()));
    check(ptc_set_background(context, 0.f, 0.f, 0.f));
    check(ptc_build(context));
}

void Renderer::run(bool progressive) {
    while (currentFrame != numFrames) {
        check(ptc_render(context, ++currentFrame, width, height));
        if (progressive) break;
    }
}

std::vector<float> Renderer::getResultFloat() {
    std::vector<float> buf((size_t)width * height * 3);
    check(ptc_read_result(context, buf.data()));
    return buf;
}

std::vector<unsigned char> Renderer::getResult() {
    std::vector<float> buf = getResultFloat();
    std::vector<unsigned char> imageData((size_t)width * height * 4);
    pth_result_to_rgba8(buf.data(), width, height, imageData.data());
    return imageData;
}
