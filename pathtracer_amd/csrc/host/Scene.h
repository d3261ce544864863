// Scene.h -- host-side scene description handed to the Renderer.
//
// Same fields, defaults and meaning as the reference's Scene
// (OptiXRenderer/Scene.h:10-46); the element types are the C-ABI PODs of
// ptcore.h, which are layout-identical to the reference's device structs.
#pragma once

#include <string>
#include <vector>

#include "ptcore.h"

struct Scene {
    // Info about the output image
    std::string outputFilename = "raytrace.png";
    unsigned int width = 0, height = 0;

    PtcConfig config;

    std::string integratorName = "raytracer";
    unsigned int light_samples = 1;
    unsigned int light_stratify = 0;
    unsigned int samples_per_pixel = 1;
    unsigned int sampling_method = PTC_HEMISPHERE_SAMPLING;

    std::vector<PtcFloat3> vertices;
    std::vector<PtcTriangle> triangles;
    std::vector<PtcSphere> spheres;
    std::vector<PtcDirectionalLight> dlights;
    std::vector<PtcPointLight> plights;
    std::vector<PtcQuadLight> qlights;

    Scene();
};
