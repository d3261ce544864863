// Scene.h -- what the scene-file loader produces and the Renderer uploads.
//
// Member names, defaults and meaning are those of the reference's Scene
// (OptiXRenderer/Scene.h:10-46), so code written against it keeps compiling; the element
// types are the C-ABI PODs of ptcore.h (layout-identical to the reference's device
// structs), which lets the vectors go through the ABI without conversion.
#pragma once

#include <string>
#include <vector>

#include "ptcore.h"

struct Scene {
    Scene();  // camera / config defaults, see SceneLoader.cpp

    // ---- geometry (world space: the loader applies the transform stack to triangles)
    std::vector<PtcTriangle> triangles;  // quad lights contribute two LIGHT triangles each
    std::vector<PtcSphere> spheres;
    std::vector<PtcFloat3> vertices;     // the `vertex` pool `tri` indexes into

    // ---- emitters
    std::vector<PtcQuadLight> qlights;
    std::vector<PtcPointLight> plights;
    std::vector<PtcDirectionalLight> dlights;

    // ---- integrator and sampling
    std::string integratorName = "raytracer";
    unsigned int samples_per_pixel = 1;                       // `spp`
    unsigned int light_samples = 1;                           // `lightsamples`
    unsigned int light_stratify = 0;                          // `lightstratify on|off`
    unsigned int sampling_method = PTC_HEMISPHERE_SAMPLING;   // `importancesampling`
    PtcConfig config;                                         // camera frame, maxdepth, NEE, roulette, gamma

    // ---- image
    unsigned int width = 0, height = 0;
    std::string outputFilename = "raytrace.png";
};
