// Renderer.h -- the reference's Renderer surface (OptiXRenderer/Renderer.h:26-87)
// implemented over the C ABI of ptcore.h instead of an OptiX context.
//
// Kept: Renderer(shared_ptr<Scene>), buildScene(), run(bool progressive = true),
// getResult() (RGBA8, top row first), the getters.  Replaced: getResultBuffer()
// returned an optix::Buffer& for the GL viewer; here getResultFloat() hands out
// the float3 image (row 0 at the bottom) that buffer held.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "Scene.h"
#include "ptcore.h"

class Renderer {
    ptc_context* context = nullptr;
    std::shared_ptr<Scene> scene;
    int numFrames = 1;
    int currentFrame = 0;
    std::string outputFilename;
    unsigned int width = 0, height = 0;

    void check(int rc);  // non-zero -> std::runtime_error(ptc_last_error)

public:
    explicit Renderer(std::shared_ptr<Scene> scene, int device = 0);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void buildScene();
    void run(bool progressive = true);
    std::vector<unsigned char> getResult();
    std::vector<float> getResultFloat();

    // build-only: semantics / seeding / bounce cap ... (PTC_OPT_*)
    void setOption(int option, long long value);
    PtcStats getStats();

    std::string getOutputFilename() { return outputFilename; }
    int getWidth() { return width; }
    int getHeight() { return height; }
    int getNumFrames() { return numFrames; }
    int getCurrentFrame() { return currentFrame; }
    std::shared_ptr<Scene> getScene() { return scene; }
};
