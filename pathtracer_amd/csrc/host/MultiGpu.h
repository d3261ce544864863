// MultiGpu.h -- one frame rendered by several GPUs of one node from ONE host process.
//
// The reference has no multi-GPU path.  The work shards over samples: device g of G renders the global
// sample indices [g*spp/G, (g+1)*spp/G) of every pixel into its own pre-gamma radiance sums (one core
// context, one stream and one host thread per device, counter-based seeds so that the image does not depend on
// G), the sums are combined on device 0 -- one ncclReduce over xGMI, or, `ordered`, copied to the host and
// added in device order, which is bit-reproducible -- and device 0 averages and gamma-corrects
// (PinholeCamera.cu:70-71 does both after its sample loop, so the combine has to come first).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "Scene.h"
#include "ptcore.h"

class MultiGpuRenderer {
    struct Dev;
    std::vector<Dev*> devs;
    std::shared_ptr<Scene> scene;
    bool ordered = false;
    unsigned int width = 0, height = 0;
    void release();  // contexts, communicators, buffers, streams

public:
    // devices 0 .. n_devices-1; throws std::runtime_error (fewer devices present, HIP / RCCL errors, core errors).
    // `share_devices` (with `ordered_sum` only: a communicator cannot hold one device twice) lets n_devices exceed the
    // devices present: context g then lives on device g mod present.  It is how a one-GPU machine runs the
    // several-context path (threads, sample ranges, ordered sum); it is not faster than one context.
    MultiGpuRenderer(std::shared_ptr<Scene> scene, int n_devices, bool ordered_sum = false, bool share_devices = false);
    ~MultiGpuRenderer();
    MultiGpuRenderer(const MultiGpuRenderer&) = delete;
    MultiGpuRenderer& operator=(const MultiGpuRenderer&) = delete;

    void setOption(int option, long long value);  // on every device's context (PTC_OPT_SEEDING stays 1)
    void run();                                   // the whole frame: sample ranges, combine, resolve on device 0
    std::vector<float> getResultFloat();          // float3 image of device 0, row 0 at the bottom
    std::vector<unsigned char> getResult();       // RGBA8, top row first (Renderer.cpp:76-106)
    PtcStats getStats();                          // ray counts summed over the devices, times of device 0
    int getWidth() const { return (int)width; }
    int getHeight() const { return (int)height; }
    std::string getOutputFilename() const { return scene->outputFilename; }
    int deviceCount() const { return (int)devs.size(); }
};
