#include "MultiGpu.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <thread>

#include "pthost.h"

struct MultiGpuRenderer::Dev {
    int device = 0;
    ptc_context* ctx = nullptr;
    hipStream_t stream = nullptr;
    float* sums = nullptr;
    float* host_sums = nullptr;  // pinned; ordered sum only
    ncclComm_t comm = nullptr;
    std::string error;  // first failure of this device's thread
};

namespace {
void hipCheck(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void ncclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
void coreCheck(ptc_context* c, int rc) {
    if (rc != PTC_OK) throw std::runtime_error(ptc_last_error(c));
}
}  // namespace

MultiGpuRenderer::MultiGpuRenderer(std::shared_ptr<Scene> scene_, int n_devices, bool ordered_sum, bool share_devices)
    : scene(scene_), ordered(ordered_sum) {
    int present = 0;
    hipCheck(hipGetDeviceCount(&present), "hipGetDeviceCount");
    if (share_devices && !ordered_sum) throw std::runtime_error("devices can only be shared with the ordered sum (an RCCL communicator cannot hold a device twice)");
    if (n_devices < 1 || present < 1 || (n_devices > present && !share_devices))
        throw std::runtime_error("asked for " + std::to_string(n_devices) + " GPUs, " + std::to_string(present) + " present");
    width = scene->width;
    height = scene->height;
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    try {
        // One host thread per context: uploads and the hierarchy build of every device run side by side
        // (the scene is replicated; each build is tens of milliseconds of launches and waits on its own device).
        for (int g = 0; g < n_devices; ++g) {
            devs.push_back(new Dev);
            devs[g]->device = g % present;
        }
        auto setUp = [this, bytes](Dev* d) {
            try {
                if (ptc_create(d->device, &d->ctx) != PTC_OK) throw std::runtime_error(ptc_last_error(nullptr));
                hipCheck(hipSetDevice(d->device), "hipSetDevice");
                hipCheck(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking), "hipStreamCreate");
                hipCheck(hipMalloc((void**)&d->sums, bytes), "hipMalloc");
                if (ordered) hipCheck(hipHostMalloc((void**)&d->host_sums, bytes), "hipHostMalloc");
                ptc_context* c = d->ctx;
                coreCheck(c, ptc_set_stream(c, d->stream));
                coreCheck(c, ptc_bind_sum_buffer(c, d->sums, bytes));
                coreCheck(c, ptc_set_option(c, PTC_OPT_SEEDING, 1));  // the image must not depend on the partition
                // the same uploads as Renderer::buildScene, on every device
                coreCheck(c, ptc_set_config(c, &scene->config));
                coreCheck(c, ptc_set_integrator(c, scene->integratorName.c_str()));
                coreCheck(c, ptc_set_params(c, scene->light_samples, scene->light_stratify, scene->config.next_event_est, scene->sampling_method,
                                            (int32_t)scene->samples_per_pixel));
                coreCheck(c, ptc_set_triangles(c, scene->triangles.data(), scene->triangles.size()));
                coreCheck(c, ptc_set_spheres(c, scene->spheres.data(), scene->spheres.size()));
                coreCheck(c, ptc_set_lights(c, scene->plights.data(), scene->plights.size(), scene->dlights.data(), scene->dlights.size(),
                                            scene->qlights.data(), scene->qlights.size()));
                coreCheck(c, ptc_set_background(c, 0.f, 0.f, 0.f));
                coreCheck(c, ptc_build(c));
            } catch (const std::exception& e) {
                d->error = e.what();
            }
        };
        {
            std::vector<std::thread> threads;
            for (int g = 1; g < n_devices; ++g) threads.emplace_back(setUp, devs[g]);
            setUp(devs[0]);
            for (auto& t : threads) t.join();
        }
        for (Dev* d : devs)
            if (!d->error.empty()) throw std::runtime_error("GPU " + std::to_string(d->device) + ": " + d->error);
        if (!ordered) {
            std::vector<ncclComm_t> comms(n_devices);
            std::vector<int> ids(n_devices);
            for (int g = 0; g < n_devices; ++g) ids[g] = devs[g]->device;
            ncclCheck(ncclCommInitAll(comms.data(), n_devices, ids.data()), "ncclCommInitAll");
            for (int g = 0; g < n_devices; ++g) devs[g]->comm = comms[g];
        }
    } catch (...) {
        release();
        throw;
    }
}

MultiGpuRenderer::~MultiGpuRenderer() { release(); }

void MultiGpuRenderer::release() {
    for (Dev* d : devs) {
        (void)hipSetDevice(d->device);
        if (d->comm) (void)ncclCommDestroy(d->comm);
        if (d->ctx) (void)ptc_destroy(d->ctx);
        if (d->sums) (void)hipFree(d->sums);
        if (d->host_sums) (void)hipHostFree(d->host_sums);
        if (d->stream) (void)hipStreamDestroy(d->stream);
        delete d;
    }
    devs.clear();
}

void MultiGpuRenderer::setOption(int option, long long value) {
    if (option == PTC_OPT_SEEDING && value != 1) throw std::runtime_error("several GPUs need counter seeding (PTC_OPT_SEEDING 1)");
    for (Dev* d : devs) coreCheck(d->ctx, ptc_set_option(d->ctx, option, value));
}

void MultiGpuRenderer::run() {
    const int G = (int)devs.size();
    const int64_t spp = scene->samples_per_pixel;
    const size_t count = (size_t)width * height * 3;
    // 1. every device its share of the sample index space, one host thread each
    std::vector<std::thread> threads;
    for (int g = 0; g < G; ++g) {
        threads.emplace_back([this, g, G, spp]() {
            Dev* d = devs[g];
            const int32_t j0 = (int32_t)(spp * g / G), j1 = (int32_t)(spp * (g + 1) / G);
            if (ptc_render_range(d->ctx, 1, width, height, j0, j1, 1) != PTC_OK) d->error = ptc_last_error(d->ctx);
        });
    }
    for (auto& t : threads) t.join();
    for (Dev* d : devs)
        if (!d->error.empty()) throw std::runtime_error("GPU " + std::to_string(d->device) + ": " + d->error);
    // 2. combine on device 0
    if (ordered) {
        // every device copies its sums to pinned host memory on its own stream, all copies in flight together;
        // the host then adds them in device order, ((s0 + s1) + s2) + ...
        for (Dev* d : devs) {
            hipCheck(hipSetDevice(d->device), "hipSetDevice");
            hipCheck(hipMemcpyAsync(d->host_sums, d->sums, count * sizeof(float), hipMemcpyDeviceToHost, d->stream), "hipMemcpy");
        }
        float* acc = devs[0]->host_sums;
        for (int g = 0; g < G; ++g) {
            Dev* d = devs[g];
            hipCheck(hipSetDevice(d->device), "hipSetDevice");
            hipCheck(hipStreamSynchronize(d->stream), "hipStreamSynchronize");
            if (g) {
                const float* part = d->host_sums;
                for (size_t i = 0; i < count; ++i) acc[i] += part[i];
            }
        }
        hipCheck(hipSetDevice(devs[0]->device), "hipSetDevice");
        hipCheck(hipMemcpyAsync(devs[0]->sums, acc, count * sizeof(float), hipMemcpyHostToDevice, devs[0]->stream), "hipMemcpy");
        hipCheck(hipStreamSynchronize(devs[0]->stream), "hipStreamSynchronize");
    } else {
        // one collective; each device's part is ordered after its render by its stream
        ncclCheck(ncclGroupStart(), "ncclGroupStart");
        for (Dev* d : devs) ncclCheck(ncclReduce(d->sums, d->sums, count, ncclFloat, ncclSum, 0, d->comm, d->stream), "ncclReduce");
        ncclCheck(ncclGroupEnd(), "ncclGroupEnd");
        for (Dev* d : devs) {
            hipCheck(hipSetDevice(d->device), "hipSetDevice");
            hipCheck(hipStreamSynchronize(d->stream), "hipStreamSynchronize");
        }
    }
    // 3. average + gamma on device 0
    coreCheck(devs[0]->ctx, ptc_resolve(devs[0]->ctx, 1, (int32_t)spp));
    coreCheck(devs[0]->ctx, ptc_synchronize(devs[0]->ctx));
}

std::vector<float> MultiGpuRenderer::getResultFloat() {
    std::vector<float> buf((size_t)width * height * 3);
    coreCheck(devs[0]->ctx, ptc_read_result(devs[0]->ctx, buf.data()));
    return buf;
}

std::vector<unsigned char> MultiGpuRenderer::getResult() {
    std::vector<float> buf = getResultFloat();
    std::vector<unsigned char> imageData((size_t)width * height * 4);
    pth_result_to_rgba8(buf.data(), width, height, imageData.data());
    return imageData;
}

PtcStats MultiGpuRenderer::getStats() {
    PtcStats total{};
    for (size_t g = 0; g < devs.size(); ++g) {
        PtcStats st;
        coreCheck(devs[g]->ctx, ptc_get_stats(devs[g]->ctx, &st));
        if (g == 0) total = st;
        else {
            total.paths += st.paths;
            total.extension_rays += st.extension_rays;
            total.shadow_rays += st.shadow_rays;
        }
    }
    return total;
}
