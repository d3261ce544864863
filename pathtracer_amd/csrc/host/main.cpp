// main.cpp -- headless command line: scene file in, PNG out.
//
// Follows the reference's offline branch (OptiXRenderer/main.cpp:27-45): load the
// scene, construct the Renderer, time renderer.run(false), print
// "Elapsed time: <s> seconds", write the PNG and print "<file> is saved.".
// Exceptions are printed and the exit status stays 0 like the reference
// (main.cpp:47-52).  Extra, optional flags select the build-only options; `--gpus N` renders the frame on N GPUs of
// the node (samples split, one RCCL reduce; `--ordered-sum`: deterministic host adds instead; with it `--share-devices`
// allows more contexts than devices, which is how a one-GPU machine exercises the several-context path).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>

#include <utility>
#include <vector>

#include "MultiGpu.h"
#include "Renderer.h"
#include "SceneLoader.h"
#include "pthost.h"

namespace {
// The command line's options: PTC_OPT_* values for every context, the number of GPUs, the way their sums are combined.
struct Options {
    std::vector<std::pair<int, long long>> core;
    int gpus = 0;  // 0: the reference's single-GPU Renderer; N >= 1: MultiGpuRenderer over N devices (RCCL reduce)
    bool ordered = false;
    bool share = false;  // more contexts than devices (ordered sum only): the several-context path on a one-GPU machine
};

Options parse(int argc, char** argv) {
    Options o;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--intended")) o.core.push_back({PTC_OPT_SEMANTICS, 1});
        else if (!std::strcmp(argv[i], "--literal")) o.core.push_back({PTC_OPT_SEMANTICS, 0});
        else if (!std::strcmp(argv[i], "--counter-seeds")) o.core.push_back({PTC_OPT_SEEDING, 1});
        else if (!std::strcmp(argv[i], "--bounce-cap") && i + 1 < argc) o.core.push_back({PTC_OPT_BOUNCE_CAP, std::atoll(argv[++i])});
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) o.gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--ordered-sum")) o.ordered = true;
        else if (!std::strcmp(argv[i], "--share-devices")) o.share = true;
        else throw std::runtime_error(std::string("unknown option ") + argv[i]);
    }
    return o;
}

template <class R>
void renderAndSave(R& renderer, const Options& o) {
    for (auto& kv : o.core) renderer.setOption(kv.first, kv.second);
    auto begin = std::chrono::steady_clock::now();
    renderer.run(false);
    auto end = std::chrono::steady_clock::now();
    std::cout << "Elapsed time: " << std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count() * float(1e-6) << " seconds"
              << std::endl;
    std::vector<unsigned char> img = renderer.getResult();
    if (pth_write_png(renderer.getOutputFilename().c_str(), img.data(), renderer.getWidth(), renderer.getHeight()) != 0)
        throw std::runtime_error(pth_last_error());
    std::cout << renderer.getOutputFilename() << " is saved." << std::endl;
    PtcStats st = renderer.getStats();
    std::cout << st.extension_rays << " extension rays, " << st.shadow_rays << " shadow rays" << std::endl;
}

// MultiGpuRenderer with Renderer's run(bool) signature
struct MultiRun {
    MultiGpuRenderer& r;
    void setOption(int k, long long v) { r.setOption(k, v); }
    void run(bool) { r.run(); }
    std::vector<unsigned char> getResult() { return r.getResult(); }
    std::string getOutputFilename() { return r.getOutputFilename(); }
    int getWidth() { return r.getWidth(); }
    int getHeight() { return r.getHeight(); }
    PtcStats getStats() { return r.getStats(); }
};
}  // namespace

int main(int argc, char** argv) {
    try {
        if (argc == 1) throw std::runtime_error("Missing path to a scene file as the first argument.");
        SceneLoader loader;
        std::shared_ptr<Scene> scene = loader.load(argv[1]);
        const Options o = parse(argc, argv);
        if (o.gpus >= 1) {
            MultiGpuRenderer multi(scene, o.gpus, o.ordered, o.share);
            std::cout << "rendering on " << multi.deviceCount() << " GPU(s), sums combined by " << (o.ordered ? "ordered host adds" : "one RCCL reduce") << std::endl;
            MultiRun run{multi};
            renderAndSave(run, o);
        } else {
            Renderer renderer(scene);
            renderAndSave(renderer, o);
        }
    } catch (const std::exception& ex) {
        std::cerr << ex.what() << std::endl;
    }
    return 0;
}
