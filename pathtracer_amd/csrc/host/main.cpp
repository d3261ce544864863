// main.cpp -- headless command line: scene file in, PNG out.
//
// Follows the reference's offline branch (OptiXRenderer/main.cpp:27-45): load the
// scene, construct the Renderer, time renderer.run(false), print
// "Elapsed time: <s> seconds", write the PNG and print "<file> is saved.".
// Exceptions are printed and the exit status stays 0 like the reference
// (main.cpp:47-52).  Extra, optional flags select the build-only options.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>

#include "Renderer.h"
#include "SceneLoader.h"
#include "pthost.h"

int main(int argc, char** argv) {
    try {
        if (argc == 1) throw std::runtime_error("Missing path to a scene file as the first argument.");
        SceneLoader loader;
        std::shared_ptr<Scene> scene = loader.load(argv[1]);
        Renderer renderer(scene);
        for (int i = 2; i < argc; ++i) {
            if (!std::strcmp(argv[i], "--intended")) renderer.setOption(PTC_OPT_SEMANTICS, 1);
            else if (!std::strcmp(argv[i], "--literal")) renderer.setOption(PTC_OPT_SEMANTICS, 0);
            else if (!std::strcmp(argv[i], "--counter-seeds")) renderer.setOption(PTC_OPT_SEEDING, 1);
            else if (!std::strcmp(argv[i], "--bounce-cap") && i + 1 < argc) renderer.setOption(PTC_OPT_BOUNCE_CAP, std::atoll(argv[++i]));
            else throw std::runtime_error(std::string("unknown option ") + argv[i]);
        }
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
    } catch (const std::exception& ex) {
        std::cerr << ex.what() << std::endl;
    }
    return 0;
}
