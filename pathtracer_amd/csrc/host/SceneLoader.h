// SceneLoader.h -- parser of the reference's text scene format.
//
// Public surface follows OptiXRenderer/SceneLoader.h:19-47: one method,
// load(filename) -> shared_ptr<Scene>, throwing std::runtime_error
// "Unable to open scene file <name>" when the file cannot be opened.
// Command set and semantics: OptiXRenderer/SceneLoader.cpp:36-379.
#pragma once

#include <istream>
#include <memory>
#include <string>
#include <vector>

#include "Scene.h"
#include "../ptmath.h"

class SceneLoader {
    std::vector<mat4> transStack;  // top = back()

public:
    std::shared_ptr<Scene> load(std::string sceneFilename);
    // Same parser on an already opened stream (used for in-memory scene text).
    std::shared_ptr<Scene> loadStream(std::istream& in);
};
