// SceneLoader.cpp -- table-driven implementation of the scene-file grammar.
//
// One handler per command; a command whose arguments fail to parse prints
// "Failed reading value <i> will skip" and is ignored, unknown commands are
// ignored silently, lines that are blank or start with '#' are skipped
// (reference behaviour: OptiXRenderer/SceneLoader.cpp:20-33, 75-79).
#include "SceneLoader.h"

#include <cmath>
#include <fstream>
#include <functional>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace {

PtcFloat3 P3(vec3 v) { return PtcFloat3{v.x, v.y, v.z}; }
vec3 V3(const PtcFloat3& v) { return mk3(v.x, v.y, v.z); }

// Config defaults, OptiXRenderer/Config.h:22-35.
PtcConfig defaultConfig() {
    PtcConfig c{};
    c.gamma = 1.0f;
    c.w = PtcFloat3{1.f, 0.f, 0.f};
    c.u = PtcFloat3{0.f, 1.f, 0.f};
    c.v = PtcFloat3{0.f, 0.f, 1.f};
    c.eye = PtcFloat3{0.f, 0.f, 0.f};
    c.hSize.x = c.hSize.y = 100.f;
    c.tanHFov.x = c.tanHFov.y = tanf(0.25f * PT_PIf);
    c.maxDepth = 5;
    c.next_event_est = PTC_NEE_OFF;
    c.russian_roul = 0;
    c.epsilon = 0.0001f;
    return c;
}

template <class T>
bool readValues(std::stringstream& s, int numvals, T* values) {
    for (int i = 0; i < numvals; i++) {
        s >> values[i];
        if (s.fail()) {
            std::cout << "Failed reading value " << i << " will skip" << std::endl;
            return false;
        }
    }
    return true;
}

// float -> unsigned the way the reference's x86-64 host build does it:
// negative values such as `maxdepth -1` come out as 0xFFFFFFFF.
unsigned int toUnsigned(float f) {
    if (f < 0.0f) return (unsigned int)(long long)f;
    return (unsigned int)f;
}

struct ParseState {
    Scene* scene;
    PtcMaterialValue mv;
    vec3 attenuation;
    std::vector<mat4>* stack;
};

using Handler = std::function<void(ParseState&, std::stringstream&)>;

void setVec(std::stringstream& s, PtcFloat3& dst) {
    float f[3];
    if (readValues(s, 3, f)) dst = PtcFloat3{f[0], f[1], f[2]};
}

bool isOn(const std::string& v) { return v.compare("on") == 0; }

const std::unordered_map<std::string, Handler>& handlers() {
    static const std::unordered_map<std::string, Handler> table = {
        {"size", [](ParseState& p, std::stringstream& s) {
             float f[2];
             if (!readValues(s, 2, f)) return;
             p.scene->width = (unsigned int)f[0];
             p.scene->height = (unsigned int)f[1];
             p.scene->config.hSize.x = f[0] / 2.f;
             p.scene->config.hSize.y = f[1] / 2.f;
         }},
        {"maxdepth", [](ParseState& p, std::stringstream& s) {
             float f[1];
             if (readValues(s, 1, f)) p.scene->config.maxDepth = toUnsigned(f[0]);
         }},
        {"output", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (readValues(s, 1, v)) p.scene->outputFilename = v[0];
         }},
        {"integrator", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (readValues(s, 1, v)) p.scene->integratorName = v[0];
         }},
        {"camera", [](ParseState& p, std::stringstream& s) {
             float f[10];
             if (!readValues(s, 10, f)) return;
             PtcConfig& c = p.scene->config;
             vec3 eye = mk3(f[0], f[1], f[2]), center = mk3(f[3], f[4], f[5]), up = mk3(f[6], f[7], f[8]);
             vec3 w = normalize(eye - center);
             vec3 u = normalize(cross(up, w));
             vec3 v = normalize(cross(w, u));
             c.eye = P3(eye); c.w = P3(w); c.u = P3(u); c.v = P3(v);
             c.tanHFov.y = tanf(f[9] / 180.f * PT_PIf * 0.5f);
             c.tanHFov.x = c.tanHFov.y * c.hSize.x / c.hSize.y;  // needs `size` first
         }},
        {"sphere", [](ParseState& p, std::stringstream& s) {
             float f[4];
             if (!readValues(s, 4, f)) return;
             mat4 t = p.stack->back();
             t = mat4_mul(t, mat4_translate(mk3(f[0], f[1], f[2])));
             t = mat4_mul(t, mat4_scale(mk3(f[3])));
             PtcSphere sp{};
             for (int i = 0; i < 16; ++i) sp.trans[i] = t.m[i];
             sp.mv = p.mv;
             sp.objType = PTC_OBJECT;
             p.scene->spheres.push_back(sp);
         }},
        {"maxverts", [](ParseState&, std::stringstream& s) { float f[1]; readValues(s, 1, f); }},
        {"vertex", [](ParseState& p, std::stringstream& s) {
             float f[3];
             if (readValues(s, 3, f)) p.scene->vertices.push_back(PtcFloat3{f[0], f[1], f[2]});
         }},
        {"tri", [](ParseState& p, std::stringstream& s) {
             int idx[3];
             if (!readValues(s, 3, idx)) return;
             const mat4& top = p.stack->back();
             vec3 v[3];
             for (int k = 0; k < 3; ++k) {
                 // points are transformed at parse time: (M * (v,1)).xyz / w
                 vec4 h = mat4_mul(top, mk4(V3(p.scene->vertices.at((size_t)idx[k])), 1.f));
                 v[k] = xyz(h) / h.w;
             }
             PtcTriangle tri{};
             tri.v1 = P3(v[0]); tri.v2 = P3(v[1]); tri.v3 = P3(v[2]);
             tri.normal = P3(normalize(cross(v[1] - v[0], v[2] - v[0])));
             tri.mv = p.mv;
             tri.objType = PTC_OBJECT;
             p.scene->triangles.push_back(tri);
         }},
        {"translate", [](ParseState& p, std::stringstream& s) {
             float f[3];
             if (readValues(s, 3, f)) p.stack->back() = mat4_mul(p.stack->back(), mat4_translate(mk3(f[0], f[1], f[2])));
         }},
        {"scale", [](ParseState& p, std::stringstream& s) {
             float f[3];
             if (readValues(s, 3, f)) p.stack->back() = mat4_mul(p.stack->back(), mat4_scale(mk3(f[0], f[1], f[2])));
         }},
        {"rotate", [](ParseState& p, std::stringstream& s) {
             float f[4];
             if (!readValues(s, 4, f)) return;
             float radians = f[3] * PT_PIf / 180.f;
             p.stack->back() = mat4_mul(p.stack->back(), mat4_rotate(radians, mk3(f[0], f[1], f[2])));
         }},
        {"pushTransform", [](ParseState& p, std::stringstream&) { p.stack->push_back(p.stack->back()); }},
        {"popTransform", [](ParseState& p, std::stringstream&) {
             if (p.stack->size() <= 1) std::cerr << "Stack has no elements. Cannot pop" << std::endl;
             else p.stack->pop_back();
         }},
        {"attenuation", [](ParseState& p, std::stringstream& s) {
             float f[3];
             if (readValues(s, 3, f)) p.attenuation = mk3(f[0], f[1], f[2]);
         }},
        {"ambient", [](ParseState& p, std::stringstream& s) { setVec(s, p.mv.ambient); }},
        {"diffuse", [](ParseState& p, std::stringstream& s) { setVec(s, p.mv.diffuse); }},
        {"specular", [](ParseState& p, std::stringstream& s) { setVec(s, p.mv.specular); }},
        {"emission", [](ParseState& p, std::stringstream& s) { setVec(s, p.mv.emission); }},
        {"shininess", [](ParseState& p, std::stringstream& s) { float f[1]; if (readValues(s, 1, f)) p.mv.shininess = f[0]; }},
        {"roughness", [](ParseState& p, std::stringstream& s) { float f[1]; if (readValues(s, 1, f)) p.mv.roughness = f[0]; }},
        {"directional", [](ParseState& p, std::stringstream& s) {
             float f[6];
             if (!readValues(s, 6, f)) return;
             PtcDirectionalLight l;
             l.direction = P3(normalize(mk3(f[0], f[1], f[2])));
             l.color = PtcFloat3{f[3], f[4], f[5]};
             p.scene->dlights.push_back(l);
         }},
        {"point", [](ParseState& p, std::stringstream& s) {
             float f[6];
             if (!readValues(s, 6, f)) return;
             PtcPointLight l;
             l.location = PtcFloat3{f[0], f[1], f[2]};
             l.color = PtcFloat3{f[3], f[4], f[5]};
             l.attenuation = P3(p.attenuation);
             p.scene->plights.push_back(l);
         }},
        {"quadLight", [](ParseState& p, std::stringstream& s) {
             float f[12];
             if (!readValues(s, 12, f)) return;
             // a, ab, ac, intensity; the quad is NOT transformed by the stack and
             // its two LIGHT triangles carry the unnormalised normal ab x ac.
             vec3 a = mk3(f[0], f[1], f[2]), ab = mk3(f[3], f[4], f[5]), ac = mk3(f[6], f[7], f[8]);
             PtcMaterialValue lmv = p.mv;
             lmv.emission = PtcFloat3{f[9], f[10], f[11]};
             PtcTriangle t1{}, t2{};
             t1.v1 = P3(a); t1.v2 = P3(a + ab); t1.v3 = P3(a + ac);
             t1.normal = P3(cross(ab, ac));
             t1.mv = lmv; t1.objType = PTC_LIGHT;
             t2.v1 = P3(a + ab); t2.v2 = P3(a + ab + ac); t2.v3 = P3(a + ac);
             t2.normal = t1.normal;
             t2.mv = lmv; t2.objType = PTC_LIGHT;
             p.scene->triangles.push_back(t1);
             p.scene->triangles.push_back(t2);
             PtcQuadLight q;
             q.tri1 = t1; q.tri2 = t2; q.color = lmv.emission;
             p.scene->qlights.push_back(q);
         }},
        {"lightsamples", [](ParseState& p, std::stringstream& s) {
             float f[1];
             if (readValues(s, 1, f)) p.scene->light_samples = toUnsigned(f[0]);
         }},
        {"lightstratify", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (readValues(s, 1, v)) p.scene->light_stratify = isOn(v[0]) ? 1u : 0u;
         }},
        {"spp", [](ParseState& p, std::stringstream& s) {
             int v[1];
             if (readValues(s, 1, v)) p.scene->samples_per_pixel = (unsigned int)v[0];
         }},
        {"nexteventestimation", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (!readValues(s, 1, v)) return;
             p.scene->config.next_event_est = isOn(v[0]) ? PTC_NEE_ON : v[0].compare("mis") == 0 ? PTC_NEE_MIS : PTC_NEE_OFF;
         }},
        {"russianroulette", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (readValues(s, 1, v)) p.scene->config.russian_roul = isOn(v[0]) ? 1u : 0u;
         }},
        {"importancesampling", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (!readValues(s, 1, v)) return;
             if (v[0].compare("cosine") == 0) p.scene->sampling_method = PTC_COSINE_SAMPLING;
             else if (v[0].compare("brdf") == 0) p.scene->sampling_method = PTC_BRDF_SAMPLING;
             // anything else leaves the current method in place
         }},
        {"brdf", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (readValues(s, 1, v)) p.mv.brdf_type = v[0].compare("ggx") == 0 ? PTC_GGX : PTC_MOD_PHONG;
         }},
        {"gamma", [](ParseState& p, std::stringstream& s) { float f[1]; if (readValues(s, 1, f)) p.scene->config.gamma = f[0]; }},
        {"ior", [](ParseState& p, std::stringstream& s) { float f[1]; if (readValues(s, 1, f)) p.mv.ior = f[0]; }},
        {"materialtype", [](ParseState& p, std::stringstream& s) {
             std::string v[1];
             if (!readValues(s, 1, v)) return;
             p.mv.matType = v[0].compare("glass") == 0 ? PTC_GLASS : v[0].compare("volumetric") == 0 ? PTC_VOLUMETRIC : PTC_DIFFUSE;
         }},
        {"scattercoeff", [](ParseState& p, std::stringstream& s) {
             float f[3];
             if (!readValues(s, 3, f)) return;
             p.mv.sigma_a = f[0];
             p.mv.sigma_s = f[1];
             p.mv.sigma_t = p.mv.sigma_a + p.mv.sigma_s;
             p.mv.g = f[2];
         }},
    };
    return table;
}

}  // namespace

Scene::Scene() : config(defaultConfig()) {}

std::shared_ptr<Scene> SceneLoader::load(std::string sceneFilename) {
    std::ifstream in(sceneFilename);
    if (!in.is_open()) throw std::runtime_error("Unable to open scene file " + sceneFilename);
    return loadStream(in);
}

std::shared_ptr<Scene> SceneLoader::loadStream(std::istream& in) {
    auto scene = std::make_shared<Scene>();

    ParseState st;
    st.scene = scene.get();
    // material defaults, OptiXRenderer/SceneLoader.cpp:49-64
    st.mv = PtcMaterialValue{};
    st.mv.shininess = 1.0f;
    st.mv.roughness = 1.0f;
    st.mv.brdf_type = PTC_MOD_PHONG;
    st.mv.ior = 1.0f;
    st.mv.matType = PTC_DIFFUSE;
    st.mv.sigma_a = st.mv.sigma_s = st.mv.sigma_t = 1.0f;
    st.mv.g = 5.0f;
    st.attenuation = mk3(1, 0, 0);
    transStack.clear();
    transStack.push_back(mat4_identity());
    st.stack = &transStack;

    const auto& table = handlers();
    std::string line, cmd;
    while (std::getline(in, line)) {
        if (line.find_first_not_of(" \t\r\n") == std::string::npos || line[0] == '#') continue;
        std::stringstream s(line);
        s >> cmd;
        auto it = table.find(cmd);
        if (it != table.end()) it->second(st, s);
    }
    return scene;
}
