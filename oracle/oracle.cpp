// oracle.cpp -- CPU oracle for the per-pixel path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED at the OptiX-SDK
// level; pinned against the reference's recorded images/sphere.png by the
// test-suite.
//
// Structure mirrors the reference: one function per OptiX program, a `trace`
// that stands in for rtTrace over the closed Trbvh traversal, and the per-pixel
// double loop of the ray-generation program.  Must be compiled with
// -ffp-contract=off so that +,-,*,/,sqrt are the plain IEEE-754 operations in
// the order written.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- vectors --
// optixu_math_namespace.h semantics: v/s multiplies by the reciprocal,
// normalize multiplies by 1/sqrt(dot), dot sums left to right.
struct F3 {
    float x, y, z;
};
struct F4 {
    float x, y, z, w;
};
inline F3 f3(float x, float y, float z) { return F3{x, y, z}; }
inline F3 f3(float s) { return F3{s, s, s}; }
inline F3 f3(const PtcFloat3& p) { return F3{p.x, p.y, p.z}; }
inline F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
inline F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
inline F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
inline F3 operator/(F3 a, float s) {
    float inv = 1.0f / s;
    return a * inv;
}
inline void operator+=(F3& a, F3 b) { a = a + b; }
inline void operator*=(F3& a, F3 b) { a = a * b; }
inline void operator*=(F3& a, float s) { a = a * s; }
inline float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(F3 v) { return sqrtf(dot(v, v)); }
inline F3 normalize(F3 v) {
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
inline F3 reflect(F3 i, F3 n) { return i - 2.0f * n * dot(n, i); }
inline F3 lerp(F3 a, F3 b, float t) { return a + t * (b - a); }
// OptiX SDK 6.x optixu_math_namespace.h `refract(r, i, n, ior)` (third-party header, not in
// the reference tree; restated from its published definition): n is flipped and ior used
// as eta when i leaves through n, 1/ior otherwise; false = total internal reflection.
inline bool refract(F3& r, F3 i, F3 n, float ior) {
    F3 nn = n;
    float negNdotV = dot(i, nn);
    float eta;
    if (negNdotV > 0.0f) {
        eta = ior;
        nn = -n;
        negNdotV = -negNdotV;
    } else {
        eta = 1.f / ior;
    }
    const float k = 1.f - eta * eta * (1.f - negNdotV * negNdotV);
    if (k < 0.0f) {
        r = f3(0.f);
        return false;
    }
    r = normalize(eta * i - (eta * negNdotV + sqrtf(k)) * nn);
    return true;
}
// same header: fresnel_schlick(cos_theta, exponent = 5, minimum = 0, maximum = 1)
inline float fresnel_schlick(float cos_theta) {
    const float v = 0.0f + (1.0f - 0.0f) * powf(fmaxf(0.0f, 1.0f - cos_theta), 5.0f);
    return fminf(fmaxf(v, 0.0f), 1.0f);
}

const float PIf = 3.14159265358979323846f;
const float RAY_MAX = 1e27f;  // RT_DEFAULT_MAX

// ---------------------------------------------------------------- matrices --
// Row-major 4x4.  inverse = adjugate * (1/det), cofactor expansion.
struct M4 {
    float m[16];
};
inline F4 mul(const M4& a, F4 v) {
    F4 r;
    r.x = a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z + a.m[3] * v.w;
    r.y = a.m[4] * v.x + a.m[5] * v.y + a.m[6] * v.z + a.m[7] * v.w;
    r.z = a.m[8] * v.x + a.m[9] * v.y + a.m[10] * v.z + a.m[11] * v.w;
    r.w = a.m[12] * v.x + a.m[13] * v.y + a.m[14] * v.z + a.m[15] * v.w;
    return r;
}
inline M4 transpose(const M4& a) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i * 4 + j] = a.m[j * 4 + i];
    return r;
}
// signed 3x3 cofactor terms written as m[a]*(m[b]*m[c]-m[d]*m[e]) + ... so the
// rounding sequence is explicit
M4 inverse(const M4& a) {
    const float* m = a.m;
    float c[16];
    c[0] = m[5] * (m[10] * m[15] - m[14] * m[11]) + m[9] * (m[14] * m[7] - m[6] * m[15]) + m[13] * (m[6] * m[11] - m[10] * m[7]);
    c[1] = m[1] * (m[14] * m[11] - m[10] * m[15]) + m[9] * (m[2] * m[15] - m[14] * m[3]) + m[13] * (m[10] * m[3] - m[2] * m[11]);
    c[2] = m[1] * (m[6] * m[15] - m[14] * m[7]) + m[5] * (m[14] * m[3] - m[2] * m[15]) + m[13] * (m[2] * m[7] - m[6] * m[3]);
    c[3] = m[1] * (m[10] * m[7] - m[6] * m[11]) + m[5] * (m[2] * m[11] - m[10] * m[3]) + m[9] * (m[6] * m[3] - m[2] * m[7]);
    c[4] = m[4] * (m[14] * m[11] - m[10] * m[15]) + m[8] * (m[6] * m[15] - m[14] * m[7]) + m[12] * (m[10] * m[7] - m[6] * m[11]);
    c[5] = m[0] * (m[10] * m[15] - m[14] * m[11]) + m[8] * (m[14] * m[3] - m[2] * m[15]) + m[12] * (m[2] * m[11] - m[10] * m[3]);
    c[6] = m[0] * (m[14] * m[7] - m[6] * m[15]) + m[4] * (m[2] * m[15] - m[14] * m[3]) + m[12] * (m[6] * m[3] - m[2] * m[7]);
    c[7] = m[0] * (m[6] * m[11] - m[10] * m[7]) + m[4] * (m[10] * m[3] - m[2] * m[11]) + m[8] * (m[2] * m[7] - m[6] * m[3]);
    c[8] = m[4] * (m[9] * m[15] - m[13] * m[11]) + m[8] * (m[13] * m[7] - m[5] * m[15]) + m[12] * (m[5] * m[11] - m[9] * m[7]);
    c[9] = m[0] * (m[13] * m[11] - m[9] * m[15]) + m[8] * (m[1] * m[15] - m[13] * m[3]) + m[12] * (m[9] * m[3] - m[1] * m[11]);
    c[10] = m[0] * (m[5] * m[15] - m[13] * m[7]) + m[4] * (m[13] * m[3] - m[1] * m[15]) + m[12] * (m[1] * m[7] - m[5] * m[3]);
    c[11] = m[0] * (m[9] * m[7] - m[5] * m[11]) + m[4] * (m[1] * m[11] - m[9] * m[3]) + m[8] * (m[5] * m[3] - m[1] * m[7]);
    c[12] = m[4] * (m[13] * m[10] - m[9] * m[14]) + m[8] * (m[5] * m[14] - m[13] * m[6]) + m[12] * (m[9] * m[6] - m[5] * m[10]);
    c[13] = m[0] * (m[9] * m[14] - m[13] * m[10]) + m[8] * (m[13] * m[2] - m[1] * m[14]) + m[12] * (m[1] * m[10] - m[9] * m[2]);
    c[14] = m[0] * (m[13] * m[6] - m[5] * m[14]) + m[4] * (m[1] * m[14] - m[13] * m[2]) + m[12] * (m[5] * m[2] - m[1] * m[6]);
    c[15] = m[0] * (m[5] * m[10] - m[9] * m[6]) + m[4] * (m[9] * m[2] - m[1] * m[10]) + m[8] * (m[1] * m[6] - m[5] * m[2]);
    const float det = m[0] * c[0] + m[4] * c[1] + m[8] * c[2] + m[12] * c[3];
    const float d = 1.0f / det;
    M4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = d * c[i];
    return r;
}

// -------------------------------------------------------------------- RNG --
// random.h:31-46
uint32_t tea16(uint32_t val0, uint32_t val1) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (unsigned n = 0; n < 16; n++) {
        s0 += 0x9e3779b9;
        v0 += ((v1 << 4) + 0xa341316c) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4);
        v1 += ((v0 << 4) + 0xad90777d) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761e);
    }
    return v0;
}
// random.h:49-55
uint32_t lcg(uint32_t& prev) {
    prev = (1664525u * prev + 1013904223u);
    return prev & 0x00FFFFFF;
}
// random.h:64-67
float rnd(uint32_t& prev) { return ((float)lcg(prev) / (float)0x01000000); }

// ------------------------------------------------------------ scene access --
struct Ray {
    F3 origin, direction;
    float tmin, tmax;
};

// Geometries.h:40-45
struct Attributes {
    F3 intersection, normal, wo, prev_intersection;
    PtcMaterialValue mv;
    int objType;
};

// Payloads.h:10-15
struct Payload {
    F3 radiance, throughput, origin, dir;
    uint32_t depth, seed;
    bool done;
};

struct Candidate {
    bool hit;
    float t, u, v;
};

// Triangle.cu:12-46 up to the rtPotentialIntersection call: the distance the
// program would report, or no candidate.
inline Candidate triangleCandidate(const PtcTriangle& tri, const Ray& ray) {
    Candidate c{false, 0.f, 0.f, 0.f};
    F3 n = f3(tri.normal);
    if (dot(ray.direction, n) >= .0f) n = -n;
    float nDotWo = dot(n, -ray.direction);
    if (nDotWo == 0.0f) return c;
    F3 v1 = f3(tri.v1), v2 = f3(tri.v2), v3 = f3(tri.v3);
    float t = dot(v1 - ray.origin, n) / dot(ray.direction, n);
    F3 P = ray.origin + t * ray.direction;
    if ((double)t < 0.001) return c;
    F3 tmp0 = v3 - v1, tmp1 = v2 - v1, tmp2 = P - v1;
    float d00 = dot(tmp0, tmp0), d01 = dot(tmp0, tmp1), d02 = dot(tmp0, tmp2);
    float d11 = dot(tmp1, tmp1), d12 = dot(tmp1, tmp2);
    float denom = d00 * d11 - d01 * d01;
    float u = (d11 * d02 - d01 * d12) / denom;
    float v = (d00 * d12 - d01 * d02) / denom;
    if (0 > u || u > 1 || 0 > v || v > 1 || u + v > 1) return c;
    c.hit = true;
    c.t = t;
    c.u = u;
    c.v = v;
    return c;
}

// Triangle.cu:48-58: attributes of an accepted hit.
inline void triangleAttributes(const PtcTriangle& tri, const Ray& ray, float t, Attributes& a) {
    F3 n = f3(tri.normal);
    if (dot(ray.direction, n) >= .0f) n = -n;
    float nDotWo = dot(n, -ray.direction);
    a.intersection = ray.origin + t * ray.direction;
    a.wo = -ray.direction;
    a.normal = nDotWo > 0 ? n : -n;
    a.mv = tri.mv;
    a.prev_intersection = ray.origin;
    a.objType = tri.objType;
}

// Sphere.cu:13-42.  `epsilon` (Sphere.cu:9) is never set by the host; taken as 0.
inline Candidate sphereCandidate(const PtcSphere& sp, const Ray& ray, F3* Pworld) {
    Candidate c{false, 0.f, 0.f, 0.f};
    M4 trans;
    std::memcpy(trans.m, sp.trans, sizeof(trans.m));
    M4 itrans = inverse(trans);
    F4 oh = mul(itrans, F4{ray.origin.x, ray.origin.y, ray.origin.z, 1.f});
    F3 rayOri = f3(oh.x, oh.y, oh.z) / oh.w;
    F4 dh = mul(itrans, F4{ray.direction.x, ray.direction.y, ray.direction.z, 0.f});
    F3 rayDir = normalize(f3(dh.x, dh.y, dh.z));
    float t = 0;
    float b = dot(rayDir, rayOri);
    float cc = dot(rayOri, rayOri);
    float r2 = 1.f;
    float disc = b * b - cc + r2;
    if (disc < 0) return c;
    if (disc == 0)
        t = -b;
    else if (cc > r2)
        t = -b - sqrtf(disc);
    else
        t = -b + sqrtf(disc);
    if (t < 0.0f) return c;
    F3 P = rayOri + t * rayDir;
    F4 ih = mul(trans, F4{P.x, P.y, P.z, 1.f});
    P = f3(ih.x, ih.y, ih.z) / ih.w;
    t = length(P - ray.origin);
    c.hit = true;
    c.t = t;
    if (Pworld) *Pworld = P;
    return c;
}

// Sphere.cu:45-56
inline void sphereAttributes(const PtcSphere& sp, const Ray& ray, Attributes& a) {
    F3 P;
    sphereCandidate(sp, ray, &P);
    M4 trans;
    std::memcpy(trans.m, sp.trans, sizeof(trans.m));
    M4 itrans = inverse(trans);
    a.intersection = P;
    a.wo = -ray.direction;
    F4 th = mul(itrans, F4{P.x, P.y, P.z, 1.f});
    F3 nobj = normalize(f3(th.x, th.y, th.z) / th.w);
    F4 nh = mul(transpose(itrans), F4{nobj.x, nobj.y, nobj.z, 0.f});
    a.normal = normalize(f3(nh.x, nh.y, nh.z));
    a.mv = sp.mv;
    a.prev_intersection = ray.origin;
    a.objType = PTC_OBJECT;
}

// --------------------------------------------------------------- CPU BVH ---
// Binned-SAH BVH over primitive boxes (Triangle.cu:62-72, Sphere.cu:60-73),
// used only to make large scenes affordable; results must equal brute force.
struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = INFINITY;
            hi[k] = -INFINITY;
        }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BNode {
    Box box;
    int left, right;   // children (internal)
    int first, count;  // leaf range into order[]
};

struct Bvh {
    std::vector<BNode> nodes;
    std::vector<int> order;
};

Box primBox(const OrcScene& s, int prim) {
    Box b;
    if ((uint64_t)prim < s.n_tris) {
        const PtcTriangle& t = s.tris[prim];
        b.lo[0] = fminf(fminf(t.v1.x, t.v2.x), t.v3.x);
        b.lo[1] = fminf(fminf(t.v1.y, t.v2.y), t.v3.y);
        b.lo[2] = fminf(fminf(t.v1.z, t.v2.z), t.v3.z);
        b.hi[0] = fmaxf(fmaxf(t.v1.x, t.v2.x), t.v3.x);
        b.hi[1] = fmaxf(fmaxf(t.v1.y, t.v2.y), t.v3.y);
        b.hi[2] = fmaxf(fmaxf(t.v1.z, t.v2.z), t.v3.z);
    } else {
        const PtcSphere& sp = s.spheres[prim - s.n_tris];
        const float* m = sp.trans;
        float x = length(f3(m[0], m[1], m[2])), y = length(f3(m[4], m[5], m[6])), z = length(f3(m[8], m[9], m[10]));
        b.lo[0] = m[3] - x; b.lo[1] = m[7] - y; b.lo[2] = m[11] - z;
        b.hi[0] = m[3] + x; b.hi[1] = m[7] + y; b.hi[2] = m[11] + z;
    }
    // a few ulps of slack: the plane-based triangle test can accept a point that
    // rounding puts marginally outside the exact box
    for (int k = 0; k < 3; ++k) {
        float pad = 1e-5f * std::max(1.0f, std::max(fabsf(b.lo[k]), fabsf(b.hi[k])));
        b.lo[k] -= pad;
        b.hi[k] += pad;
    }
    return b;
}

int buildRec(Bvh& bvh, const std::vector<Box>& boxes, int first, int count) {
    int id = (int)bvh.nodes.size();
    bvh.nodes.push_back(BNode());
    Box nb, cb;
    nb.reset();
    cb.reset();
    for (int i = first; i < first + count; ++i) {
        const Box& b = boxes[bvh.order[i]];
        nb.grow(b);
        Box c;
        for (int k = 0; k < 3; ++k) c.lo[k] = c.hi[k] = 0.5f * (b.lo[k] + b.hi[k]);
        cb.grow(c);
    }
    bvh.nodes[id].box = nb;
    bvh.nodes[id].left = bvh.nodes[id].right = -1;
    bvh.nodes[id].first = first;
    bvh.nodes[id].count = count;
    if (count <= 2) return id;
    const int NB = 16;
    float bestCost = INFINITY;
    int bestAxis = -1, bestSplit = -1;
    for (int ax = 0; ax < 3; ++ax) {
        float lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
        if (!(ext > 0)) continue;
        Box bb[NB];
        int cnt[NB] = {0};
        for (int i = 0; i < NB; ++i) bb[i].reset();
        for (int i = first; i < first + count; ++i) {
            const Box& b = boxes[bvh.order[i]];
            int bi = std::min(NB - 1, (int)(NB * ((0.5f * (b.lo[ax] + b.hi[ax]) - lo) / ext)));
            bb[bi].grow(b);
            cnt[bi]++;
        }
        float rightArea[NB];
        Box acc;
        acc.reset();
        for (int i = NB - 1; i > 0; --i) {
            acc.grow(bb[i]);
            rightArea[i] = acc.area();
        }
        acc.reset();
        int nl = 0;
        for (int i = 0; i < NB - 1; ++i) {
            acc.grow(bb[i]);
            nl += cnt[i];
            int nr = count - nl;
            if (nl == 0 || nr == 0) continue;
            float cost = acc.area() * nl + rightArea[i + 1] * nr;
            if (cost < bestCost) {
                bestCost = cost;
                bestAxis = ax;
                bestSplit = i;
            }
        }
    }
    int mid;
    if (bestAxis < 0) {
        if (count <= 8) return id;
        mid = first + count / 2;
    } else {
        if (count <= 4 && bestCost >= nb.area() * count) return id;
        float lo = cb.lo[bestAxis], ext = cb.hi[bestAxis] - cb.lo[bestAxis];
        int* beg = bvh.order.data() + first;
        int* p = std::partition(beg, beg + count, [&](int prim) {
            const Box& b = boxes[prim];
            int bi = std::min(NB - 1, (int)(NB * ((0.5f * (b.lo[bestAxis] + b.hi[bestAxis]) - lo) / ext)));
            return bi <= bestSplit;
        });
        mid = (int)(p - bvh.order.data());
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    int l = buildRec(bvh, boxes, first, mid - first);
    int r = buildRec(bvh, boxes, mid, first + count - mid);
    bvh.nodes[id].left = l;
    bvh.nodes[id].right = r;
    bvh.nodes[id].count = 0;
    return id;
}

// slab test; returns entry distance or INFINITY when the box is missed within [tmin,tmax]
inline float boxEntry(const Box& b, const Ray& r, const float inv[3], float tmax) {
    const float o[3] = {r.origin.x, r.origin.y, r.origin.z};
    float t0 = r.tmin, t1 = tmax;
    for (int k = 0; k < 3; ++k) {
        float a = (b.lo[k] - o[k]) * inv[k], c = (b.hi[k] - o[k]) * inv[k];
        float n = fminf(a, c), f = fmaxf(a, c);  // NaN-ignoring min/max
        t0 = fmaxf(t0, n);
        t1 = fminf(t1, f);
    }
    // widen: slab arithmetic may be off by a few ulps
    return t0 <= t1 * 1.00001f + 1e-6f ? t0 : INFINITY;
}

struct World {
    const OrcScene* s;
    const OrcOptions* o;
    Bvh bvh;
    bool useBvh;
    uint32_t maxDepthEff;
};

// Stand-in for rtTrace with ray type 0: closest accepted candidate.  A candidate
// is accepted like rtPotentialIntersection does it: tmin < t < current tmax;
// among equal distances the lowest primitive index wins (triangles before
// spheres), which makes the result independent of the traversal order.
struct Closest {
    int prim;
    float t, u, v;
};

inline void testPrim(const World& w, const Ray& ray, int prim, Closest& best) {
    Candidate c;
    if ((uint64_t)prim < w.s->n_tris)
        c = triangleCandidate(w.s->tris[prim], ray);
    else
        c = sphereCandidate(w.s->spheres[prim - w.s->n_tris], ray, nullptr);
    if (!c.hit) return;
    if (!(c.t > ray.tmin && c.t < ray.tmax)) return;
    if (best.prim < 0 || c.t < best.t || (c.t == best.t && prim < best.prim)) {
        best.prim = prim;
        best.t = c.t;
        best.u = c.u;
        best.v = c.v;
    }
}

Closest traceClosest(const World& w, const Ray& ray) {
    Closest best{-1, 0.f, 0.f, 0.f};
    const int nprim = (int)(w.s->n_tris + w.s->n_spheres);
    if (!w.useBvh) {
        for (int p = 0; p < nprim; ++p) testPrim(w, ray, p, best);
        return best;
    }
    if (nprim == 0) return best;
    const float inv[3] = {1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z};
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BNode& n = w.bvh.nodes[stack[--sp]];
        float lim = best.prim < 0 ? ray.tmax : best.t;
        if (boxEntry(n.box, ray, inv, lim) == INFINITY) continue;
        if (n.left < 0) {
            for (int i = n.first; i < n.first + n.count; ++i) testPrim(w, ray, w.bvh.order[i], best);
        } else {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    return best;
}

// Stand-in for rtTrace with ray type 1 (shadow): Common.cu:30-36, any accepted
// candidate makes the ray occluded.
bool traceOccluded(const World& w, const Ray& ray) {
    const int nprim = (int)(w.s->n_tris + w.s->n_spheres);
    auto any = [&](int prim) {
        Candidate c;
        if ((uint64_t)prim < w.s->n_tris)
            c = triangleCandidate(w.s->tris[prim], ray);
        else
            c = sphereCandidate(w.s->spheres[prim - w.s->n_tris], ray, nullptr);
        return c.hit && c.t > ray.tmin && c.t < ray.tmax;
    };
    if (!w.useBvh) {
        for (int p = 0; p < nprim; ++p)
            if (any(p)) return true;
        return false;
    }
    if (nprim == 0) return false;
    const float inv[3] = {1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z};
    int stack[128], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BNode& n = w.bvh.nodes[stack[--sp]];
        if (boxEntry(n.box, ray, inv, ray.tmax) == INFINITY) continue;
        if (n.left < 0) {
            for (int i = n.first; i < n.first + n.count; ++i)
                if (any(w.bvh.order[i])) return true;
        } else {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    return false;
}

struct Counters {
    uint64_t paths = 0, ext = 0, shadow = 0;
};

struct Ctx {
    const World* w;
    Counters* cnt;
};

inline bool shadowVisible(const Ctx& c, F3 origin, F3 dir, float tmin, float tmax) {
    Ray r{origin, dir, tmin, tmax};
    c.cnt->shadow++;
    return !traceOccluded(*c.w, r);
}

// ------------------------------------------------------------ integrators --

// RayTracer.cu:259-263
inline F3 phongBRDF(F3 wi, F3 /*wo*/, F3 reflect_vec, float shininess, F3 diffuse, F3 specular) {
    return (diffuse / PIf) + (specular * ((shininess + 2.0f) / (2.0f * PIf)) * powf(fmaxf(dot(reflect_vec, wi), .0f), shininess));
}

// RayTracer.cu:265-285
inline F3 ggxBRDF(F3 wi, F3 wo, F3 n, float roughness, F3 specular) {
    float wi_dot_n_dir = dot(wi, n);
    float wo_dot_n_dir = dot(wo, n);
    float alpha = roughness;
    F3 K_s = specular;
    F3 h = normalize(wi + wo);
    float theta_h = acosf(dot(h, n));
    float cos_theta_h_4 = powf(cosf(theta_h), 4.0f);
    float alpha_tan_theta_h_sq = (alpha * alpha) + powf(tanf(theta_h), 2.0f);
    float D = (alpha * alpha) / (PIf * cos_theta_h_4 * powf(alpha_tan_theta_h_sq, 2.0f));
    float G_1_wi = (wi_dot_n_dir > .0f) ? 2.0f / (1.0f + sqrtf(1.0f + (alpha * alpha) * powf(tanf(acosf(dot(wi, n))), 2.0f))) : .0f;
    float G_1_wo = (wo_dot_n_dir > .0f) ? 2.0f / (1.0f + sqrtf(1.0f + (alpha * alpha) * powf(tanf(acosf(dot(wo, n))), 2.0f))) : .0f;
    float G = G_1_wi * G_1_wo;
    F3 F = K_s + (f3(1.0f) - K_s) * powf(1.0f - dot(wi, h), 5.0f);
    return (F * G * D) / (4.0f * wi_dot_n_dir * wo_dot_n_dir);
}

// RayTracer.cu:287-294
inline float ggxPDF(F3 wi, F3 /*wo*/, F3 n, F3 half_angle_vec, float t_val, float roughness) {
    float alpha = roughness;
    float theta_h = acosf(dot(half_angle_vec, n));
    float alpha_tan_theta_h_sq = (alpha * alpha) + powf(tanf(theta_h), 2.0f);
    float D = (alpha * alpha) / (PIf * powf(cosf(theta_h), 4.0f) * powf(alpha_tan_theta_h_sq, 2.0f));
    return ((1.0f - t_val) * fmaxf(dot(n, wi), .0f) / PIf) + ((t_val * D * dot(n, half_angle_vec)) / (4.0f * dot(half_angle_vec, wi)));
}

// RayTracer.cu:232-257.  Under literal semantics phi/theta are by-value
// parameters there: the caller's copies keep their old values.
inline void computePolarAngles(int semantics, uint32_t sampling_method, float& phi, float& theta, const float zeta[3], float t, float shininess) {
    float p = phi, th = theta;
    switch (sampling_method) {
        case PTC_HEMISPHERE_SAMPLING:
            p = 2.0f * PIf * zeta[2];
            th = acosf(zeta[1]);
            break;
        case PTC_COSINE_SAMPLING:
            p = 2.0f * PIf * zeta[2];
            th = acosf(sqrtf(zeta[1]));
            break;
        case PTC_BRDF_SAMPLING:
            p = 2.0f * PIf * zeta[2];
            if (zeta[0] > t)
                th = acosf(sqrtf(zeta[1]));
            else
                th = acosf(powf(zeta[1], (1.0f / (shininess + 1.0f))));
            break;
        default:
            break;
    }
    if (semantics == 1) {
        phi = p;
        theta = th;
    }
}

// RayTracer.cu:302-312.  The test reads v (always the zero vector at every call
// site), so a = (1,0,0) is always chosen.  Literal: u, v by value -> unchanged.
inline void genCoordFrame(int semantics, F3 w, F3& u, F3& v) {
    F3 a = f3(.0f, 1.0f, .0f);
    if (1.0f - fabsf(dot(a, v)) <= 1.0f) a = f3(1.0f, .0f, .0f);
    F3 uu = normalize(cross(a, w));
    F3 vv = normalize(cross(w, uu));
    if (semantics == 1) {
        u = uu;
        v = vv;
    }
}

// RayTracer.cu:316-322
inline F3 sphericalDir(float theta, float phi, F3 u, F3 v, F3 w) {
    F3 s = f3(cosf(phi) * sinf(theta), sinf(phi) * sinf(theta), cosf(theta));
    return normalize((s.x * u + s.y * v + s.z * w));
}

struct Shared {
    uint32_t light_samples, light_stratify, sampling_method;
};

// RayTracer.cu:34-105 ("raytracer")
void closestHit(const Ctx& c, const Attributes& attrib, Payload& payload) {
    const OrcScene& s = *c.w->s;
    const PtcMaterialValue& mv = attrib.mv;
    const PtcConfig& cf = s.config;
    F3 result = f3(mv.ambient) + f3(mv.emission);
    for (uint64_t i = 0; i < s.n_plights; i++) {
        F3 loc = f3(s.plights[i].location);
        F3 lightDir = normalize(loc - attrib.intersection);
        float lightDist = length(loc - attrib.intersection);
        if (shadowVisible(c, attrib.intersection + lightDir * cf.epsilon, lightDir, cf.epsilon, lightDist)) {
            F3 H = normalize(lightDir + attrib.wo);
            float att = dot(f3(s.plights[i].attenuation), f3(1, lightDist, lightDist * lightDist));
            F3 I = f3(mv.diffuse) * fmaxf(dot(attrib.normal, lightDir), 0);
            I += f3(mv.specular) * powf(fmaxf(dot(attrib.normal, H), 0), mv.shininess);
            I *= f3(s.plights[i].color) / att;
            result += I;
        }
    }
    for (uint64_t i = 0; i < s.n_dlights; i++) {
        F3 lightDir = f3(s.dlights[i].direction);
        float lightDist = RAY_MAX;
        if (shadowVisible(c, attrib.intersection + lightDir * cf.epsilon, lightDir, cf.epsilon, lightDist)) {
            F3 H = normalize(lightDir + attrib.wo);
            F3 I = f3(mv.diffuse) * fmaxf(dot(attrib.normal, lightDir), 0);
            I += f3(mv.specular) * powf(fmaxf(dot(attrib.normal, H), 0), mv.shininess);
            I *= f3(s.dlights[i].color);
            result += I;
        }
    }
    payload.radiance = result * payload.throughput;
    if (length(f3(mv.specular)) > 0) {
        payload.origin = attrib.intersection;
        payload.dir = reflect(-attrib.wo, attrib.normal);
        payload.depth++;
        payload.throughput *= f3(mv.specular);
    } else {
        payload.done = true;
    }
}

// RayTracer.cu:107-151
void analyticDirect(const Ctx& c, const Attributes& attrib, Payload& payload) {
    const OrcScene& s = *c.w->s;
    const PtcMaterialValue& mv = attrib.mv;
    F3 result = f3(mv.ambient) + f3(mv.emission);
    if (s.light_samples == 0) {
        for (uint64_t i = 0; i < s.n_qlights; ++i) {
            F3 f_brdf = f3(mv.diffuse) / PIf;
            F3 hitPt = attrib.intersection;
            F3 hitPtNormal = attrib.normal;
            const PtcQuadLight& q = s.qlights[i];
            F3 p1 = f3(q.tri1.v1), p2 = f3(q.tri1.v2), p3 = f3(q.tri2.v2), p4 = f3(q.tri1.v3);
            float theta_1 = acosf(dot(normalize(p1 - hitPt), normalize(p2 - hitPt)));
            float theta_2 = acosf(dot(normalize(p2 - hitPt), normalize(p3 - hitPt)));
            float theta_3 = acosf(dot(normalize(p3 - hitPt), normalize(p4 - hitPt)));
            float theta_4 = acosf(dot(normalize(p4 - hitPt), normalize(p1 - hitPt)));
            F3 gamma_1 = normalize(cross((p1 - hitPt), (p2 - hitPt)));
            F3 gamma_2 = normalize(cross((p2 - hitPt), (p3 - hitPt)));
            F3 gamma_3 = normalize(cross((p3 - hitPt), (p4 - hitPt)));
            F3 gamma_4 = normalize(cross((p4 - hitPt), (p1 - hitPt)));
            F3 irradiance_vec = 0.5f * ((theta_1 * gamma_1) + (theta_2 * gamma_2) + (theta_3 * gamma_3) + (theta_4 * gamma_4));
            F3 dir_radiance = f_brdf * f3(q.color) * dot(irradiance_vec, hitPtNormal);
            result += dir_radiance;
        }
    }
    payload.radiance = result;
    payload.done = true;
}

// RayTracer.cu:153-230
void direct(const Ctx& c, const Attributes& attrib, Payload& payload) {
    const OrcScene& s = *c.w->s;
    const PtcMaterialValue& mv = attrib.mv;
    const PtcConfig& cf = s.config;
    F3 result = f3(mv.ambient) + f3(mv.emission);
    if (attrib.objType == PTC_LIGHT) {
        payload.radiance = result;
        payload.depth = c.w->maxDepthEff;
        payload.done = true;
        return;
    }
    for (uint64_t k = 0; k < s.n_qlights; ++k) {
        F3 sampled_result = f3(.0f);
        const PtcQuadLight& q = s.qlights[k];
        F3 a = f3(q.tri1.v1), b = f3(q.tri1.v2), cc = f3(q.tri2.v3);
        F3 ac = cc - a, ab = b - a;
        float area = fabsf(length(cross(ab, ac)));
        int root_light_samples = (int)sqrtf((float)s.light_samples);
        for (int i = 0; i < root_light_samples; ++i) {
            for (int j = 0; j < root_light_samples; ++j) {
                float u1 = rnd(payload.seed);
                float u2 = rnd(payload.seed);
                F3 sampled_light_pos;
                if (s.light_stratify)
                    sampled_light_pos = a + ((j + u1) * (ab / (float)root_light_samples)) + ((i + u2) * (ac / (float)root_light_samples));
                else
                    sampled_light_pos = a + u1 * ab + u2 * ac;
                F3 shadow_ray_origin = attrib.intersection;
                F3 shadow_ray_dir = normalize(sampled_light_pos - shadow_ray_origin);
                float light_dist = length(sampled_light_pos - shadow_ray_origin);
                if (shadowVisible(c, shadow_ray_origin, shadow_ray_dir, cf.epsilon, light_dist - cf.epsilon)) {
                    F3 f_brdf = (f3(mv.diffuse) / PIf) +
                                (f3(mv.specular) * ((mv.shininess + 2.0f) / (2.0f * PIf)) *
                                 powf(fmaxf(dot(normalize(reflect(-attrib.wo, attrib.normal)), normalize(sampled_light_pos - shadow_ray_origin)), .0f), mv.shininess));
                    F3 x_prime = sampled_light_pos;
                    F3 x = shadow_ray_origin;
                    F3 n = attrib.normal;
                    F3 n_light = normalize(cross(ab, ac));
                    float R = length(x - x_prime);
                    float G = (1.0f / powf(R, 2.0f)) * fmaxf(dot(n, normalize(x_prime - x)), .0f) * (fmaxf(dot(n_light, normalize(x_prime - x)), .0f));
                    sampled_result += f_brdf * G;
                }
            }
        }
        result += f3(q.color) * sampled_result * (area / (float)s.light_samples);
    }
    payload.radiance = result;
    payload.done = true;
}

// RayTracer.cu:373-427
F3 computeDirectLight(const Ctx& c, const Attributes& attrib, Payload& payload) {
    const OrcScene& s = *c.w->s;
    const PtcConfig& cf = s.config;
    F3 L_d = f3(.0f);
    for (uint64_t k = 0; k < s.n_qlights; ++k) {
        int rt_light_samples = (int)sqrtf((float)s.light_samples);
        const PtcQuadLight& q = s.qlights[k];
        F3 a = f3(q.tri1.v1), b = f3(q.tri1.v2), cc = f3(q.tri1.v3);
        F3 ab = b - a, ac = cc - a;
        float area = fabsf(length(cross(ab, ac)));
        F3 sampled_radiance = f3(.0f);
        for (int i = 0; i < rt_light_samples; ++i) {
            for (int j = 0; j < rt_light_samples; ++j) {
                float u1 = rnd(payload.seed);
                float u2 = rnd(payload.seed);
                F3 light_sample_pos;
                if (s.light_stratify)
                    light_sample_pos = a + (j + u1) * (ab / (float)rt_light_samples) + (i + u2) * (ac / (float)rt_light_samples);
                else
                    light_sample_pos = a + ab * u1 + ac * u2;
                F3 light_sample_dir = normalize(light_sample_pos - attrib.intersection);
                float dist = length(light_sample_pos - attrib.intersection);
                F3 n_light = normalize(cross(ab, ac));
                if (shadowVisible(c, attrib.intersection, light_sample_dir, cf.epsilon, dist - cf.epsilon)) {
                    F3 reflect_dir = reflect(-attrib.wo, attrib.normal);
                    float G = (1.0f / (dist * dist)) * fmaxf(dot(attrib.normal, light_sample_dir), .0f) * fmaxf(dot(n_light, light_sample_dir), .0f);
                    F3 f_brdf = phongBRDF(light_sample_dir, attrib.wo, reflect_dir, attrib.mv.shininess, f3(attrib.mv.diffuse), f3(attrib.mv.specular));
                    sampled_radiance += f_brdf * G;
                }
            }
        }
        L_d += f3(q.color) * (area / (float)s.light_samples) * sampled_radiance;
    }
    return L_d;
}

// RayTracer.cu:429-494
void pathTracer2(const Ctx& c, const Attributes& attrib, Payload& payload) {
    F3 L_d = f3(.0f);
    F3 L_e = f3(attrib.mv.emission);
    if (payload.depth > 0 && attrib.objType == PTC_LIGHT) {
        payload.radiance = f3(.0f);
        payload.depth = c.w->maxDepthEff;
        payload.done = true;
        return;
    }
    L_d = computeDirectLight(c, attrib, payload);
    float zeta1 = rnd(payload.seed), zeta2(rnd(payload.seed));
    float theta = acosf(zeta1), phi = 2.0f * PIf * zeta2;
    F3 w = normalize(attrib.normal), a = f3(0, 1, 0);
    if (1.0f - fabsf(dot(a, w)) <= 1.0f) a = f3(1, 0, 0);
    F3 u = normalize(cross(a, w)), v = normalize(cross(w, u));
    F3 w_i = normalize(sphericalDir(theta, phi, u, v, w));
    F3 norm = normalize(attrib.normal);
    F3 incoming_ray_dir = normalize(-attrib.wo);
    F3 reflected_ray_dir = normalize(reflect(incoming_ray_dir, norm));
    F3 f_brdf = phongBRDF(w_i, normalize(attrib.wo), reflected_ray_dir, attrib.mv.shininess, f3(attrib.mv.diffuse), f3(attrib.mv.specular));
    float pdf = 1.0f / (2.0f * PIf);
    F3 attenuation = (f_brdf * fmaxf(dot(norm, w_i), .0f)) * (1.0f / pdf);
    if (payload.depth == 0) payload.radiance += L_e * payload.throughput;
    payload.radiance += L_d * payload.throughput;
    payload.throughput *= attenuation;
    float q = fminf(fmaxf(fmaxf(payload.throughput.x, payload.throughput.y), payload.throughput.z), 1.0f);
    if (rnd(payload.seed) > q) {
        payload.depth = c.w->maxDepthEff;
        payload.done = true;
        return;
    } else {
        payload.throughput *= 1.0f / (q);
    }
    payload.origin = attrib.intersection;
    payload.dir = w_i;
    payload.depth++;
}

// RayTracer.cu:327-332
inline float phaseHG(float wi_dot_wo, float g) {
    float numerator = (1.0f - (g * g));
    float denom = powf((1.0f + powf(g, 2.0f) + (2.0f * g * wi_dot_wo)), (3.0f / 2.0f));
    return (1.0f / (4.0f * PIf)) * (numerator / denom);
}

// RayTracer.cu:337-360.  `wi` is taken by value there (and the only caller passes a
// throw-away vector), so the sampled direction never leaves the function under either
// semantics; only the phase-function value does.
inline float samplePhaseHG(int semantics, F3 wo, F3 wi, float g, const float rando_samples[]) {
    float cos_theta = .0f;
    if (fabsf(g) < 0.001f) {
        cos_theta = 1.0f - (2.0f * rando_samples[0]);
    } else {
        cos_theta = (1.0f / (2.0f * g)) * (1.0f + powf(g, 2.0f) - powf((1 - g * g) / (1 - g + (2.0f * g * rando_samples[0])), 2.0f));
    }
    float sin_theta = sqrtf(fmaxf(0.f, 1.0f - (cos_theta * cos_theta)));
    (void)sin_theta;
    float phi = 2.0f * PIf * rando_samples[1];
    F3 v1 = f3(.0f), v2 = f3(.0f);
    genCoordFrame(semantics, wo, v1, v2);
    wi = sphericalDir(acosf(cos_theta), phi, v1, v2, wo);
    (void)wi;
    return phaseHG(-cos_theta, g);
}

// RayTracer.cu:362-365
inline float computeTransmittance(float sigma_t, float param_dist_t, float ray_length) { return expf(-sigma_t * fminf(param_dist_t, RAY_MAX) * ray_length); }

// RayTracer.cu:496-963, all three material types.
void pathTracer(const Ctx& c, const Attributes& attrib, Payload& payload) {
    const OrcScene& s = *c.w->s;
    const int sem = c.w->o->semantics;
    const PtcMaterialValue& mv = attrib.mv;
    const PtcConfig& cf = s.config;

    F3 result = f3(.0f);
    F3 L_d = f3(.0f);
    F3 n = normalize(attrib.normal);
    F3 L_e = f3(mv.emission);
    F3 K_s = f3(mv.specular);
    F3 K_d = f3(mv.diffuse);
    F3 r = normalize(reflect(-attrib.wo, attrib.normal));
    float theta = .0f;
    float phi = .0f;
    float exp_beta = 2.0f;
    F3 w = f3(.0f);
    F3 w_i = f3(.0f);
    F3 addon_throughput = f3(1.0f);
    F3 sampled_result = f3(1.0f);
    F3 new_pos = f3(.0f);  // :517
    float transmit = .0f;
    float pdf_NEE = .0f;

    if (cf.next_event_est && attrib.objType == PTC_LIGHT && payload.depth != 0) {  // :528-533
        payload.depth = c.w->maxDepthEff;
        payload.done = true;
        payload.radiance = result;
        return;
    }

    float zeta_0 = rnd(payload.seed);
    float zeta_1 = rnd(payload.seed);
    float zeta_2 = rnd(payload.seed);
    float K_s_avg = (K_s.x + K_s.y + K_s.z) / 3.0f;
    float K_d_avg = (K_d.x + K_d.y + K_d.z) / 3.0f;
    float t = .0f;
    if ((K_s_avg == .0f) && (K_d_avg == .0f)) {  // :546-553
        if (mv.brdf_type == PTC_MOD_PHONG)
            t = .0f;
        else if (mv.brdf_type == PTC_GGX)
            t = 1.0f;
    } else {
        float k_val = K_s_avg / (K_s_avg + K_d_avg);
        if (mv.brdf_type == PTC_MOD_PHONG)
            t = k_val;
        else
            t = fmaxf(.25f, k_val);
    }

    float rand_vals[3] = {zeta_0, zeta_1, zeta_2};
    computePolarAngles(sem, s.sampling_method, phi, theta, rand_vals, t, mv.shininess);  // :566

    if (mv.matType == PTC_GLASS) {  // :574-621
        F3 inc_ray_dir = -attrib.wo;
        F3 norm = n;
        float n1 = 1.0f;
        float n2 = mv.ior;
        float eta = .0f;
        float i_dot_n = .0f;
        float reflect_probability = .0f;
        F3 chosen_ray_dir = f3(.0f);
        if (dot(inc_ray_dir, norm) >= .0f) {
            // swapValue(n1, n2) takes its arguments by value (:296-300): no effect as committed
            if (sem == 1) {
                float tmp = n1;
                n1 = n2;
                n2 = tmp;
            }
            i_dot_n = dot(inc_ray_dir, norm);
        } else {
            i_dot_n = dot(inc_ray_dir, -norm);
        }
        eta = n2 / n1;
        if (refract(chosen_ray_dir, inc_ray_dir, norm, eta))
            reflect_probability = fresnel_schlick(i_dot_n);
        else
            reflect_probability = 1.0f;
        float rand_valz = rnd(payload.seed);  // :614, one more draw before the light samples
        if (rand_valz < reflect_probability)
            w_i = normalize(reflect(inc_ray_dir, norm));
        else
            w_i = normalize(chosen_ray_dir);
    } else if (mv.matType == PTC_VOLUMETRIC) {  // :622-672
        F3 inc_ray_dir = -attrib.wo;
        F3 norm = n;
        if (dot(inc_ray_dir, norm) >= .0f) {
            w_i = payload.dir;
        } else {
            F3 surface_pt = attrib.intersection;
            F3 curr_pt = attrib.prev_intersection;
            float ray_distance = length(surface_pt - curr_pt);
            float param_dist_t = 0.08f;
            transmit = computeTransmittance(mv.sigma_t, param_dist_t, ray_distance);
            F3 place_holder_vec = f3(.0f);
            float weight = samplePhaseHG(sem, inc_ray_dir, place_holder_vec, mv.g, rand_vals);
            addon_throughput *= transmit * weight;  // overwritten by every case of the switch below
            if (param_dist_t < ray_distance) {
                w_i = payload.dir;
                new_pos = curr_pt + (w_i * param_dist_t);
            } else {
                w_i = payload.dir;
                new_pos = surface_pt;
            }
        }
    } else if (mv.brdf_type == PTC_MOD_PHONG) {  // :675-684
        w = ((zeta_0 <= t) && (s.sampling_method == PTC_BRDF_SAMPLING)) ? r : n;
        F3 u = f3(.0f);
        F3 v = f3(.0f);
        genCoordFrame(sem, w, u, v);
        w_i = sphericalDir(theta, phi, u, v, w);
    } else {  // :685-696
        float theta_h_sample = atanf((mv.roughness * sqrtf(zeta_1)) / sqrtf(1.0f - zeta_1));
        float phi_h_sample = 2.0f * PIf * zeta_2;
        F3 w_h = n;
        F3 u_h = f3(.0f);
        F3 v_h = f3(.0f);
        genCoordFrame(sem, w_h, u_h, v_h);
        w_i = normalize(reflect(-attrib.wo, sphericalDir(theta_h_sample, phi_h_sample, u_h, v_h, w_h)));
    }

    if (cf.next_event_est) {  // :700-801
        for (uint64_t k = 0; k < s.n_qlights; ++k) {
            sampled_result = f3(.0f);
            float pdf_lights_k = .0f;
            float pdf_brdf = .0f;
            F3 w_i_dir = f3(.0f);
            F3 h = f3(.0f);
            float D = .0f;
            const PtcQuadLight& q = s.qlights[k];
            F3 a = f3(q.tri1.v1), b = f3(q.tri1.v2), cc = f3(q.tri2.v3);
            F3 ac = cc - a, ab = b - a;
            float area = length(cross(ab, ac));
            int root_light_samples = (int)sqrtf((float)s.light_samples);
            for (int i = 0; i < root_light_samples; ++i) {
                for (int j = 0; j < root_light_samples; ++j) {
                    float u1 = rnd(payload.seed);
                    float u2 = rnd(payload.seed);
                    F3 sampled_light_pos;
                    if (s.light_stratify)
                        sampled_light_pos = a + ((j + u1) * (ab / (float)root_light_samples)) + ((i + u2) * (ac / (float)root_light_samples));
                    else
                        sampled_light_pos = a + u1 * ab + u2 * ac;
                    F3 shadow_ray_origin = attrib.intersection;
                    F3 shadow_ray_dir = normalize(sampled_light_pos - shadow_ray_origin);
                    float light_dist = length(sampled_light_pos - shadow_ray_origin);
                    if (shadowVisible(c, shadow_ray_origin, shadow_ray_dir, cf.epsilon, light_dist - cf.epsilon)) {
                        w_i_dir = normalize(sampled_light_pos - shadow_ray_origin);
                        F3 f_brdf = f3(.0f);
                        if (mv.brdf_type == PTC_MOD_PHONG) {
                            f_brdf = phongBRDF(w_i_dir, attrib.wo, r, mv.shininess, f3(mv.diffuse), f3(mv.specular));
                        } else {
                            float wi_dot_n_dir = dot(w_i_dir, n);
                            float wo_dot_n_dir = dot(attrib.wo, n);
                            if (wi_dot_n_dir > .0f && wo_dot_n_dir > .0f) {
                                F3 f_brdf_GGX = ggxBRDF(w_i_dir, attrib.wo, n, mv.roughness, K_s);
                                f_brdf = (K_d / PIf) + f_brdf_GGX;
                            }
                        }
                        F3 x_prime = sampled_light_pos;
                        F3 x = shadow_ray_origin;
                        F3 n_geo = attrib.normal;  // the inner `n` of :767 (not normalised)
                        F3 n_light = normalize(cross(ab, ac));
                        float R = length(x - x_prime);
                        float G = (1.0f / powf(R, 2.0f)) * fmaxf(dot(n_geo, normalize(x_prime - x)), .0f) * (fmaxf(dot(n_light, normalize(x_prime - x)), .0f));
                        sampled_result += f_brdf * G;
                    }
                }
            }
            pdf_NEE = pdf_lights_k;  // :782 (never assigned: 0)
            pdf_brdf = ((1.0f - t) * fmaxf(dot(n, w_i_dir), .0f) / PIf) + ((t * D * dot(n, h)) / (4.0f * dot(h, w_i_dir)));  // :785
            float pdf_denom_sum = powf(pdf_NEE, exp_beta) + powf(pdf_brdf, exp_beta);
            float pdf_numerator = powf(pdf_NEE, exp_beta);
            float weight_i = pdf_numerator / pdf_denom_sum;
            if (cf.next_event_est == PTC_NEE_MIS)
                L_d += f3(q.color) * sampled_result * (area / (float)s.light_samples) * (1.0f / pdf_NEE) * weight_i;  // :794
            else
                L_d += f3(q.color) * sampled_result * (area / (float)s.light_samples);  // :798
        }
    }

    F3 f_brdf = f3(1.0f);
    float pdf = 1.0f;
    switch (s.sampling_method) {  // :811-897
        case PTC_HEMISPHERE_SAMPLING:
            f_brdf = (f3(mv.diffuse) / PIf) + (f3(mv.specular) * ((mv.shininess + 2.0f) / (2.0f * PIf)) * powf(fmaxf(dot(r, w_i), .0f), mv.shininess));
            pdf = 1.0f / (2.0f * PIf);
            addon_throughput = (f_brdf * fmaxf(dot(n, w_i), .0f) * (1.0f / pdf));
            break;
        case PTC_COSINE_SAMPLING:
            f_brdf = (f3(mv.diffuse) / PIf) + (f3(mv.specular) * ((mv.shininess + 2.0f) / (2.0f * PIf)) * powf(fmaxf(dot(r, w_i), .0f), mv.shininess));
            pdf = fmaxf(dot(n, w_i), .0f) / (PIf);
            addon_throughput = (f_brdf)*fmaxf(dot(n, w_i), .0f) * (1.0f / pdf);
            break;
        case PTC_BRDF_SAMPLING: {
            // :831-860 traces one probe shadow ray per quad light along w_i; its
            // results only feed locals that are never read again (pdf_lights_k,
            // pdf_NEE_brdf, weight_i_brdf, the shadowing brdf_sampling_contribution),
            // so it has no effect on the payload and is not traced here.
            if (mv.brdf_type == PTC_MOD_PHONG) {
                f_brdf = phongBRDF(w_i, attrib.wo, r, mv.shininess, K_d, K_s);
                pdf = ((1.0f - t) * (dot(n, w_i) / PIf)) + t * ((mv.shininess + 1.0f) / (2.0f * PIf)) * powf(fmaxf(dot(r, w_i), .0f), mv.shininess);
            } else {
                float wi_dot_n = dot(w_i, n);
                float wo_dot_n = dot(attrib.wo, n);
                if (wi_dot_n > .0f && wo_dot_n > .0f) {
                    F3 h = normalize(w_i + attrib.wo);
                    F3 f_brdf_GGX = ggxBRDF(w_i, attrib.wo, n, mv.roughness, K_s);
                    f_brdf = (K_d / PIf) + f_brdf_GGX;
                    pdf = ggxPDF(w_i, attrib.wo, n, h, t, mv.roughness);
                } else
                    f_brdf = f3(.0f);
            }
            addon_throughput = (f_brdf * fmaxf(dot(n, w_i), .0f) * (1.0f / pdf));
            break;
        }
        default:
            break;
    }

    if (cf.next_event_est && (payload.depth == 0)) {  // :900-917
        result += L_e + L_d;
    } else {
        if (cf.next_event_est == PTC_NEE_ON || cf.next_event_est == PTC_NEE_MIS)
            result += L_d;
        else
            result += L_e;
    }
    payload.radiance = result * payload.throughput;  // :918

    float q = .0f;
    if (cf.russian_roul) {  // :923-933
        q = 1.0f - fminf(fmaxf(fmaxf(payload.throughput.x, payload.throughput.y), payload.throughput.z), 1.0f);
        if (rnd(payload.seed) < q) {
            addon_throughput *= f3(.0f);
        } else {
            float thru_put_boost = (1.0f / (1.0f - q));
            addon_throughput *= thru_put_boost;
        }
    }
    if (mv.matType != PTC_GLASS) payload.throughput *= addon_throughput;  // :935-941 (glass: times 1)
    if (mv.matType == PTC_VOLUMETRIC || mv.matType == PTC_GLASS)             // :943-957
        payload.origin = new_pos;  // glass never assigns new_pos: the path continues from (0,0,0)
    else
        payload.origin = attrib.intersection;
    payload.dir = w_i;
    payload.depth++;  // :959-962 (the condition is a tautology)
}

// ------------------------------------------------------------ ray generation --
inline Ray cameraRay(const PtcConfig& cf, uint32_t x, uint32_t y, float jx, float jy) {
    // PinholeCamera.cu:44-51
    float px = (float)x + jx, py = (float)y + jy;
    float abx = cf.tanHFov.x * (px - cf.hSize.x) / cf.hSize.x;
    float aby = cf.tanHFov.y * (py - cf.hSize.y) / cf.hSize.y;
    F3 dir = normalize(abx * f3(cf.u) + aby * f3(cf.v) - f3(cf.w));
    return Ray{f3(cf.eye), dir, cf.epsilon, RAY_MAX};
}

// state of the LCG after n steps
uint32_t lcgSkip(uint32_t st, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) lcg(st);
    return st;
}

// PinholeCamera.cu:20-81 for one pixel; returns the pre-gamma radiance sum of the
// samples [j0, j1).
F3 generateRays(const Ctx& c, uint32_t x, uint32_t y, int j0, int j1) {
    const World& w = *c.w;
    const OrcScene& s = *w.s;
    const OrcOptions& o = *w.o;
    const PtcConfig& cf = s.config;
    const uint32_t index = x * s.height + y;  // :23, x-major
    const uint32_t key0 = index * (uint32_t)o.frame_id;
    uint32_t seed = tea16(key0, 0);  // :24 (jitter stream)
    F3 result = f3(0.f);
    Payload payload;
    uint32_t i = 0;
    void (*program)(const Ctx&, const Attributes&, Payload&) =
        s.integrator == ORC_RAYTRACER ? closestHit : s.integrator == ORC_ANALYTICDIRECT ? analyticDirect : s.integrator == ORC_DIRECT ? direct : s.integrator == ORC_PATHTRACER ? pathTracer : pathTracer2;

    // Reference seeding is history dependent, so a range that does not start at 0
    // still has to replay the earlier samples to know `i` and the jitter stream.
    const int jstart = o.seeding == 0 ? 0 : j0;
    if (o.seeding == 1 && j0 > 1) seed = lcgSkip(seed, 2u * (uint32_t)(j0 - 1));
    for (int j = jstart; j < j1; ++j) {
        const bool counted = j >= j0;
        payload.radiance = f3(.0f);
        payload.throughput = f3(1.0f);
        payload.depth = 0;
        payload.done = false;
        float jx = (j == 0) ? 0.5f : rnd(seed);
        float jy = (j == 0) ? 0.5f : rnd(seed);
        Ray ray = cameraRay(cf, x, y, jx, jy);
        F3 origin = ray.origin, dir = ray.direction;
        F3 sampleSum = f3(0.f);
        uint32_t b = 0;
        do {
            payload.seed = o.seeding == 0 ? tea16(key0, i++) : tea16(key0, ((uint32_t)j << 12) | b);  // :56
            Ray r{origin, dir, cf.epsilon, RAY_MAX};
            // The volumetric branch reads payload.dir (RayTracer.cu:633, 661).  From the second
            // segment on that is this ray's direction (PinholeCamera.cu:65); for a camera ray
            // the reference leaves it uninitialised / stale from the previous sample, so it is
            // DEFINED here as the camera ray's direction.
            payload.origin = origin;
            payload.dir = dir;
            if (counted) c.cnt->ext++;
            Closest hit = traceClosest(w, r);
            if (hit.prim < 0) {  // Common.cu:13-18
                payload.radiance = f3(s.background[0], s.background[1], s.background[2]);
                payload.done = true;
            } else {
                Attributes attrib;
                if ((uint64_t)hit.prim < s.n_tris)
                    triangleAttributes(s.tris[hit.prim], r, hit.t, attrib);
                else
                    sphereAttributes(s.spheres[hit.prim - s.n_tris], r, attrib);
                program(c, attrib, payload);
            }
            if (counted) {
                if (o.seeding == 0)
                    result += payload.radiance;  // :61
                else
                    sampleSum += payload.radiance;
            }
            payload.radiance = f3(0.f);
            origin = payload.origin;
            dir = payload.dir;
            ++b;
            // Result-preserving cull, counter seeding only: under reference seeding every segment the
            // reference traces advances `i` (:56), which seeds all later samples of the pixel, and how
            // many segments a dead path still has is not knowable without tracing them.
            if (o.cull_zero && o.seeding == 1 && payload.throughput.x == 0.f && payload.throughput.y == 0.f && payload.throughput.z == 0.f) break;
        } while (!payload.done && payload.depth != w.maxDepthEff && b < o.bounce_cap);  // :66
        if (counted) {
            c.cnt->paths++;
            if (o.seeding == 1) result += sampleSum;
        }
    }
    return result;
}

thread_local std::string g_err;

int validate(const OrcScene* s, const OrcOptions* o) {
    if (!s || !o) { g_err = "null argument"; return PTC_ERR_INVALID; }
    if (s->integrator < 0 || s->integrator > 4) { g_err = "bad integrator"; return PTC_ERR_INTEGRATOR; }
    if (o->bounce_cap == 0 || o->bounce_cap > 4096) { g_err = "bounce_cap must be 1..4096"; return PTC_ERR_INVALID; }
    return PTC_OK;
}

void makeWorld(World& w, const OrcScene* s, const OrcOptions* o) {
    w.s = s;
    w.o = o;
    w.useBvh = o->use_bvh != 0;
    w.maxDepthEff = std::min<uint32_t>(s->config.maxDepth, o->bounce_cap);
    if (w.useBvh) {
        int n = (int)(s->n_tris + s->n_spheres);
        std::vector<Box> boxes(n);
        for (int i = 0; i < n; ++i) boxes[i] = primBox(*s, i);
        w.bvh.order.resize(n);
        for (int i = 0; i < n; ++i) w.bvh.order[i] = i;
        w.bvh.nodes.reserve(2 * n + 1);
        if (n) buildRec(w.bvh, boxes, 0, n);
    }
}

struct Window {
    uint32_t x0, y0, x1, y1, stride;
};
Window window(const OrcScene* s, const OrcOptions* o) {
    Window w{o->x0, o->y0, o->x1, o->y1, o->stride ? o->stride : 1u};
    if (w.x1 == 0 && w.y1 == 0) {
        w.x0 = w.y0 = 0;
        w.x1 = s->width;
        w.y1 = s->height;
    }
    w.x1 = std::min(w.x1, s->width);
    w.y1 = std::min(w.y1, s->height);
    return w;
}

}  // namespace

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

uint32_t orc_tea16(uint32_t v0, uint32_t v1) { return tea16(v0, v1); }
uint32_t orc_lcg(uint32_t* state) { return lcg(*state); }
float orc_rnd(uint32_t* state) { return rnd(*state); }

int orc_refract(const float* i3, const float* n3, float ior, float* r3) {
    F3 r;
    const bool ok = refract(r, f3(i3[0], i3[1], i3[2]), f3(n3[0], n3[1], n3[2]), ior);
    r3[0] = r.x; r3[1] = r.y; r3[2] = r.z;
    return ok ? 1 : 0;
}

float orc_fresnel_schlick(float cos_theta) { return fresnel_schlick(cos_theta); }
float orc_phase_hg(float wi_dot_wo, float g) { return phaseHG(wi_dot_wo, g); }

void orc_mat4_inverse(const float* m16, float* out16) {
    M4 a;
    std::memcpy(a.m, m16, sizeof(a.m));
    M4 r = inverse(a);
    std::memcpy(out16, r.m, sizeof(r.m));
}

void orc_to_rgba8(const float* result, uint32_t width, uint32_t height, unsigned char* out) {
    auto cast = [](float v) {
        v = v > 1.f ? 1.f : v < 0.f ? 0.f : v;
        return static_cast<unsigned char>(v * 255);
    };
    for (uint32_t i = 0; i < height; i++)
        for (uint32_t j = 0; j < width; j++) {
            size_t index = ((size_t)i * width + j) * 4;
            const float* px = result + ((size_t)(height - i - 1) * width + j) * 3;
            out[index + 0] = cast(px[0]);
            out[index + 1] = cast(px[1]);
            out[index + 2] = cast(px[2]);
            out[index + 3] = 255;
        }
}

// One pass of the bounce loop's body (PinholeCamera.cu:56-66) for caller-made segments.
int orc_probe_segments(const OrcScene* s, const OrcOptions* o, uint32_t n, const PtcSegmentProbe* in, PtcSegmentResult* out) {
    int rc = validate(s, o);
    if (rc) return rc;
    World w;
    makeWorld(w, s, o);
    Counters cnt;
    Ctx c{&w, &cnt};
    void (*program)(const Ctx&, const Attributes&, Payload&) =
        s->integrator == ORC_RAYTRACER ? closestHit : s->integrator == ORC_ANALYTICDIRECT ? analyticDirect : s->integrator == ORC_DIRECT ? direct : s->integrator == ORC_PATHTRACER ? pathTracer : pathTracer2;
    for (uint32_t i = 0; i < n; ++i) {
        const PtcSegmentProbe& p = in[i];
        Payload payload;
        payload.radiance = f3(.0f);
        payload.throughput = f3(p.thr[0], p.thr[1], p.thr[2]);
        payload.depth = p.depth;
        payload.done = false;
        payload.seed = p.seed;
        const F3 origin = f3(p.o[0], p.o[1], p.o[2]), dir = f3(p.d[0], p.d[1], p.d[2]);
        payload.origin = origin;
        payload.dir = dir;
        Ray r{origin, dir, s->config.epsilon, RAY_MAX};
        Closest hit = traceClosest(w, r);
        if (hit.prim < 0) {  // Common.cu:13-18
            payload.radiance = f3(s->background[0], s->background[1], s->background[2]);
            payload.done = true;
        } else {
            Attributes attrib;
            if ((uint64_t)hit.prim < s->n_tris)
                triangleAttributes(s->tris[hit.prim], r, hit.t, attrib);
            else
                sphereAttributes(s->spheres[hit.prim - s->n_tris], r, attrib);
            program(c, attrib, payload);
        }
        PtcSegmentResult& q = out[i];
        std::memset(&q, 0, sizeof(q));
        q.radiance[0] = payload.radiance.x; q.radiance[1] = payload.radiance.y; q.radiance[2] = payload.radiance.z;
        q.prim = hit.prim;
        q.t = hit.prim < 0 ? 0.f : hit.t;
        q.alive = (!payload.done && payload.depth != w.maxDepthEff) ? 1 : 0;
        if (q.alive) {
            q.thr[0] = payload.throughput.x; q.thr[1] = payload.throughput.y; q.thr[2] = payload.throughput.z;
            q.o[0] = payload.origin.x; q.o[1] = payload.origin.y; q.o[2] = payload.origin.z;
            q.d[0] = payload.dir.x; q.d[1] = payload.dir.y; q.d[2] = payload.dir.z;
            q.depth = payload.depth;
        }
    }
    return PTC_OK;
}

// The shading helpers on raw inputs; layouts as for ptc_eval_shading (include/ptcore.h).
int orc_eval_shading(int fn, uint32_t n, const float* in, float* out) {
    if (fn < 0 || fn > 4 || !in || !out) return PTC_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        const float* a = in + 16 * (size_t)i;
        const F3 v0 = f3(a[0], a[1], a[2]), v1 = f3(a[3], a[4], a[5]), v2 = f3(a[6], a[7], a[8]), v3 = f3(a[9], a[10], a[11]);
        F3 r = f3(.0f);
        if (fn == 0) {
            r = phongBRDF(v0, f3(.0f), v1, a[12], v2, v3);
        } else if (fn == 1) {
            r = ggxBRDF(v0, v1, v2, a[12], v3);
        } else if (fn == 2) {
            r = f3(ggxPDF(v0, f3(.0f), v1, v2, a[12], a[13]), .0f, .0f);
        } else if (fn == 3) {  // RayTracer.cu:566, 675-684 with by-reference helpers
            float theta = .0f, phi = .0f;
            const float zeta[3] = {a[3], a[4], a[5]};
            computePolarAngles(1, (uint32_t)a[8], phi, theta, zeta, a[6], a[7]);
            F3 u = f3(.0f), v = f3(.0f);
            genCoordFrame(1, v0, u, v);
            r = sphericalDir(theta, phi, u, v, v0);
        } else {  // :685-696
            const float zeta_1 = a[6], zeta_2 = a[7], roughness = a[8];
            const float theta_h_sample = atanf((roughness * sqrtf(zeta_1)) / sqrtf(1.0f - zeta_1));
            const float phi_h_sample = 2.0f * PIf * zeta_2;
            F3 u_h = f3(.0f), v_h = f3(.0f);
            genCoordFrame(1, v0, u_h, v_h);
            r = normalize(reflect(-v1, sphericalDir(theta_h_sample, phi_h_sample, u_h, v_h, v0)));
        }
        out[4 * (size_t)i] = r.x;
        out[4 * (size_t)i + 1] = r.y;
        out[4 * (size_t)i + 2] = r.z;
        out[4 * (size_t)i + 3] = 0.f;
    }
    return PTC_OK;
}

int orc_primary_hits(const OrcScene* s, const OrcOptions* o, PtcHitRecord* out) {
    int rc = validate(s, o);
    if (rc) return rc;
    World w;
    makeWorld(w, s, o);
    Window win = window(s, o);
    int nthreads = std::max(1, o->threads);
    std::atomic<uint32_t> next(win.y0);
    auto work = [&]() {
        for (;;) {
            uint32_t y = next.fetch_add(win.stride);
            if (y >= win.y1) break;
            for (uint32_t x = win.x0; x < win.x1; x += win.stride) {
                Ray r = cameraRay(s->config, x, y, 0.5f, 0.5f);
                Closest h = traceClosest(w, r);
                PtcHitRecord& rec = out[(size_t)y * s->width + x];
                rec.prim = h.prim;
                rec.t = h.prim < 0 ? 0.f : h.t;
                rec.u = h.prim < 0 ? 0.f : h.u;
                rec.v = h.prim < 0 ? 0.f : h.v;
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; ++i) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return PTC_OK;
}

int orc_render(const OrcScene* s, const OrcOptions* o, float* sum_out, float* result_out, OrcStats* stats) {
    int rc = validate(s, o);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    World w;
    makeWorld(w, s, o);
    Window win = window(s, o);
    int j0 = o->spp_begin, j1 = o->spp_end;
    if (j1 <= 0) {
        j0 = 0;
        j1 = s->samples_per_pixel;
    }
    const int spp_total = s->samples_per_pixel;
    int nthreads = std::max(1, o->threads);
    std::vector<Counters> counters(nthreads);
    std::atomic<uint32_t> next(win.y0);
    auto work = [&](int tid) {
        Ctx c{&w, &counters[tid]};
        for (;;) {
            uint32_t y = next.fetch_add(win.stride);
            if (y >= win.y1) break;
            for (uint32_t x = win.x0; x < win.x1; x += win.stride) {
                F3 sum = generateRays(c, x, y, j0, j1);
                size_t at = ((size_t)y * s->width + x) * 3;
                if (sum_out) {
                    sum_out[at] = sum.x;
                    sum_out[at + 1] = sum.y;
                    sum_out[at + 2] = sum.z;
                }
                if (result_out) {
                    // PinholeCamera.cu:70-71
                    F3 r = sum / (float)spp_total;
                    const float g = s->config.gamma;
                    result_out[at] = powf(r.x, 1.0f / g);
                    result_out[at + 1] = powf(r.y, 1.0f / g);
                    result_out[at + 2] = powf(r.z, 1.0f / g);
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; ++i) th.emplace_back(work, i);
    work(0);
    for (auto& t : th) t.join();
    if (stats) {
        stats->paths = stats->extension_rays = stats->shadow_rays = 0;
        for (auto& c : counters) {
            stats->paths += c.paths;
            stats->extension_rays += c.ext;
            stats->shadow_rays += c.shadow;
        }
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return PTC_OK;
}

}  // extern "C"
