// ptmath.h -- small vector / matrix library shared by the host code and the HIP
// kernels of the path-tracing core.
//
// The operation ORDER of every function is part of the contract: hit records
// are compared bit for bit against the CPU oracle, so each helper spells out
// the sequence of IEEE-754 single-precision operations the OptiX SDK 6.x math
// headers (optixu_math_namespace.h / optixu_matrix_namespace.h, not shipped
// with the reference) are documented to perform:
//   a / s        = a * (1.0f / s)              (vector divided by scalar)
//   normalize(v) = v * (1.0f / sqrtf(dot(v,v)))
//   length(v)    = sqrtf(dot(v,v))
//   dot(a,b)     = a.x*b.x + a.y*b.y + a.z*b.z (left to right)
//   reflect(i,n) = i - 2.0f * n * dot(n,i)
//   lerp(a,b,t)  = a + t*(b-a)
// Everything must be compiled with -ffp-contract=off (no FMA contraction).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

#define PT_PIf 3.14159265358979323846f
#define PT_RAY_MAX 1e27f /* RT_DEFAULT_MAX */

struct vec3 {
    float x, y, z;
};
struct vec2 {
    float x, y;
};
struct vec4 {
    float x, y, z, w;
};

PT_HD vec3 mk3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD vec3 mk3(float s) { return mk3(s, s, s); }
PT_HD vec4 mk4(vec3 a, float w) { vec4 r; r.x = a.x; r.y = a.y; r.z = a.z; r.w = w; return r; }
PT_HD vec3 xyz(vec4 a) { return mk3(a.x, a.y, a.z); }

PT_HD vec3 operator+(vec3 a, vec3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD vec3 operator-(vec3 a, vec3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD vec3 operator-(vec3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_HD vec3 operator*(vec3 a, vec3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD vec3 operator*(vec3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD vec3 operator*(float s, vec3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
PT_HD vec3 operator/(vec3 a, float s) { float inv = 1.0f / s; return a * inv; }
PT_HD vec3 operator/(vec3 a, vec3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_HD vec3& operator+=(vec3& a, vec3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
PT_HD vec3& operator*=(vec3& a, vec3 b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
PT_HD vec3& operator*=(vec3& a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }

PT_HD float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD vec3 cross(vec3 a, vec3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_HD float length(vec3 v) { return sqrtf(dot(v, v)); }
PT_HD vec3 normalize(vec3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
PT_HD vec3 reflect(vec3 i, vec3 n) { return i - 2.0f * n * dot(n, i); }
// refract(r, i, n, ior) of the OptiX 6.x vector header: n is flipped and `ior` used as eta
// when i leaves through n, 1/ior otherwise; false (r = 0) on total internal reflection.
PT_HD bool refract(vec3& r, vec3 i, vec3 n, float ior) {
    vec3 nn = n;
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
        r = mk3(0.f);
        return false;
    }
    r = normalize(eta * i - (eta * negNdotV + sqrtf(k)) * nn);
    return true;
}
// fresnel_schlick(cos_theta, exponent 5, minimum 0, maximum 1) of the same header
PT_HD float fresnel_schlick(float cos_theta) {
    const float v = 0.0f + (1.0f - 0.0f) * powf(fmaxf(0.0f, 1.0f - cos_theta), 5.0f);
    return fminf(fmaxf(v, 0.0f), 1.0f);
}
PT_HD vec3 lerp3(vec3 a, vec3 b, float t) { return a + t * (b - a); }

// Row-major 4x4 matrix; element (r,c) at m[r*4+c].
struct mat4 {
    float m[16];
};

PT_HD mat4 mat4_identity() {
    mat4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = 0.0f;
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}

PT_HD mat4 mat4_mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.m[i * 4 + k] * b.m[k * 4 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}

PT_HD vec4 mat4_mul(const mat4& a, vec4 v) {
    vec4 r;
    r.x = a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z + a.m[3] * v.w;
    r.y = a.m[4] * v.x + a.m[5] * v.y + a.m[6] * v.z + a.m[7] * v.w;
    r.z = a.m[8] * v.x + a.m[9] * v.y + a.m[10] * v.z + a.m[11] * v.w;
    r.w = a.m[12] * v.x + a.m[13] * v.y + a.m[14] * v.z + a.m[15] * v.w;
    return r;
}

PT_HD mat4 mat4_transpose(const mat4& a) {
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i * 4 + j] = a.m[j * 4 + i];
    return r;
}

PT_HD mat4 mat4_translate(vec3 t) {
    mat4 r = mat4_identity();
    r.m[3] = t.x; r.m[7] = t.y; r.m[11] = t.z;
    return r;
}

PT_HD mat4 mat4_scale(vec3 s) {
    mat4 r = mat4_identity();
    r.m[0] = s.x; r.m[5] = s.y; r.m[10] = s.z;
    return r;
}

// Rotation by `radians` about `axis` (taken as given, not normalised).
inline mat4 mat4_rotate(float radians, vec3 axis) {
    mat4 r = mat4_identity();
    const float s = sinf(radians), c = cosf(radians);
    const float ux = axis.x, uy = axis.y, uz = axis.z;
    r.m[0] = ux * ux + c * (1 - ux * ux);
    r.m[1] = ux * uy * (1 - c) - uz * s;
    r.m[2] = uz * ux * (1 - c) + uy * s;
    r.m[4] = ux * uy * (1 - c) + uz * s;
    r.m[5] = uy * uy + c * (1 - uy * uy);
    r.m[6] = uy * uz * (1 - c) - ux * s;
    r.m[8] = uz * ux * (1 - c) - uy * s;
    r.m[9] = uy * uz * (1 - c) + ux * s;
    r.m[10] = uz * uz + c * (1 - uz * uz);
    return r;
}

// Inverse by cofactor expansion: inv = adj(M) * (1/det), det expanded along
// the first column.
PT_HD mat4 mat4_inverse(const mat4& a) {
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
    mat4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = d * c[i];
    return r;
}

// ------------------------------------------------------------------- RNG --
// tea<16> / lcg / rnd as in the reference's random.h:31-67 (pure 32-bit
// integer arithmetic, bit-exact on every target).
PT_HD uint32_t tea16(uint32_t val0, uint32_t val1) {
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (int n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

PT_HD uint32_t lcg_next(uint32_t& prev) {
    prev = 1664525u * prev + 1013904223u;
    return prev & 0x00FFFFFFu;
}

PT_HD float rnd(uint32_t& prev) { return (float)lcg_next(prev) / (float)0x01000000; }

// State of the LCG after `n` steps from `state` (O(log n) jump-ahead).
PT_HD uint32_t lcg_skip(uint32_t state, uint32_t n) {
    uint32_t a = 1664525u, c = 1013904223u;  // one step: x -> a*x + c
    uint32_t A = 1u, C = 0u;                 // accumulated map
    while (n) {
        if (n & 1u) { A = a * A; C = a * C + c; }
        c = (a + 1u) * c;
        a = a * a;
        n >>= 1;
    }
    return A * state + C;
}
