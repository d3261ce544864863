// shading.h -- BRDFs, light sampling and the per-segment integrator logic
// (device code).  The formulas and their operation order follow the reference's
// closest-hit programs in OptiXRenderer/RayTracer.cu; each function cites the
// lines it restates.  What differs is the schedule: the shadow rays of a
// segment are traced by their own kernel (one thread per shadow ray) which
// leaves one float3 term per ray, and the shading kernel then folds those
// terms in the order the reference's loops add them.
#pragma once

#include "device_scene.h"
#include "intersect.h"

// Division, square root and normalisation of the shading code, in two flavours.  FAST = false: correctly rounded
// IEEE operations in the reference's order (what the CPU restatement computes).  FAST = true: the hardware's 1-ulp
// reciprocal, square root and reciprocal square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32), one or two instructions
// where the correctly rounded sequences take ten or more.  Radiance and sampled directions are compared with the
// CPU restatement to a tolerance (another libm gives other last bits anyway), so the path tracer under `intended`
// semantics takes the fast flavour; hit records, ray origins and everything else in intersect.h stay exact, and
// so does `literal` semantics, where a last bit decides between 1 and NaN.
template <bool FAST>
struct ShMath {
    static __device__ __forceinline__ float rcp(float x) { return FAST ? __builtin_amdgcn_rcpf(x) : 1.0f / x; }
    static __device__ __forceinline__ float div(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
    static __device__ __forceinline__ float sqrt(float x) { return FAST ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); }
    static __device__ __forceinline__ float length(vec3 v) { return sqrt(dot(v, v)); }
    static __device__ __forceinline__ vec3 normalize(vec3 v) { return FAST ? v * __builtin_amdgcn_rsqf(dot(v, v)) : ::normalize(v); }
    static __device__ __forceinline__ vec3 div(vec3 a, float s) { return a * rcp(s); }  // a / s is a * (1 / s) in the exact flavour too
};

// powf(x, y) for the non-negative bases of the Phong terms (a clamped cosine, a uniform
// random number), as exp2(y * log2(x)) on the hardware's 1-ulp log2 / exp2 units.  The
// error is that of the exponent, |y * log2 x| * 2^-23, i.e. below 2e-6 relative wherever
// the result is above 1e-6; libm's powf is several times the cost of everything else in
// the Phong branch.  PT_LIBM_POWF restores the libm call.
__device__ __forceinline__ float pow_pos(float x, float y) {
#ifdef PT_LIBM_POWF
    return powf(x, y);
#else
    if (y == 0.0f) return 1.0f;  // powf(x, 0) is 1 for every x
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));  // v_exp_f32, v_log_f32
#endif
}

// RayTracer.cu:259-263
template <bool FAST = false>
__device__ __forceinline__ vec3 phong_brdf(vec3 wi, vec3 reflect_vec, float shininess, vec3 diffuse, vec3 specular) {
    const float norm = FAST ? (shininess + 2.0f) * (1.0f / (2.0f * PT_PIf)) : (shininess + 2.0f) / (2.0f * PT_PIf);
    return (diffuse / PT_PIf) + (specular * norm * pow_pos(fmaxf(dot(reflect_vec, wi), .0f), shininess));
}

// The reference evaluates its GGX terms through chains of libm calls on an angle:
// tanf(acosf(x))^2, powf(cosf(acosf(x)), 4), sinf/cosf(atanf(a)).  Written with the angle
// eliminated these are a handful of multiplications instead of several hundred
// instructions, and agree with the libm chains to a few ulp (the same size as the
// differences between two libm implementations).  What the chains do beyond rounding
// is kept: acosf(x) is NaN for |x| > 1, which normalised vectors reach by one ulp.
template <bool FAST = false>
__device__ __forceinline__ float tan2_of_cos(float x) {  // tanf(acosf(x))^2
    if (!(fabsf(x) <= 1.0f)) return __builtin_nanf("");
    return ShMath<FAST>::div((1.0f - x) * (1.0f + x), x * x);
}
__device__ __forceinline__ float cos4_of_cos(float x) {  // powf(cosf(acosf(x)), 4)
    if (!(fabsf(x) <= 1.0f)) return __builtin_nanf("");
    const float x2 = x * x;
    return x2 * x2;
}

// RayTracer.cu:265-285
template <bool FAST = false>
__device__ __forceinline__ vec3 ggx_brdf(vec3 wi, vec3 wo, vec3 n, float roughness, vec3 K_s) {
    typedef ShMath<FAST> M;
    float wi_dot_n_dir = dot(wi, n);
    float wo_dot_n_dir = dot(wo, n);
    float alpha = roughness;
    // h and n are normalised with correctly rounded operations in either flavour: acosf(dot(h, n)) is NaN in the source
    // when the product exceeds 1, which it does by an ulp now and then for h close to n; a sloppier length would make
    // that more frequent than it is in the reference
    vec3 h = normalize(wi + wo);
    const float cos_h = dot(h, n);  // theta_h = acosf(cos_h)
    float cos_theta_h_4 = cos4_of_cos(cos_h);
    float alpha_tan_theta_h_sq = (alpha * alpha) + tan2_of_cos<FAST>(cos_h);
    float D = M::div(alpha * alpha, PT_PIf * cos_theta_h_4 * (alpha_tan_theta_h_sq * alpha_tan_theta_h_sq));
    float G_1_wi = (wi_dot_n_dir > .0f) ? M::div(2.0f, 1.0f + M::sqrt(1.0f + (alpha * alpha) * tan2_of_cos<FAST>(wi_dot_n_dir))) : .0f;
    float G_1_wo = (wo_dot_n_dir > .0f) ? M::div(2.0f, 1.0f + M::sqrt(1.0f + (alpha * alpha) * tan2_of_cos<FAST>(wo_dot_n_dir))) : .0f;
    float G = G_1_wi * G_1_wo;
    const float om = 1.0f - dot(wi, h), om2 = om * om;
    vec3 F = K_s + (mk3(1.0f) - K_s) * (om2 * om2 * om);  // powf(., 5.0f)
    return M::div(F * G * D, 4.0f * wi_dot_n_dir * wo_dot_n_dir);
}

// RayTracer.cu:287-294
template <bool FAST = false>
__device__ __forceinline__ float ggx_pdf(vec3 wi, vec3 n, vec3 half_angle_vec, float t_val, float roughness) {
    typedef ShMath<FAST> M;
    float alpha = roughness;
    const float cos_h = dot(half_angle_vec, n);
    float alpha_tan_theta_h_sq = (alpha * alpha) + tan2_of_cos<FAST>(cos_h);
    float D = M::div(alpha * alpha, PT_PIf * cos4_of_cos(cos_h) * (alpha_tan_theta_h_sq * alpha_tan_theta_h_sq));
    return ((1.0f - t_val) * fmaxf(dot(n, wi), .0f) / PT_PIf) + M::div(t_val * D * dot(n, half_angle_vec), 4.0f * dot(half_angle_vec, wi));
}

// A polar angle carried as its cosine and sine.
struct PolarAngle {
    float c, s;
};
__device__ __forceinline__ PolarAngle angle_zero() { return PolarAngle{1.0f, 0.0f}; }
// theta = acosf(x), x in [0, 1]
template <bool FAST = false>
__device__ __forceinline__ PolarAngle angle_acos(float x) {
    if (!(fabsf(x) <= 1.0f)) return PolarAngle{__builtin_nanf(""), __builtin_nanf("")};
    return PolarAngle{x, ShMath<FAST>::sqrt((1.0f - x) * (1.0f + x))};
}
// theta = atanf(a), a >= 0.  `libm_chain`: go through the angle like the source does.
// Literal semantics need that: their GGX half vector is cos(theta) * n, renormalised, and
// whether dot(h, n) then lands on 1 or on 1 + 1 ulp (NaN downstream) depends on the last
// bit of that cosine.
template <bool FAST = false>
__device__ __forceinline__ PolarAngle angle_atan(float a, bool libm_chain) {
    if (libm_chain) {
        const float th = atanf(a);
        return PolarAngle{cosf(th), sinf(th)};
    }
    const float c = FAST ? __builtin_amdgcn_rsqf(1.0f + a * a) : 1.0f / sqrtf(1.0f + a * a);
    return PolarAngle{c, a * c};
}

// RayTracer.cu:316-322.  The azimuth is always 2*pi*zeta in the source; it is passed here
// as zeta ("turns") so that sine and cosine need no reduction by an inexact 2*pi.
template <bool FAST = false>
__device__ __forceinline__ vec3 spherical_dir(PolarAngle theta, float phi_turns, vec3 u, vec3 v, vec3 w) {
    float sp, cp;
    sincospif(2.0f * phi_turns, &sp, &cp);
    vec3 s = mk3(cp * theta.s, sp * theta.s, theta.c);
    return ShMath<FAST>::normalize((s.x * u + s.y * v + s.z * w));
}

// RayTracer.cu:302-312.  The aligned-axis test there reads `v`, which is the
// zero vector at every call site, so a = (1,0,0) is always taken.  Under
// literal semantics u and v are by-value parameters: the caller keeps zeros.
template <bool FAST = false>
__device__ __forceinline__ void gen_coord_frame(int semantics, vec3 w, vec3& u, vec3& v) {
    if (semantics == 1) {
        vec3 a = mk3(1.0f, .0f, .0f);
        u = ShMath<FAST>::normalize(cross(a, w));
        v = ShMath<FAST>::normalize(cross(w, u));
    }
}

// RayTracer.cu:232-257 (by-value outputs under literal semantics).
template <bool FAST = false>
__device__ __forceinline__ void compute_polar_angles(int semantics, uint32_t sampling_method, float& phi_turns, PolarAngle& theta, float z0, float z1, float z2, float t, float shininess) {
    typedef ShMath<FAST> M;
    if (semantics != 1) return;
    switch (sampling_method) {
        case PTC_HEMISPHERE_SAMPLING:
            phi_turns = z2;
            theta = angle_acos<FAST>(z1);
            break;
        case PTC_COSINE_SAMPLING:
            phi_turns = z2;
            theta = angle_acos<FAST>(M::sqrt(z1));
            break;
        case PTC_BRDF_SAMPLING:
            phi_turns = z2;
            if (z0 > t)
                theta = angle_acos<FAST>(M::sqrt(z1));
            else
                theta = angle_acos<FAST>(pow_pos(z1, M::rcp(shininess + 1.0f)));
            break;
        default:
            break;
    }
}

__device__ __forceinline__ vec3 P3v(const PtcFloat3& p) { return mk3(p.x, p.y, p.z); }

// Point on quad light k for light sample (i,j) with random numbers u1,u2
// (RayTracer.cu:187-194 / 391-398 / 727-734).
struct QuadFrame {
    vec3 a, ab, ac;
};
__device__ __forceinline__ vec3 quad_sample(const QuadFrame& q, bool stratify, int i, int j, int root, float u1, float u2, bool variant_cdl) {
    if (stratify) {
        if (variant_cdl)  // computeDirectLight: (j+u1)*(ab/N) without the extra parentheses - same value order
            return q.a + (j + u1) * (q.ab / (float)root) + (i + u2) * (q.ac / (float)root);
        return q.a + ((j + u1) * (q.ab / (float)root)) + ((i + u2) * (q.ac / (float)root));
    }
    if (variant_cdl) return q.a + q.ab * u1 + q.ac * u2;
    return q.a + u1 * q.ab + u2 * q.ac;
}
