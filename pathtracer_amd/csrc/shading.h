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

// RayTracer.cu:259-263
__device__ __forceinline__ vec3 phong_brdf(vec3 wi, vec3 reflect_vec, float shininess, vec3 diffuse, vec3 specular) {
    return (diffuse / PT_PIf) + (specular * ((shininess + 2.0f) / (2.0f * PT_PIf)) * powf(fmaxf(dot(reflect_vec, wi), .0f), shininess));
}

// RayTracer.cu:265-285
__device__ __forceinline__ vec3 ggx_brdf(vec3 wi, vec3 wo, vec3 n, float roughness, vec3 K_s) {
    float wi_dot_n_dir = dot(wi, n);
    float wo_dot_n_dir = dot(wo, n);
    float alpha = roughness;
    vec3 h = normalize(wi + wo);
    float theta_h = acosf(dot(h, n));
    float cos_theta_h_4 = powf(cosf(theta_h), 4.0f);
    float alpha_tan_theta_h_sq = (alpha * alpha) + powf(tanf(theta_h), 2.0f);
    float D = (alpha * alpha) / (PT_PIf * cos_theta_h_4 * powf(alpha_tan_theta_h_sq, 2.0f));
    float G_1_wi = (wi_dot_n_dir > .0f) ? 2.0f / (1.0f + sqrtf(1.0f + (alpha * alpha) * powf(tanf(acosf(dot(wi, n))), 2.0f))) : .0f;
    float G_1_wo = (wo_dot_n_dir > .0f) ? 2.0f / (1.0f + sqrtf(1.0f + (alpha * alpha) * powf(tanf(acosf(dot(wo, n))), 2.0f))) : .0f;
    float G = G_1_wi * G_1_wo;
    vec3 F = K_s + (mk3(1.0f) - K_s) * powf(1.0f - dot(wi, h), 5.0f);
    return (F * G * D) / (4.0f * wi_dot_n_dir * wo_dot_n_dir);
}

// RayTracer.cu:287-294
__device__ __forceinline__ float ggx_pdf(vec3 wi, vec3 n, vec3 half_angle_vec, float t_val, float roughness) {
    float alpha = roughness;
    float theta_h = acosf(dot(half_angle_vec, n));
    float alpha_tan_theta_h_sq = (alpha * alpha) + powf(tanf(theta_h), 2.0f);
    float D = (alpha * alpha) / (PT_PIf * powf(cosf(theta_h), 4.0f) * powf(alpha_tan_theta_h_sq, 2.0f));
    return ((1.0f - t_val) * fmaxf(dot(n, wi), .0f) / PT_PIf) + ((t_val * D * dot(n, half_angle_vec)) / (4.0f * dot(half_angle_vec, wi)));
}

// RayTracer.cu:316-322
__device__ __forceinline__ vec3 spherical_dir(float theta, float phi, vec3 u, vec3 v, vec3 w) {
    vec3 s = mk3(cosf(phi) * sinf(theta), sinf(phi) * sinf(theta), cosf(theta));
    return normalize((s.x * u + s.y * v + s.z * w));
}

// RayTracer.cu:302-312.  The aligned-axis test there reads `v`, which is the
// zero vector at every call site, so a = (1,0,0) is always taken.  Under
// literal semantics u and v are by-value parameters: the caller keeps zeros.
__device__ __forceinline__ void gen_coord_frame(int semantics, vec3 w, vec3& u, vec3& v) {
    if (semantics == 1) {
        vec3 a = mk3(1.0f, .0f, .0f);
        u = normalize(cross(a, w));
        v = normalize(cross(w, u));
    }
}

// RayTracer.cu:232-257 (by-value outputs under literal semantics).
__device__ __forceinline__ void compute_polar_angles(int semantics, uint32_t sampling_method, float& phi, float& theta, float z0, float z1, float z2, float t, float shininess) {
    if (semantics != 1) return;
    switch (sampling_method) {
        case PTC_HEMISPHERE_SAMPLING:
            phi = 2.0f * PT_PIf * z2;
            theta = acosf(z1);
            break;
        case PTC_COSINE_SAMPLING:
            phi = 2.0f * PT_PIf * z2;
            theta = acosf(sqrtf(z1));
            break;
        case PTC_BRDF_SAMPLING:
            phi = 2.0f * PT_PIf * z2;
            if (z0 > t)
                theta = acosf(sqrtf(z1));
            else
                theta = acosf(powf(z1, (1.0f / (shininess + 1.0f))));
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
