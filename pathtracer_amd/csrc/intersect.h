// intersect.h -- primitive tests and BVH traversal (device code).
//
// The primitive tests restate the reference's intersection programs operation
// by operation (Triangle.cu:12-60, Sphere.cu:13-58) so that t/u/v are bit-equal
// to a plain IEEE evaluation of that source; the BVH only decides WHICH
// primitives get tested.  Acceptance follows rtPotentialIntersection:
// tmin < t < current tmax; equal distances are resolved towards the lowest
// global primitive index, which makes the closest hit independent of
// traversal order.
#pragma once

#include "device_scene.h"

struct Cand {
    bool hit;
    float t, u, v;
};

// The first half of the triangle test (Triangle.cu:15-30): the distance to the triangle's plane, and whether it can still be a hit (not
// parallel, in front, inside [t_lo, t_hi] and not below the source's 0.001).  tri_candidate goes on from here; the
// deferred-test kernels run this half alone on the scene's large triangles to decide whether a full test is worth queueing.
__device__ __forceinline__ bool tri_plane_distance(vec3 n0, vec3 v1, vec3 o, vec3 d, float t_lo, float t_hi, float& t) {
    // The source flips the normal when dot(d, n) >= 0 and then forms dot(n, -d), dot(v1 - o, n) and dot(d, n) with
    // the flipped normal.  Negating one factor of every product negates the rounded sum exactly, so these are
    // -|dot(d, n0)|... with the same bits: nothing is recomputed, and the early exits are one test each side of
    // the division instead of a chain of branches.
    const float dn = dot(d, n0);
    const bool flip = dn >= .0f;
    const float den = flip ? -dn : dn;                 // dot(d, n) = -dot(n, -d): <= 0 or NaN
    const float num0 = dot(v1 - o, n0);
    const float num = flip ? -num0 : num0;             // dot(v1 - o, n)
    // nDotWo == 0 is den == 0; a positive numerator over the negative denominator rounds to a value <= -0 < 0.001
    if ((den == 0.0f) | (num > 0.f)) return false;
    t = num / den;
    // `t < 0.001` is a double comparison in the source; 0.001f is the smallest float above the double 0.001, so
    // the float comparison is equivalent (a NaN passes it in the source and fails the interval test here as there)
    return (t >= 0.001f) & (t > t_lo) & (t <= t_hi);
}

// Triangle.cu:15-46.  q0=(v1,n.x) q1=(v2,n.y) q2=(v3,n.z).
// [t_lo, t_hi] is the interval outside of which the caller discards the candidate
// anyway (tmin < t < tmax, and t <= current closest distance): the barycentric part
// is skipped for such t, which cannot change any result.  Likewise a negative
// quotient is known from the signs of numerator and denominator before dividing.
__device__ __forceinline__ Cand tri_candidate(float4 q0, float4 q1, float4 q2, vec3 o, vec3 d, float t_lo, float t_hi) {
    Cand c;
    c.hit = false;
    c.t = c.u = c.v = 0.f;
    const vec3 n0 = mk3(q0.w, q1.w, q2.w);
    vec3 v1 = mk3(q0.x, q0.y, q0.z), v2 = mk3(q1.x, q1.y, q1.z), v3 = mk3(q2.x, q2.y, q2.z);
    float t;
    if (!tri_plane_distance(n0, v1, o, d, t_lo, t_hi, t)) return c;
    const vec3 P = o + t * d;
    vec3 tmp0 = v3 - v1, tmp1 = v2 - v1, tmp2 = P - v1;
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

// Sphere.cu:16-42 with the inverse transform precomputed at upload (same
// cofactor routine, so the same bits).  `epsilon` of Sphere.cu:9 is never set
// by the reference host; it is taken as 0.  Pw receives the world-space point.
__device__ __forceinline__ Cand sphere_candidate(const DevSphere& sp, vec3 o, vec3 d, vec3* Pw) {
    Cand c;
    c.hit = false;
    c.t = c.u = c.v = 0.f;
    const float* it = sp.itrans;
    const float* tr = sp.trans;
    float ohx = it[0] * o.x + it[1] * o.y + it[2] * o.z + it[3] * 1.f;
    float ohy = it[4] * o.x + it[5] * o.y + it[6] * o.z + it[7] * 1.f;
    float ohz = it[8] * o.x + it[9] * o.y + it[10] * o.z + it[11] * 1.f;
    float ohw = it[12] * o.x + it[13] * o.y + it[14] * o.z + it[15] * 1.f;
    vec3 rayOri = mk3(ohx, ohy, ohz) / ohw;
    float dhx = it[0] * d.x + it[1] * d.y + it[2] * d.z + it[3] * 0.f;
    float dhy = it[4] * d.x + it[5] * d.y + it[6] * d.z + it[7] * 0.f;
    float dhz = it[8] * d.x + it[9] * d.y + it[10] * d.z + it[11] * 0.f;
    vec3 rayDir = normalize(mk3(dhx, dhy, dhz));
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
    vec3 P = rayOri + t * rayDir;
    float ihx = tr[0] * P.x + tr[1] * P.y + tr[2] * P.z + tr[3] * 1.f;
    float ihy = tr[4] * P.x + tr[5] * P.y + tr[6] * P.z + tr[7] * 1.f;
    float ihz = tr[8] * P.x + tr[9] * P.y + tr[10] * P.z + tr[11] * 1.f;
    float ihw = tr[12] * P.x + tr[13] * P.y + tr[14] * P.z + tr[15] * 1.f;
    P = mk3(ihx, ihy, ihz) / ihw;
    t = length(P - o);
    c.hit = true;
    c.t = t;
    if (Pw) *Pw = P;
    return c;
}

// Attributes of an accepted hit (Geometries.h:40-45).
struct HitAttr {
    vec3 P, normal, wo;
    uint32_t mat;  // material id | MAT_LIGHT_BIT
};

// SPH = false: the caller knows that the scene has no spheres (the sphere branch is compiled out)
template <bool SPH = true>
__device__ __forceinline__ void hit_attributes(const DevScene& sc, int ref, float t, vec3 o, vec3 d, HitAttr& a) {
    if (!SPH || ref >= 0) {  // Triangle.cu:48-58
        const float4 nm = sc.tri_nm[ref];  // the stored normal and the material word: one record, the vertices are not needed here
        vec3 n = mk3(nm.x, nm.y, nm.z);
        if (dot(d, n) >= .0f) n = -n;
        float nDotWo = dot(n, -d);
        a.P = o + t * d;
        a.wo = -d;
        a.normal = nDotWo > 0 ? n : -n;
        a.mat = __float_as_uint(nm.w);
    } else {  // Sphere.cu:45-56
        int s = HIT_SPHERE_SLOT(ref);
        const DevSphere& sp = sc.spheres[s];
        vec3 P;
        sphere_candidate(sp, o, d, &P);
        const float* it = sp.itrans;
        a.P = P;
        a.wo = -d;
        float tx = it[0] * P.x + it[1] * P.y + it[2] * P.z + it[3] * 1.f;
        float ty = it[4] * P.x + it[5] * P.y + it[6] * P.z + it[7] * 1.f;
        float tz = it[8] * P.x + it[9] * P.y + it[10] * P.z + it[11] * 1.f;
        float tw = it[12] * P.x + it[13] * P.y + it[14] * P.z + it[15] * 1.f;
        vec3 nobj = normalize(mk3(tx, ty, tz) / tw);
        // (inverse)^T * (n,0): column sums of the inverse
        float nx = it[0] * nobj.x + it[4] * nobj.y + it[8] * nobj.z + it[12] * 0.f;
        float ny = it[1] * nobj.x + it[5] * nobj.y + it[9] * nobj.z + it[13] * 0.f;
        float nz = it[2] * nobj.x + it[6] * nobj.y + it[10] * nobj.z + it[14] * 0.f;
        a.normal = normalize(mk3(nx, ny, nz));
        a.mat = sc.sph_mat[s];  // spheres always report objType OBJECT (Sphere.cu:55)
    }
}

// ------------------------------------------------------------- traversal --

struct TravCounters {
    uint32_t nodes, prims;
};

struct ClosestHit {
    float t;
    int ref;       // HIT_* encoding
    int prim;      // global primitive index of the current best, -1 when not loaded yet
};

__device__ __forceinline__ void accept_tri(const DevScene& sc, uint32_t slot, vec3 o, vec3 d, float tmin, float tmax, ClosestHit& best) {
    const float4* q = sc.tri_q + 3 * (size_t)slot;
    Cand c = tri_candidate(q[0], q[1], q[2], o, d, tmin, best.t);
    if (!c.hit || !(c.t > tmin && c.t < tmax)) return;
    if (best.ref != HIT_MISS && !(c.t < best.t)) {
        if (c.t != best.t) return;
        // exact tie: lowest global primitive index wins
        if (best.prim < 0) best.prim = best.ref >= 0 ? sc.tri_prim[best.ref] : sc.sph_prim[HIT_SPHERE_SLOT(best.ref)];
        int prim = sc.tri_prim[slot];
        if (prim > best.prim) return;
        best.prim = prim;
    } else {
        best.prim = -1;
    }
    best.t = c.t;
    best.ref = (int)slot;
}

__device__ __forceinline__ void accept_sphere(const DevScene& sc, uint32_t slot, vec3 o, vec3 d, float tmin, float tmax, ClosestHit& best) {
    Cand c = sphere_candidate(sc.spheres[slot], o, d, nullptr);
    if (!c.hit || !(c.t > tmin && c.t < tmax)) return;
    if (best.ref != HIT_MISS && !(c.t < best.t)) {
        if (c.t != best.t) return;
        if (best.prim < 0) best.prim = best.ref >= 0 ? sc.tri_prim[best.ref] : sc.sph_prim[HIT_SPHERE_SLOT(best.ref)];
        int prim = sc.sph_prim[slot];
        if (prim > best.prim) return;
        best.prim = prim;
    } else {
        best.prim = -1;
    }
    best.t = c.t;
    best.ref = HIT_SPHERE(slot);
}

__device__ __forceinline__ bool any_tri(const DevScene& sc, uint32_t slot, vec3 o, vec3 d, float tmin, float tmax) {
    const float4* q = sc.tri_q + 3 * (size_t)slot;
    Cand c = tri_candidate(q[0], q[1], q[2], o, d, tmin, tmax);
    return c.hit && c.t > tmin && c.t < tmax;
}
__device__ __forceinline__ bool any_sphere(const DevScene& sc, uint32_t slot, vec3 o, vec3 d, float tmin, float tmax) {
    Cand c = sphere_candidate(sc.spheres[slot], o, d, nullptr);
    return c.hit && c.t > tmin && c.t < tmax;
}

__device__ __forceinline__ bool dir_is_nan(vec3 d) { return (d.x != d.x) || (d.y != d.y) || (d.z != d.z); }

// ---- traversal, written as resumable rounds ---------------------------------
// A ray's traversal state is (cur, sp, stack); one ROUND descends through
// internal nodes until a leaf is reached (or the stack runs dry) and then tests
// that leaf's primitives.  The kernels call rounds in a loop and may hand a
// finished lane a new ray between rounds, so that the lanes of a wave spend
// their time in the same one of the two phases ("while-while" schedule).
//   ANY = false: closest hit into `best` (rtTrace, ray type 0)
//   ANY = true : stops at the first accepted primitive (ray type 1: the any-hit
//                program of Common.cu:30-36 terminates the ray)
// `lds_stack` points at this thread's column of a [TRAV_LDS_STACK][TRAV_BLOCK]
// LDS array (entry i at lds_stack[i*TRAV_BLOCK]); deeper entries go to `spill`.
struct TravRay {
    vec3 o, d;
    float ix, iy, iz;     // 1/d (see safe_inv)
    float nox, noy, noz;  // -o/d
    float tmin, tmax;
};

// Reciprocal for the slab tests.  A zero component would give inf and then NaN in
// lo*inv - o*inv; a huge finite value keeps the slab test meaningful, and the error it
// introduces is far inside the padding of the boxes.  The slab arithmetic is allowed to
// contract into FMAs (explicitly): only the primitive tests decide a hit.
__device__ __forceinline__ float safe_inv(float d) { return fabsf(d) > 1e-30f ? 1.0f / d : copysignf(1e30f, d); }

#define REF_DONE REF_EMPTY_WORD

template <int LDS_N = TRAV_LDS_STACK>
__device__ __forceinline__ void trav_push(uint32_t* lds_stack, uint32_t* spill, int& sp, uint32_t v) {
    if (sp < LDS_N) lds_stack[sp * TRAV_BLOCK] = v;
    else spill[sp - LDS_N] = v;
    ++sp;
}

template <int LDS_N = TRAV_LDS_STACK>
__device__ __forceinline__ uint32_t trav_pop(const uint32_t* lds_stack, const uint32_t* spill, int& sp) {
    if (sp == 0) return REF_DONE;
    --sp;
    // The LDS entry is read unconditionally (index clamped) and replaced in the rare deep case: written as one
    // conditional expression the compiler selects between the two ADDRESSES and emits a flat load, which goes
    // through the vector memory pipeline -- the unit these kernels keep busiest -- instead of a ds_read.
    // (a volatile access is the one thing it neither merges nor sinks)
    typedef __attribute__((address_space(3))) const volatile uint32_t lds_word;
    uint32_t v = *(lds_word*)&lds_stack[(sp < LDS_N ? sp : LDS_N - 1) * TRAV_BLOCK];
    if (sp >= LDS_N) v = spill[sp - LDS_N];
    return v;
}

// A ray whose direction has a NaN component can never be accepted by either
// primitive test (every comparison on t fails): it is a miss without traversal.
__device__ __forceinline__ bool trav_init(const DevScene& sc, vec3 o, vec3 d, float tmin, float tmax, TravRay& r, int& sp, uint32_t& cur, ClosestHit& best) {
    r.o = o;
    r.d = d;
    r.ix = safe_inv(d.x);
    r.iy = safe_inv(d.y);
    r.iz = safe_inv(d.z);
    r.nox = -(o.x * r.ix);
    r.noy = -(o.y * r.iy);
    r.noz = -(o.z * r.iz);
    r.tmin = tmin;
    r.tmax = tmax;
    best.t = tmax;
    best.ref = HIT_MISS;
    best.prim = -1;
    sp = 0;
    cur = 0;  // internal node 0 = root
    return !(dir_is_nan(d) || sc.n_nodes == 0);
}

// The 64 bytes of an internal node.
struct NodeData {
    float4 a, b, c;
    uint4 ch;
};

__device__ __forceinline__ NodeData node_fetch(const DevScene& sc, uint32_t cur) {
    const float4* n4 = reinterpret_cast<const float4*>(sc.nodes + REF_NODE(cur));
    NodeData n;
    n.a = n4[0];
    n.b = n4[1];
    n.c = n4[2];
    n.ch = *reinterpret_cast<const uint4*>(n4 + 3);
    return n;
}

// Slab-test both child boxes against [tmin, lim], go to the nearer hit child, push the
// farther one.
template <int LDS_N = TRAV_LDS_STACK>
__device__ __forceinline__ uint32_t node_eval(const NodeData& nd, const TravRay& r, float lim, uint32_t* lds_stack, uint32_t* spill, int& sp) {
    const float4 a = nd.a, b = nd.b, c = nd.c;
    const uint4 ch = nd.ch;
    // child 0: lo=(a.x,a.y,a.z) hi=(a.w,b.x,b.y)
    float t0 = __builtin_fmaf(a.x, r.ix, r.nox), t1 = __builtin_fmaf(a.w, r.ix, r.nox);
    float n0 = fminf(t0, t1), f0 = fmaxf(t0, t1);
    t0 = __builtin_fmaf(a.y, r.iy, r.noy); t1 = __builtin_fmaf(b.x, r.iy, r.noy);
    n0 = fmaxf(n0, fminf(t0, t1)); f0 = fminf(f0, fmaxf(t0, t1));
    t0 = __builtin_fmaf(a.z, r.iz, r.noz); t1 = __builtin_fmaf(b.y, r.iz, r.noz);
    n0 = fmaxf(n0, fminf(t0, t1)); f0 = fminf(f0, fmaxf(t0, t1));
    n0 = fmaxf(n0, r.tmin); f0 = fminf(f0, lim);
    // child 1: lo=(b.z,b.w,c.x) hi=(c.y,c.z,c.w)
    t0 = __builtin_fmaf(b.z, r.ix, r.nox); t1 = __builtin_fmaf(c.y, r.ix, r.nox);
    float n1 = fminf(t0, t1), f1 = fmaxf(t0, t1);
    t0 = __builtin_fmaf(b.w, r.iy, r.noy); t1 = __builtin_fmaf(c.z, r.iy, r.noy);
    n1 = fmaxf(n1, fminf(t0, t1)); f1 = fminf(f1, fmaxf(t0, t1));
    t0 = __builtin_fmaf(c.x, r.iz, r.noz); t1 = __builtin_fmaf(c.w, r.iz, r.noz);
    n1 = fmaxf(n1, fminf(t0, t1)); f1 = fminf(f1, fmaxf(t0, t1));
    n1 = fmaxf(n1, r.tmin); f1 = fminf(f1, lim);
    const bool h0 = (n0 <= f0) && (REF_KIND(ch.x) != REF_EMPTY);
    const bool h1 = (n1 <= f1) && (REF_KIND(ch.y) != REF_EMPTY);
    if (h0 && h1) {
        const bool swap = n1 < n0;
        trav_push<LDS_N>(lds_stack, spill, sp, swap ? ch.x : ch.y);
        return swap ? ch.y : ch.x;
    }
    if (h0) return ch.x;
    if (h1) return ch.y;
    return trav_pop<LDS_N>(lds_stack, spill, sp);
}


// ---- four-wide nodes ----------------------------------------------------------
// The box part of a wide node; the child references (w3) are read by wide_eval itself, from
// the cache line this fetch brought in: they are needed last, and four registers fewer
// stay occupied while the fetch is in flight.
struct WideData {
    uint4 w0, w1, w2;
    const uint4* refs;
};

__device__ __forceinline__ WideData wide_fetch(const DevScene& sc, uint32_t cur) {
    const uint4* p = reinterpret_cast<const uint4*>(sc.wnodes + REF_NODE(cur));
    WideData n;
    n.w0 = p[0];
    n.w1 = p[1];
    n.w2 = p[2];
    n.refs = p + 3;
    return n;
}

__device__ __forceinline__ float byte_f(uint32_t w, int c) { return (float)((w >> (8 * c)) & 0xFFu); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }

// Order two (entry distance, child) pairs by distance; the distances are non-negative
// floats or 0xFFFFFFFF (not hit), so their bit patterns order like unsigned integers.
__device__ __forceinline__ void wide_order(uint32_t& ka, uint32_t& ra, uint32_t& kb, uint32_t& rb) {
    const bool sw = kb < ka;
    const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb, r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0;
    kb = k1;
    ra = r0;
    rb = r1;
}

// Slab-test the (up to) four child boxes against [tmin, lim]; returns the child to visit
// next and pushes the other hit children.  SORT: nearest first (closest-hit rays); without
// it the children go in storage order (any-hit rays, where every hit child has to be
// visited anyway unless an occluder turns up).
template <bool SORT, int LDS_N = TRAV_LDS_STACK>
__device__ __forceinline__ uint32_t wide_eval(const WideData& nd, const TravRay& r, float lim, uint32_t* lds_stack, uint32_t* spill, int& sp) {
    // plane distance of grid coordinate q along x: (origin.x + q*scale.x - o.x)/d.x = q*sx + bx, bx = origin.x/d.x - o.x/d.x
    const float sx = __uint_as_float(nd.w0.w) * r.ix, sy = __uint_as_float(nd.w1.x) * r.iy, sz = __uint_as_float(nd.w1.y) * r.iz;
    // -o/d is kept in registers: one FMA per axis (the boxes are padded for the rounding of o/d, as for the two-box nodes)
    const float bx = __builtin_fmaf(__uint_as_float(nd.w0.x), r.ix, r.nox), by = __builtin_fmaf(__uint_as_float(nd.w0.y), r.iy, r.noy),
                bz = __builtin_fmaf(__uint_as_float(nd.w0.z), r.iz, r.noz);
    // the planes the ray enters / leaves through, per axis, for all four children at once
    const bool px = r.ix >= 0.f, py = r.iy >= 0.f, pz = r.iz >= 0.f;
    const uint32_t nx = px ? nd.w1.z : nd.w2.y, fx = px ? nd.w2.y : nd.w1.z;
    const uint32_t ny = py ? nd.w1.w : nd.w2.z, fy = py ? nd.w2.z : nd.w1.w;
    const uint32_t nz = pz ? nd.w2.x : nd.w2.w, fz = pz ? nd.w2.w : nd.w2.x;
    const uint4 w3 = *nd.refs;
    uint32_t key[4], ref[4] = {w3.x, w3.y, w3.z, w3.w};
    bool hit[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tn = fmaxf(max3f(__builtin_fmaf(byte_f(nx, c), sx, bx), __builtin_fmaf(byte_f(ny, c), sy, by), __builtin_fmaf(byte_f(nz, c), sz, bz)), r.tmin);
        const float tf = fminf(min3f(__builtin_fmaf(byte_f(fx, c), sx, bx), __builtin_fmaf(byte_f(fy, c), sy, by), __builtin_fmaf(byte_f(fz, c), sz, bz)), lim);
        hit[c] = tn <= tf;  // an unused child slot has an inside-out box (and a reference to a triangle no ray hits)
        key[c] = hit[c] ? __float_as_uint(tn) : 0xFFFFFFFFu;
    }
    if (SORT) {
        wide_order(key[0], ref[0], key[1], ref[1]);
        wide_order(key[2], ref[2], key[3], ref[3]);
        wide_order(key[0], ref[0], key[2], ref[2]);
        wide_order(key[1], ref[1], key[3], ref[3]);
        wide_order(key[1], ref[1], key[2], ref[2]);
        // nearest in [0]; farther ones are pushed first
        if (sp + 3 <= LDS_N) {  // no branches: a child that was not hit is written above the top and not counted
            lds_stack[sp * TRAV_BLOCK] = ref[3];
            sp += key[3] != 0xFFFFFFFFu ? 1 : 0;
            lds_stack[sp * TRAV_BLOCK] = ref[2];
            sp += key[2] != 0xFFFFFFFFu ? 1 : 0;
            lds_stack[sp * TRAV_BLOCK] = ref[1];
            sp += key[1] != 0xFFFFFFFFu ? 1 : 0;
        } else {
            if (key[3] != 0xFFFFFFFFu) trav_push<LDS_N>(lds_stack, spill, sp, ref[3]);
            if (key[2] != 0xFFFFFFFFu) trav_push<LDS_N>(lds_stack, spill, sp, ref[2]);
            if (key[1] != 0xFFFFFFFFu) trav_push<LDS_N>(lds_stack, spill, sp, ref[1]);
        }
        if (key[0] != 0xFFFFFFFFu) return ref[0];
        return trav_pop<LDS_N>(lds_stack, spill, sp);
    }
    // storage order: the first hit child is visited next, the later ones are pushed last to first
    const bool p3 = hit[3] & (hit[0] | hit[1] | hit[2]), p2 = hit[2] & (hit[0] | hit[1]), p1 = hit[1] & hit[0];
    if (sp + 3 <= LDS_N) {  // the same branch-free pushes as above
        lds_stack[sp * TRAV_BLOCK] = ref[3];
        sp += p3 ? 1 : 0;
        lds_stack[sp * TRAV_BLOCK] = ref[2];
        sp += p2 ? 1 : 0;
        lds_stack[sp * TRAV_BLOCK] = ref[1];
        sp += p1 ? 1 : 0;
    } else {
        if (p3) trav_push<LDS_N>(lds_stack, spill, sp, ref[3]);
        if (p2) trav_push<LDS_N>(lds_stack, spill, sp, ref[2]);
        if (p1) trav_push<LDS_N>(lds_stack, spill, sp, ref[1]);
    }
    if (hit[0] | hit[1] | hit[2] | hit[3]) return hit[0] ? ref[0] : hit[1] ? ref[1] : hit[2] ? ref[2] : ref[3];
    return trav_pop<LDS_N>(lds_stack, spill, sp);
}

// Visit one internal node.
template <bool COUNT, int LDS_N = TRAV_LDS_STACK>
__device__ __forceinline__ uint32_t trav_node(const DevScene& sc, const TravRay& r, uint32_t cur, float lim, uint32_t* lds_stack, uint32_t* spill, int& sp,
                                              TravCounters& cnt) {
    if (COUNT) cnt.nodes++;
    return node_eval<LDS_N>(node_fetch(sc, cur), r, lim, lds_stack, spill, sp);
}

// One single step (either one internal node or one leaf): the "if-if" schedule.
template <bool ANY, bool COUNT, int LDS_N = TRAV_LDS_STACK>
__device__ __forceinline__ bool trav_step(const DevScene& sc, const TravRay& r, uint32_t* lds_stack, uint32_t* spill, int& sp, uint32_t& cur,
                                          ClosestHit& best, TravCounters& cnt, bool& occluded);

// One round; returns true when the ray is finished (for ANY also when occluded,
// reported through `occluded`).
template <bool ANY, bool COUNT, int LDS_N = TRAV_LDS_STACK>
__device__ __forceinline__ bool trav_round(const DevScene& sc, const TravRay& r, uint32_t* lds_stack, uint32_t* spill, int& sp, uint32_t& cur,
                                           ClosestHit& best, TravCounters& cnt, bool& occluded) {
    while (REF_KIND(cur) == REF_INTERNAL) cur = trav_node<COUNT, LDS_N>(sc, r, cur, ANY ? r.tmax : best.t, lds_stack, spill, sp, cnt);
    const uint32_t kind = REF_KIND(cur);
    if (kind == REF_TRI) {
        const uint32_t first = REF_FIRST(cur), count = REF_COUNT(cur);
        for (uint32_t k = 0; k < count; ++k) {
            if (COUNT) cnt.prims++;
            if (ANY) {
                if (any_tri(sc, first + k, r.o, r.d, r.tmin, r.tmax)) {
                    occluded = true;
                    return true;
                }
            } else {
                accept_tri(sc, first + k, r.o, r.d, r.tmin, r.tmax, best);
            }
        }
        cur = trav_pop<LDS_N>(lds_stack, spill, sp);
    } else if (kind == REF_SPHERE) {
        const uint32_t first = REF_FIRST(cur), count = REF_COUNT(cur);
        for (uint32_t k = 0; k < count; ++k) {
            if (COUNT) cnt.prims++;
            if (ANY) {
                if (any_sphere(sc, first + k, r.o, r.d, r.tmin, r.tmax)) {
                    occluded = true;
                    return true;
                }
            } else {
                accept_sphere(sc, first + k, r.o, r.d, r.tmin, r.tmax, best);
            }
        }
        cur = trav_pop<LDS_N>(lds_stack, spill, sp);
    }
    return cur == REF_DONE;
}

template <bool ANY, bool COUNT, int LDS_N>
__device__ __forceinline__ bool trav_step(const DevScene& sc, const TravRay& r, uint32_t* lds_stack, uint32_t* spill, int& sp, uint32_t& cur,
                                          ClosestHit& best, TravCounters& cnt, bool& occluded) {
    if (REF_KIND(cur) == REF_INTERNAL) {
        cur = trav_node<COUNT, LDS_N>(sc, r, cur, ANY ? r.tmax : best.t, lds_stack, spill, sp, cnt);
        return cur == REF_DONE;
    }
    return trav_round<ANY, COUNT, LDS_N>(sc, r, lds_stack, spill, sp, cur, best, cnt, occluded);
}

// The same traversal as one tight loop for a thread that owns a single ray.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse_simple(const DevScene& sc, vec3 o, vec3 d, float tmin, float tmax, uint32_t* lds_stack, ClosestHit& best, TravCounters& cnt) {
    best.t = tmax;
    best.ref = HIT_MISS;
    best.prim = -1;
    if (dir_is_nan(d) || sc.n_nodes == 0) return false;
    const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
    const float nox = -(o.x * ix), noy = -(o.y * iy), noz = -(o.z * iz);
    uint32_t spill[TRAV_SPILL_STACK];
    int sp = 0;
    uint32_t cur = 0;  // internal node 0 = root
    for (;;) {
        const uint32_t kind = REF_KIND(cur);
        if (kind == REF_INTERNAL) {
            const float4* n4 = reinterpret_cast<const float4*>(sc.nodes + REF_NODE(cur));
            const float4 a = n4[0], b = n4[1], c = n4[2];
            const uint4 ch = *reinterpret_cast<const uint4*>(n4 + 3);
            if (COUNT) cnt.nodes++;
            const float lim = ANY ? tmax : best.t;
            // child 0: lo=(a.x,a.y,a.z) hi=(a.w,b.x,b.y)
            float t0 = __builtin_fmaf(a.x, ix, nox), t1 = __builtin_fmaf(a.w, ix, nox);
            float n0 = fminf(t0, t1), f0 = fmaxf(t0, t1);
            t0 = __builtin_fmaf(a.y, iy, noy); t1 = __builtin_fmaf(b.x, iy, noy);
            n0 = fmaxf(n0, fminf(t0, t1)); f0 = fminf(f0, fmaxf(t0, t1));
            t0 = __builtin_fmaf(a.z, iz, noz); t1 = __builtin_fmaf(b.y, iz, noz);
            n0 = fmaxf(n0, fminf(t0, t1)); f0 = fminf(f0, fmaxf(t0, t1));
            n0 = fmaxf(n0, tmin); f0 = fminf(f0, lim);
            // child 1: lo=(b.z,b.w,c.x) hi=(c.y,c.z,c.w)
            t0 = __builtin_fmaf(b.z, ix, nox); t1 = __builtin_fmaf(c.y, ix, nox);
            float n1 = fminf(t0, t1), f1 = fmaxf(t0, t1);
            t0 = __builtin_fmaf(b.w, iy, noy); t1 = __builtin_fmaf(c.z, iy, noy);
            n1 = fmaxf(n1, fminf(t0, t1)); f1 = fminf(f1, fmaxf(t0, t1));
            t0 = __builtin_fmaf(c.x, iz, noz); t1 = __builtin_fmaf(c.w, iz, noz);
            n1 = fmaxf(n1, fminf(t0, t1)); f1 = fminf(f1, fmaxf(t0, t1));
            n1 = fmaxf(n1, tmin); f1 = fminf(f1, lim);
            const bool h0 = (n0 <= f0) && (REF_KIND(ch.x) != REF_EMPTY);
            const bool h1 = (n1 <= f1) && (REF_KIND(ch.y) != REF_EMPTY);
            if (h0 && h1) {
                const bool swap = n1 < n0;
                const uint32_t nearc = swap ? ch.y : ch.x, farc = swap ? ch.x : ch.y;
                if (sp < TRAV_LDS_STACK) lds_stack[sp * TRAV_BLOCK] = farc;
                else spill[sp - TRAV_LDS_STACK] = farc;
                ++sp;
                cur = nearc;
                continue;
            } else if (h0) {
                cur = ch.x;
                continue;
            } else if (h1) {
                cur = ch.y;
                continue;
            }
        } else if (kind == REF_TRI) {
            const uint32_t first = REF_FIRST(cur), count = REF_COUNT(cur);
            for (uint32_t k = 0; k < count; ++k) {
                if (COUNT) cnt.prims++;
                if (ANY) {
                    if (any_tri(sc, first + k, o, d, tmin, tmax)) return true;
                } else {
                    accept_tri(sc, first + k, o, d, tmin, tmax, best);
                }
            }
        } else if (kind == REF_SPHERE) {
            const uint32_t first = REF_FIRST(cur), count = REF_COUNT(cur);
            for (uint32_t k = 0; k < count; ++k) {
                if (COUNT) cnt.prims++;
                if (ANY) {
                    if (any_sphere(sc, first + k, o, d, tmin, tmax)) return true;
                } else {
                    accept_sphere(sc, first + k, o, d, tmin, tmax, best);
                }
            }
        }
        if (sp == 0) break;
        --sp;
        {  // see trav_pop
            typedef __attribute__((address_space(3))) const volatile uint32_t lds_word;
            cur = *(lds_word*)&lds_stack[(sp < TRAV_LDS_STACK ? sp : TRAV_LDS_STACK - 1) * TRAV_BLOCK];
            if (sp >= TRAV_LDS_STACK) cur = spill[sp - TRAV_LDS_STACK];
        }
    }
    return false;
}

// Brute force over every primitive (parity aid; also the reference point the BVH
// path is compared with).
template <bool ANY>
__device__ __forceinline__ bool brute_force(const DevScene& sc, vec3 o, vec3 d, float tmin, float tmax, ClosestHit& best) {
    best.t = tmax;
    best.ref = HIT_MISS;
    best.prim = -1;
    for (uint32_t s = 0; s < sc.n_tris; ++s) {
        if (ANY) {
            if (any_tri(sc, s, o, d, tmin, tmax)) return true;
        } else {
            accept_tri(sc, s, o, d, tmin, tmax, best);
        }
    }
    for (uint32_t s = 0; s < sc.n_spheres; ++s) {
        if (ANY) {
            if (any_sphere(sc, s, o, d, tmin, tmax)) return true;
        } else {
            accept_sphere(sc, s, o, d, tmin, tmax, best);
        }
    }
    return false;
}
