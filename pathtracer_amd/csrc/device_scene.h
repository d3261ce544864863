// device_scene.h -- device-side data layout of the path-tracing core.
//
// The host hands over the reference's AoS PODs (136-byte Triangle with an
// embedded 84-byte material, 152-byte Sphere).  On upload they are repacked:
//   * triangle geometry  -> 3 x float4 per triangle in BVH leaf order
//                           (v1|n.x, v2|n.y, v3|n.z): 48 bytes, one aligned
//                           fetch per component, leaves contiguous;
//   * materials          -> de-duplicated table, one 32-bit id per primitive
//                           (bit 31 = objectType LIGHT);
//   * spheres            -> transform + precomputed inverse (128 bytes);
//   * BVH                -> 64-byte binary nodes holding BOTH child boxes.
#pragma once

#include <stdint.h>

#include "ptcore.h"
#include "ptmath.h"

// ---- BVH node: 64 bytes, read as 4 x float4 ---------------------------------
//   q0 = (lo0.x, lo0.y, lo0.z, hi0.x)
//   q1 = (hi0.y, hi0.z, lo1.x, lo1.y)
//   q2 = (lo1.z, hi1.x, hi1.y, hi1.z)
//   q3 = (child0, child1, unused, unused)   [bit patterns]
struct BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    uint32_t child[2];
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BVH node is 64 bytes");

// ---- wide node: 64 bytes, up to four children with 8-bit boxes ----------------
// Built from the binary tree by pulling grandchildren up (largest box first).  A child
// box is stored on a grid of the node's own box: plane = origin + q * scale per axis,
// q in 0..255, scale a power of two (so scale * (1/d) is exact), lower planes rounded
// down and upper planes up.  Unused child slots have an inside-out box (lower planes 255, upper planes 0) that no
// ray enters, and refer to a leaf of one all-zero triangle stored after the scene's (the host-side collapse utility
// of the ABI writes REF_EMPTY_WORD and zero bytes instead; its output is not meant for the kernels).
//   w0 = (origin.x, origin.y, origin.z, scale.x)
//   w1 = (scale.y, scale.z, qlo.x[4], qlo.y[4])      [one byte per child]
//   w2 = (qlo.z[4], qhi.x[4], qhi.y[4], qhi.z[4])
//   w3 = (child0, child1, child2, child3)            internal: index into the wide array
struct WideNode {
    float origin[3];
    float scale_x;
    float scale_y, scale_z;
    uint8_t qlo_x[4], qlo_y[4];
    uint8_t qlo_z[4], qhi_x[4], qhi_y[4], qhi_z[4];
    uint32_t child[4];
};
static_assert(sizeof(WideNode) == 64, "wide BVH node is 64 bytes");

// Child reference encoding.
//   kind (bits 31..30): 0 internal node, 1 triangle leaf, 2 sphere leaf, 3 empty
//   internal: bits 29..0 = node index
//   leaf:     bits 29..27 = count-1 (1..8 primitives), bits 26..0 = first slot
#define REF_KIND(r) ((r) >> 30)
#define REF_INTERNAL 0u
#define REF_TRI 1u
#define REF_SPHERE 2u
#define REF_EMPTY 3u
#define REF_EMPTY_WORD 0xC0000000u
#define REF_NODE(r) ((r)&0x3FFFFFFFu)
#define REF_FIRST(r) ((r)&0x07FFFFFFu)
#define REF_COUNT(r) ((((r) >> 27) & 7u) + 1u)
#define MAKE_LEAF(kind, first, count) (((uint32_t)(kind) << 30) | ((uint32_t)((count)-1u) << 27) | (uint32_t)(first))
#ifndef BVH_LEAF_MAX
#define BVH_LEAF_MAX 3  /* primitives per leaf below the top level (measured: 3 > 2 > 4 > 1 > 6 > 8 with the wide nodes) */
#endif
#define BVH_BIG_MAX 8   /* primitives per top-level "large primitive" leaf */

// Traversal stack: entries per thread kept in LDS, further entries in per-thread
// scratch; the build refuses trees deeper than their sum.
#define TRAV_LDS_STACK 24
#define TRAV_SPILL_STACK 72
#define TRAV_BLOCK 256

struct DevSphere {
    float trans[16];
    float itrans[16];
};

// Hit encoding in the per-path hit buffer: `ref`
//   >= 0 : triangle slot (index into tri_q / tri_mat / tri_prim)
//   -1   : miss
//   <= -2: sphere slot s = -(ref + 2)
#define HIT_MISS (-1)
#define HIT_SPHERE(s) (-(int)(s)-2)
#define HIT_SPHERE_SLOT(ref) (-(ref)-2)

#define MAT_LIGHT_BIT 0x80000000u

struct DevScene {
    // geometry (BVH leaf order)
    const float4* tri_q;        // 3 per triangle slot
    const float4* tri_nm;       // 1 per triangle slot: normal.xyz, bits of tri_mat (the shading side's view of a triangle)
    const uint32_t* tri_mat;    // material id | MAT_LIGHT_BIT
    const int32_t* tri_prim;    // slot -> global primitive index
    const DevSphere* spheres;   // sphere slot order
    const uint32_t* sph_mat;
    const int32_t* sph_prim;    // slot -> global primitive index (T + sphere index)
    const PtcMaterialValue* materials;
    const BvhNode* nodes;       // nodes[0] = root
    const WideNode* wnodes;     // the same hierarchy with four children per node; null when not built
    uint32_t n_tris, n_spheres, n_nodes, n_wnodes;
    uint32_t n_materials;       // entries of `materials`
    // not REF_EMPTY_WORD: the one-ray-per-round kernels test every sphere (one or two) for every ray up front and start the
    // traversal at this reference, the part of the hierarchy that holds everything else
    uint32_t spheres_first_root;
    // big_count != 0: the scene's (at most three) large triangles, slots big_first ..., sit in a leaf of their own under the root;
    // k_surface marks the rays it makes that can hit none of them (the first half of the triangle test fails for each), and the
    // deferred-test kernels start such a ray at the root of the hierarchy without that leaf (big_root_bin in `nodes`,
    // big_root_wide in `wnodes`; REF_EMPTY_WORD: that form has none)
    uint32_t big_first, big_count, big_root_bin, big_root_wide;
    // lights
    const PtcPointLight* plights;
    const PtcDirectionalLight* dlights;
    const PtcQuadLight* qlights;
    uint32_t n_plights, n_dlights, n_qlights;
};

enum { INT_RAYTRACER = 0, INT_ANALYTICDIRECT = 1, INT_DIRECT = 2, INT_PATHTRACER = 3, INT_PATHTRACER2 = 4 };

struct RenderParams {
    PtcConfig cf;
    uint32_t light_samples, light_stratify, sampling_method;
    int32_t integrator;
    float bg[3];
    uint32_t width, height;
    double inv_tiles_x;                // reciprocal of the number of 64x64-pixel blocks per image row
    double inv_batch_samples;          // reciprocal of batch_samples
    uint32_t batch_samples;            // samples of every pixel in the current batch (path numbering)
    uint32_t frame_id;
    uint32_t max_depth_eff;   // min(cf.maxDepth, bounce cap)
    uint32_t bounce_cap;
    int32_t semantics;        // 0 literal, 1 intended
    int32_t seeding;          // 0 reference, 1 counter
    int32_t cull_zero;
    uint32_t init_L;          // this bounce gives every path's radiance sum its first value (camera segments): written, not added to
    uint32_t camera_rays;     // the segments of this bounce are camera segments: origin = the eye, one record (direction | seed) per path
    int32_t brute_force;
    uint32_t samples_per_path;  // shadow-ray samples per segment (integrator dependent)
    uint32_t group_hits;        // group the hits of a bounce by shading class (mixed scenes)
    // k_surface marks the rays it makes that can hit none of the scene's large triangles (DevScene::big_count): their planes,
    // (normal, first vertex), equal ones once; 0 planes = no marking
    uint32_t n_big_planes;
    float big_plane[3][6];
    uint32_t root_light_samples;
    // traversal tuning: idle lanes that trigger a refill, 0 = one ray per thread, 1 = refilling waves + while-while rounds, 2 = refilling waves + single steps,
    // entries a wave reserves per atomic (0 = chosen from the launch size)
    uint32_t tune_refill, tune_mode, tune_shadow, tune_chunk;
};

// Path state.  The LIVE paths of a bounce are kept packed in entries [0, count) of
// o_seed / d_depth / thr / pid / hit_*; the surface kernel writes the survivors, packed
// again, to the `*_next` arrays and the host swaps the two sets.  (With the state indexed by
// path number instead, the live entries thin out from bounce to bounce and every 16-byte
// access drags a mostly dead cache line along.)  `L` is per path number.
struct PathState {
    float4* o_seed;    // origin.xyz, seed of the coming segment (bits)
    float4* d_depth;   // direction.xyz, depth (bits)
    float4* thr;       // throughput.xyz, path number (bits); camera segments have none (throughput 1, path number in `pid`)
    uint32_t* pid;     // path number of the camera segments: ((tile * samples in batch + sample) * 64 + pixel within its 8x8 tile)
    float4* o_seed_next;
    float4* d_depth_next;
    float4* thr_next;
    float4* L;         // radiance accumulated by this path
    uint32_t* g_p;     // segments waiting for their light terms: position in the bounce's path state | flags (emission rule, constant part present)
    float4* g_a;       //   constant part (written only when it is not +0)
    float* hit_t;      // closest-hit distance of the current segment
    int32_t* hit_ref;  // see HIT_*
    float* contrib;    // [light sample][capacity] records of three floats: per-shadow-sample f*G terms
    float4* sh_o;      // queued shadow rays: origin.xyz, tmax
    float4* sh_d;      // direction.xyz, index of the ray's term in `contrib` (bits)
    uint32_t capacity;
};
