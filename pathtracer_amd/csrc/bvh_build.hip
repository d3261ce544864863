// bvh_build.hip -- acceleration-structure build on the GPU.
//
// Stands in for the three closed-source "Trbvh" builds of the reference
// (Renderer.cpp:202-217).  Primitive boxes come from the reference's bounding
// programs (Triangle.cu:62-72, Sphere.cu:60-73).  Per primitive kind:
//   1. primitive boxes                                   (k_bounds_*)
//   2. primitives much larger than the rest of the scene (a ground plane under
//      a detailed mesh) go into one top-level leaf instead of the tree: a
//      Morton-ordered tree would otherwise inflate every ancestor box
//   3. 63-bit Morton codes of the remaining centroids, radix sort (rocPRIM)
//   4. Karras' binary radix tree, one thread per internal node
//   5. bottom-up boxes, one launch per tree level (no intra-launch hand-off)
//   6. subtrees of <= BVH_LEAF_MAX primitives collapse into leaves; the kept
//      nodes are renumbered densely and written as 64-byte two-box nodes
// Triangles are repacked into leaf order as 3 x float4.  Every box is padded by
// a few dozen ulps of the scene's largest coordinate so that the box test can
// never reject a primitive the exact primitive test would accept.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "bvh_device.h"
#include "wavefront.h"

#define BB 256
#define HIPCHK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) return std::string(#x) + ": " + hipGetErrorString(e_);     \
    } while (0)

namespace {

__global__ void k_bounds_tri(const PtcTriangle* __restrict__ tris, uint32_t n, Box6* __restrict__ out) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const PtcTriangle& t = tris[i];
    Box6 b;  // Triangle.cu:66-71
    b.lo[0] = fminf(fminf(t.v1.x, t.v2.x), t.v3.x);
    b.lo[1] = fminf(fminf(t.v1.y, t.v2.y), t.v3.y);
    b.lo[2] = fminf(fminf(t.v1.z, t.v2.z), t.v3.z);
    b.hi[0] = fmaxf(fmaxf(t.v1.x, t.v2.x), t.v3.x);
    b.hi[1] = fmaxf(fmaxf(t.v1.y, t.v2.y), t.v3.y);
    b.hi[2] = fmaxf(fmaxf(t.v1.z, t.v2.z), t.v3.z);
    out[i] = b;
}

__global__ void k_bounds_sphere(const PtcSphere* __restrict__ sph, uint32_t n, Box6* __restrict__ out) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const float* m = sph[i].trans;  // Sphere.cu:63-72
    const float x = length(mk3(m[0], m[1], m[2])), y = length(mk3(m[4], m[5], m[6])), z = length(mk3(m[8], m[9], m[10]));
    Box6 b;
    b.lo[0] = m[3] - x; b.lo[1] = m[7] - y; b.lo[2] = m[11] - z;
    b.hi[0] = m[3] + x; b.hi[1] = m[7] + y; b.hi[2] = m[11] + z;
    out[i] = b;
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // 21 bits -> every third bit
    uint64_t x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(const Box6* __restrict__ boxes, const uint32_t* __restrict__ ids, uint32_t n, float3 cmin, float3 cinv,
                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const uint32_t prim = ids[i];
    const Box6 b = boxes[prim];
    const float cx = (0.5f * (b.lo[0] + b.hi[0]) - cmin.x) * cinv.x;
    const float cy = (0.5f * (b.lo[1] + b.hi[1]) - cmin.y) * cinv.y;
    const float cz = (0.5f * (b.lo[2] + b.hi[2]) - cmin.z) * cinv.z;
    const float scale = 2097151.0f;  // 2^21 - 1
    const uint32_t qx = (uint32_t)fminf(fmaxf(cx * scale, 0.f), scale);
    const uint32_t qy = (uint32_t)fminf(fmaxf(cy * scale, 0.f), scale);
    const uint32_t qz = (uint32_t)fminf(fmaxf(cz * scale, 0.f), scale);
    keys[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
    vals[i] = prim;
}

// Length of the common prefix of keys i and j; equal keys are told apart by
// their position so that the tree stays a proper binary tree.
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz(i ^ j);
    return __clzll((long long)(a ^ b));
}

// Tree arrays: internal nodes 0..n-2, leaves addressed as n-1+leaf in `parent`
// and as (leaf | LEAF_FLAG) in the child arrays.
#define LEAF_FLAG 0x80000000u

__global__ void k_karras(const uint64_t* __restrict__ keys, int n, uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                         int* __restrict__ first, int* __restrict__ last, int* __restrict__ parent) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    first[i] = lo;
    last[i] = hi;
    if (lo == gamma) {
        left[i] = (uint32_t)gamma | LEAF_FLAG;
        parent[n - 1 + gamma] = i;
    } else {
        left[i] = (uint32_t)gamma;
        parent[gamma] = i;
    }
    if (hi == gamma + 1) {
        right[i] = (uint32_t)(gamma + 1) | LEAF_FLAG;
        parent[n - 1 + gamma + 1] = i;
    } else {
        right[i] = (uint32_t)(gamma + 1);
        parent[gamma + 1] = i;
    }
    if (i == 0) parent[0] = -1;
}

__global__ void k_leaf_boxes(const Box6* __restrict__ boxes, const uint32_t* __restrict__ sorted_prims, int n, float pad, Box6* __restrict__ node_box /* [2n-1] */) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    Box6 b = boxes[sorted_prims[i]];
    for (int k = 0; k < 3; ++k) {
        b.lo[k] -= pad;
        b.hi[k] += pad;
    }
    node_box[n - 1 + i] = b;
}

__global__ void k_depth(const int* __restrict__ parent, int n, uint32_t* __restrict__ max_depth) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    uint32_t d = 0;
    int cur = parent[n - 1 + i];
    while (cur >= 0 && d < 4096u) {
        ++d;
        cur = parent[cur];
    }
    atomicMax(max_depth, d);
}

// One level of the bottom-up box pass: a node is finished in pass `pass` when both
// children were finished in an earlier pass (done[] holds the finishing pass, 0 = not
// yet; leaves count as finished before pass 1).
__global__ void k_refit_pass(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, int n, int pass, int* __restrict__ done,
                             Box6* __restrict__ node_box) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    if (done[i] != 0) return;
    const uint32_t l = left[i], r = right[i];
    const bool lready = (l & LEAF_FLAG) || (done[l] != 0 && done[l] < pass);
    const bool rready = (r & LEAF_FLAG) || (done[r] != 0 && done[r] < pass);
    if (!lready || !rready) return;
    const Box6 a = node_box[(l & LEAF_FLAG) ? n - 1 + (int)(l & ~LEAF_FLAG) : (int)l];
    const Box6 b = node_box[(r & LEAF_FLAG) ? n - 1 + (int)(r & ~LEAF_FLAG) : (int)r];
    Box6 u;
    for (int k = 0; k < 3; ++k) {
        u.lo[k] = fminf(a.lo[k], b.lo[k]);
        u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
    node_box[i] = u;
    done[i] = pass;
}

__global__ void k_emit(const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, const int* __restrict__ first, const int* __restrict__ last,
                       const int* __restrict__ newidx /* -1 = collapsed */, const Box6* __restrict__ node_box, int n, uint32_t node_base, uint32_t kind,
                       BvhNode* __restrict__ nodes) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    if (newidx[i] < 0) return;
    BvhNode out;
    const uint32_t ch[2] = {left[i], right[i]};
    for (int c = 0; c < 2; ++c) {
        const uint32_t r = ch[c];
        Box6 b;
        uint32_t ref;
        if (r & LEAF_FLAG) {
            const int leaf = (int)(r & ~LEAF_FLAG);
            b = node_box[n - 1 + leaf];
            ref = MAKE_LEAF(kind, leaf, 1u);
        } else {
            b = node_box[r];
            if (newidx[r] >= 0) ref = node_base + (uint32_t)newidx[r];
            else ref = MAKE_LEAF(kind, first[r], (uint32_t)(last[r] - first[r] + 1));
        }
        float* lo = c == 0 ? out.lo0 : out.lo1;
        float* hi = c == 0 ? out.hi0 : out.hi1;
        for (int k = 0; k < 3; ++k) {
            lo[k] = b.lo[k];
            hi[k] = b.hi[k];
        }
        out.child[c] = ref;
    }
    out.pad[0] = out.pad[1] = 0u;
    nodes[node_base + (uint32_t)newidx[i]] = out;
}


// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ------
// Clusters live in Morton order.  Every iteration each cluster looks `radius`
// positions to either side for the neighbour whose union box has the smallest
// area; mutual nearest neighbours merge into a new node; the cluster array is
// compacted (order preserved).  Tree nodes: leaves 0..n-1 (sorted position),
// internal nodes n..2n-2 in creation order.
__device__ __forceinline__ float union_area(const Box6& a, const Box6& b) {
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_ploc_init(const Box6* __restrict__ boxes, const uint32_t* __restrict__ sorted_prims, int n, float pad, Box6* __restrict__ node_box,
                            int* __restrict__ cluster) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    Box6 b = boxes[sorted_prims[i]];
    for (int k = 0; k < 3; ++k) {
        b.lo[k] -= pad;
        b.hi[k] += pad;
    }
    node_box[i] = b;
    cluster[i] = i;
}

__global__ void k_ploc_nn(const int* __restrict__ cluster, int m, const Box6* __restrict__ node_box, int radius, int* __restrict__ nn) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= m) return;
    const Box6 me = node_box[cluster[i]];
    float best = INFINITY;
    int bj = -1;
    const int lo = max(0, i - radius), hi = min(m - 1, i + radius);
    for (int j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const float a = union_area(me, node_box[cluster[j]]);
        // Ties go to the nearest position, and between the two nearest to the partner i^1: a run of coincident
        // primitives then pairs up completely in every round (a balanced subtree) instead of merging one pair per
        // round (a chain as deep as the run is long).
        const int dj = abs(j - i), db = abs(bj - i);
        if (a < best || (a == best && bj >= 0 && (dj < db || (dj == db && j == (i ^ 1))))) {
            best = a;
            bj = j;
        }
    }
    // no neighbour with a finite union area (astronomically large or non-finite boxes): pair up by
    // position, which still merges mutual pairs and so keeps the clustering moving
    if (bj < 0 && m > 1) bj = (i ^ 1) < m ? (i ^ 1) : i - 1;
    nn[i] = bj;
}

__global__ void k_ploc_merge(const int* __restrict__ cluster, const int* __restrict__ nn, int m, int n, Box6* __restrict__ node_box,
                             int* __restrict__ left, int* __restrict__ right, int* __restrict__ node_count, int* __restrict__ cluster_out,
                             int* __restrict__ valid) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    if (j >= 0 && nn[j] == i) {
        if (i < j) {
            const int idx = atomicAdd(node_count, 1);  // internal node n + idx
            const int a = cluster[i], b = cluster[j];
            left[idx] = a;
            right[idx] = b;
            const Box6 ba = node_box[a], bb = node_box[b];
            Box6 u;
            for (int k = 0; k < 3; ++k) {
                u.lo[k] = fminf(ba.lo[k], bb.lo[k]);
                u.hi[k] = fmaxf(ba.hi[k], bb.hi[k]);
            }
            node_box[n + idx] = u;
            cluster_out[i] = n + idx;
            valid[i] = 1;
        } else {
            valid[i] = 0;
        }
    } else {
        cluster_out[i] = cluster[i];
        valid[i] = 1;
    }
}

__global__ void k_ploc_scatter(const int* __restrict__ cluster_out, const int* __restrict__ valid, const int* __restrict__ pos, int m,
                               int* __restrict__ cluster_next) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= m) return;
    if (valid[i]) cluster_next[pos[i]] = cluster_out[i];
}

__global__ void k_pack_tris(const PtcTriangle* __restrict__ tris, const int32_t* __restrict__ slot_prim, const uint32_t* __restrict__ tri_mat, uint32_t n,
                            float4* __restrict__ tri_q, float4* __restrict__ tri_nm) {
    const uint32_t s = blockIdx.x * BB + threadIdx.x;
    if (s >= n) return;
    const PtcTriangle& t = tris[slot_prim[s]];
    tri_nm[s] = make_float4(t.normal.x, t.normal.y, t.normal.z, __uint_as_float(tri_mat[s]));
    tri_q[3 * (size_t)s + 0] = make_float4(t.v1.x, t.v1.y, t.v1.z, t.normal.x);
    tri_q[3 * (size_t)s + 1] = make_float4(t.v2.x, t.v2.y, t.v2.z, t.normal.y);
    tri_q[3 * (size_t)s + 2] = make_float4(t.v3.x, t.v3.y, t.v3.z, t.normal.z);
}

inline uint32_t nblocks(size_t n) { return (uint32_t)((n + BB - 1) / BB); }

float maxExtent(const Box6& b) { return std::max(b.hi[0] - b.lo[0], std::max(b.hi[1] - b.lo[1], b.hi[2] - b.lo[2])); }

Box6 emptyBox() {
    Box6 b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = INFINITY;
        b.hi[k] = -INFINITY;
    }
    return b;
}
void grow(Box6& a, const Box6& b) {
    for (int k = 0; k < 3; ++k) {
        a.lo[k] = std::min(a.lo[k], b.lo[k]);
        a.hi[k] = std::max(a.hi[k], b.hi[k]);
    }
}

// Split the primitives of one kind into "small" (tree) and "big" (top-level leaf).
void classify(const std::vector<Box6>& boxes, std::vector<uint32_t>& small, std::vector<uint32_t>& big) {
    const size_t n = boxes.size();
    small.clear();
    big.clear();
    if (n <= BVH_BIG_MAX) {
        for (uint32_t i = 0; i < n; ++i) big.push_back(i);
        return;
    }
    std::vector<char> isBig(n, 0);
    for (int round = 0; round < 4; ++round) {
        Box6 u = emptyBox();
        for (size_t i = 0; i < n; ++i)
            if (!isBig[i]) grow(u, boxes[i]);
        const float E = maxExtent(u);
        std::vector<uint32_t> cand;
        for (uint32_t i = 0; i < n; ++i)
            if (!isBig[i] && maxExtent(boxes[i]) > 0.5f * E) cand.push_back(i);
        if (cand.empty()) break;
        // at most BVH_BIG_MAX in total, largest first; never empty the small set
        std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return maxExtent(boxes[a]) > maxExtent(boxes[b]); });
        size_t nbig = 0;
        for (char c : isBig) nbig += c;
        bool added = false;
        for (uint32_t c : cand) {
            if (nbig >= BVH_BIG_MAX) break;
            isBig[c] = 1;
            ++nbig;
            added = true;
        }
        if (!added || nbig >= BVH_BIG_MAX) break;
    }
    for (uint32_t i = 0; i < n; ++i) (isBig[i] ? big : small).push_back(i);
}

// Adds the wall-clock time of a host-side pass to a total (BvhBuildOutput::host_ms).
struct HostPass {
    double& total;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit HostPass(double& t) : total(t) {}
    ~HostPass() { total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

struct KindResult {
    std::vector<int32_t> slot_prim;  // slot -> primitive index within its kind
    uint32_t n_small = 0, n_big = 0;
    uint32_t n_tree_nodes = 0;       // emitted nodes written at node_base..
    bool has_tree = false;
    Box6 tree_box, small_box, big_box;
    uint32_t max_depth = 0;
    double host_ms = 0.0;
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};


// Binary tree (leaves 0..n-1, internal n..2n-2 with left/right) -> traversal nodes.
// Subtrees of <= BVH_LEAF_MAX primitives become leaves whose primitives get
// consecutive slots; kept nodes are numbered so that the two children of a node
// sit next to each other.  `order` receives the leaf index of every slot.
void emitTree(const std::vector<int>& left, const std::vector<int>& right, const std::vector<Box6>& box, int n, int root, uint32_t kind, uint32_t node_base,
              std::vector<BvhNode>& out, std::vector<int>& order, uint32_t& max_depth) {
    std::vector<int> size(2 * (size_t)n - 1, 1);
    {  // subtree sizes, children before parents
        std::vector<int> stack{root}, post;
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            post.push_back(v);
            if (v >= n) {
                stack.push_back(left[v - n]);
                stack.push_back(right[v - n]);
            }
        }
        for (size_t k = post.size(); k-- > 0;) {
            const int v = post[k];
            if (v >= n) size[v] = size[left[v - n]] + size[right[v - n]];
        }
    }
    out.clear();
    order.clear();
    max_depth = 0;
    struct Item {
        int node;
        uint32_t idx, depth;
    };
    std::vector<Item> work;
    out.push_back(BvhNode());
    work.push_back({root, 0u, 1u});
    std::vector<int> leaves;
    while (!work.empty()) {
        const Item it = work.back();
        work.pop_back();
        max_depth = std::max(max_depth, it.depth);
        const int ch[2] = {left[it.node - n], right[it.node - n]};
        BvhNode nd;
        std::memset(&nd, 0, sizeof(nd));
        Item pending[2];
        int npending = 0;
        for (int c = 0; c < 2; ++c) {
            const int v = ch[c];
            uint32_t ref;
            if (size[v] <= BVH_LEAF_MAX) {
                const uint32_t first = (uint32_t)order.size();
                leaves.assign(1, v);
                while (!leaves.empty()) {
                    const int u = leaves.back();
                    leaves.pop_back();
                    if (u < n) order.push_back(u);
                    else {
                        leaves.push_back(right[u - n]);
                        leaves.push_back(left[u - n]);
                    }
                }
                ref = MAKE_LEAF(kind, first, (uint32_t)size[v]);
            } else {
                const uint32_t idx = (uint32_t)out.size();
                out.push_back(BvhNode());
                pending[npending++] = {v, idx, it.depth + 1u};
                ref = node_base + idx;
            }
            float* lo = c == 0 ? nd.lo0 : nd.lo1;
            float* hi = c == 0 ? nd.hi0 : nd.hi1;
            for (int k = 0; k < 3; ++k) {
                lo[k] = box[v].lo[k];
                hi[k] = box[v].hi[k];
            }
            nd.child[c] = ref;
        }
        out[it.idx] = nd;
        for (int k = npending; k-- > 0;) work.push_back(pending[k]);
    }
}

// PLOC over the n Morton-sorted primitives; the tree stays on the device: d_left / d_right [n-1], node_box [2n-1], d_root [1].
std::string plocBuild(hipStream_t s, const Box6* d_boxes, const uint32_t* d_sorted, int n, float pad, int* d_left, int* d_right, Box6* node_box_out, int* d_root) {
#ifndef PLOC_RADIUS
#define PLOC_RADIUS 50
#endif
    const int radius = PLOC_RADIUS;
    struct {
        Box6* p;
    } node_box{node_box_out};
    struct {
        int* p;
    } d_left_{d_left}, d_right_{d_right};
    DevBuf<int> cl_a, cl_b, cl_out, nn, valid, pos, node_count;
    HIPCHK(cl_a.alloc(n)); HIPCHK(cl_b.alloc(n)); HIPCHK(cl_out.alloc(n));
    HIPCHK(nn.alloc(n)); HIPCHK(valid.alloc(n)); HIPCHK(pos.alloc(n)); HIPCHK(node_count.alloc(1));
    HIPCHK(hipMemsetAsync(node_count.p, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_ploc_init, dim3(nblocks(n)), dim3(BB), 0, s, d_boxes, d_sorted, n, pad, node_box.p, cl_a.p);
    size_t temp_bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, temp_bytes, valid.p, pos.p, 0, (size_t)n, rocprim::plus<int>(), s));
    DevBuf<char> temp;
    HIPCHK(temp.alloc(temp_bytes));
    int m = n;
    int* cur = cl_a.p;
    int* nxt = cl_b.p;
    for (int iter = 0; m > 1; ++iter) {
        if (iter > 4 * n + 64) return "BVH build: clustering does not converge";
        hipLaunchKernelGGL(k_ploc_nn, dim3(nblocks(m)), dim3(BB), 0, s, cur, m, node_box.p, radius, nn.p);
        hipLaunchKernelGGL(k_ploc_merge, dim3(nblocks(m)), dim3(BB), 0, s, cur, nn.p, m, n, node_box.p, d_left_.p, d_right_.p, node_count.p, cl_out.p, valid.p);
        size_t tb = temp_bytes;
        HIPCHK(rocprim::exclusive_scan(temp.p, tb, valid.p, pos.p, 0, (size_t)m, rocprim::plus<int>(), s));
        hipLaunchKernelGGL(k_ploc_scatter, dim3(nblocks(m)), dim3(BB), 0, s, cl_out.p, valid.p, pos.p, m, nxt);
        int tail[2];
        HIPCHK(hipMemcpyAsync(&tail[0], pos.p + (m - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&tail[1], valid.p + (m - 1), sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const int m2 = tail[0] + tail[1];
        if (m2 >= m || m2 < 1) return "BVH build: clustering made no progress";
        m = m2;
        std::swap(cur, nxt);
    }
    int h_count = 0;
    HIPCHK(hipMemcpyAsync(&h_count, node_count.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(d_root, cur, sizeof(int), hipMemcpyDeviceToDevice, s));  // the one cluster left
    HIPCHK(hipStreamSynchronize(s));
    if (h_count != n - 1) return "BVH build: clustering produced a wrong node count";
    return "";
}

// Top-down surface-area-heuristic build on the host (builder 2): every node is split where the sweep over
// the primitives sorted by centroid, along the best of the three axes, minimises
// area(left) * count(left) + area(right) * count(right).  Same output format as plocBuild: leaves 0..n-1
// (positions in `prims`), internal nodes n..2n-2, boxes padded.  O(n log^2 n), single-threaded: the
// better-quality / slower-build end of the builder choice.
void sahBuild(const std::vector<Box6>& boxes, const std::vector<uint32_t>& prims, float pad, std::vector<int>& left, std::vector<int>& right,
              std::vector<Box6>& box, int& root) {
    const int n = (int)prims.size();
    left.assign(n - 1, -1);
    right.assign(n - 1, -1);
    box.assign(2 * (size_t)n - 1, emptyBox());
    std::vector<float> cen[3];
    for (int k = 0; k < 3; ++k) cen[k].resize(n);
    for (int i = 0; i < n; ++i) {
        Box6 b = boxes[prims[i]];
        for (int k = 0; k < 3; ++k) {
            cen[k][i] = 0.5f * (b.lo[k] + b.hi[k]);
            b.lo[k] -= pad;
            b.hi[k] += pad;
        }
        box[i] = b;
    }
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    struct Job {
        int first, count, node;  // idx[first, first+count) -> internal node `node` (n-based)
    };
    std::atomic<int> next_internal(n);
    root = next_internal++;
    auto area = [](const Box6& b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    // One split: a job only touches its own range of idx and its own node, so jobs run on any thread; the node
    // labels depend on the schedule, the tree does not.
    auto split = [&](const Job& j, std::vector<int>& tmp, std::vector<float>& suffix, Job out[2]) -> int {
        int* a = idx.data() + j.first;
        Box6 nb = emptyBox();
        for (int i = 0; i < j.count; ++i) grow(nb, box[a[i]]);
        box[j.node] = nb;
        int best_axis = -1, best_split = j.count / 2;
        float best_cost = INFINITY;
        if ((int)tmp.size() < j.count) tmp.resize(j.count);
        if ((int)suffix.size() < j.count + 1) suffix.resize(j.count + 1);
        for (int k = 0; k < 3 && j.count > 2; ++k) {
            std::copy(a, a + j.count, tmp.begin());
            std::sort(tmp.begin(), tmp.begin() + j.count, [&](int x, int y) { return cen[k][x] < cen[k][y] || (cen[k][x] == cen[k][y] && x < y); });
            Box6 acc = emptyBox();
            suffix[j.count] = 0.f;
            for (int i = j.count - 1; i > 0; --i) {
                grow(acc, box[tmp[i]]);
                suffix[i] = area(acc);
            }
            acc = emptyBox();
            for (int i = 1; i < j.count; ++i) {
                grow(acc, box[tmp[i - 1]]);
                const float cost = area(acc) * (float)i + suffix[i] * (float)(j.count - i);
                // equal costs (coincident primitives: every split costs the same): the most even split, or the
                // subtree becomes a chain
                if (cost < best_cost || (cost == best_cost && std::abs(2 * i - j.count) < std::abs(2 * best_split - j.count))) {
                    best_cost = cost;
                    best_axis = k;
                    best_split = i;
                }
            }
        }
        if (best_axis >= 0)
            std::sort(a, a + j.count, [&](int x, int y) { return cen[best_axis][x] < cen[best_axis][y] || (cen[best_axis][x] == cen[best_axis][y] && x < y); });
        const int nl = best_split, nr = j.count - best_split;
        int child[2], n_out = 0;
        const int firsts[2] = {j.first, j.first + nl}, counts[2] = {nl, nr};
        for (int c = 0; c < 2; ++c) {
            if (counts[c] == 1) {
                child[c] = idx[firsts[c]];
            } else {
                child[c] = next_internal++;
                out[n_out++] = Job{firsts[c], counts[c], child[c]};
            }
        }
        left[j.node - n] = child[0];
        right[j.node - n] = child[1];
        return n_out;
    };
    // Large jobs go through a shared list (so that the two halves of a split run on different threads), small ones
    // are finished by the thread that made them.
    const int SHARE_MIN = 4096;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Job> shared{{0, n, root}};
    int pending = 1;  // jobs in `shared` or being split by a thread (under `mu`)
    std::exception_ptr failure;  // the first exception of any thread (under `mu`); rethrown by the caller's thread
    auto work = [&]() {
        std::vector<int> tmp;
        std::vector<float> suffix;
        std::vector<Job> local;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return failure || !shared.empty() || pending == 0; });
                if (failure || shared.empty()) return;
                j = shared.back();
                shared.pop_back();
            }
            local.assign(1, j);
            while (!local.empty()) {
                const Job cur = local.back();
                local.pop_back();
                Job out[2];
                const int n_out = split(cur, tmp, suffix, out);
                for (int c = 0; c < n_out; ++c) {
                    if (out[c].count >= SHARE_MIN) {
                        std::lock_guard<std::mutex> lock(mu);
                        shared.push_back(out[c]);
                        ++pending;
                        cv.notify_one();
                    } else {
                        local.push_back(out[c]);
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                if (--pending == 0) cv.notify_all();
            }
        }
    };
    // an exception in one thread (out of memory) ends the build for all of them instead of ending the process
    auto worker = [&]() {
        try {
            work();
        } catch (...) {
            std::lock_guard<std::mutex> lock(mu);
            if (!failure) failure = std::current_exception();
            cv.notify_all();
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_threads = n < 2 * SHARE_MIN ? 1 : (int)std::min<unsigned>(8u, std::max<unsigned>(1u, hw));
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    } catch (const std::system_error&) {
        // no more threads to be had: the ones that did start and this one do the work
    }
    worker();
    for (std::thread& t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
}

#ifndef BVH_ROTATE_PASSES
#define BVH_ROTATE_PASSES 4   /* dragon, sweep hierarchy: 6.77 -> 6.72 (1 pass) -> 6.53 (4) -> 6.52 (16) node visits per extension ray; the clustering's is a fixed point */
#endif
// Tree rotations on a finished binary tree (leaves 0..n-1, internal nodes n..2n-2): bottom-up, every internal node
// may swap one child with a grandchild on the other side when that shrinks the box of the child that is rebuilt
// (the node's own box does not change); `passes` sweeps.  A cheap local repair of what a clustering or a sweep
// decided early.
void rotateTree(std::vector<int>& left, std::vector<int>& right, std::vector<Box6>& box, int n, int root, int passes) {
    if (n < 3) return;
    auto area = [](const Box6& b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto unite = [](const Box6& a, const Box6& b) {
        Box6 u = a;
        grow(u, b);
        return u;
    };
    // post-order of the internal nodes
    std::vector<int> order;
    order.reserve(n - 1);
    {
        std::vector<int> stack{root};
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            if (v < n) continue;
            order.push_back(v);
            stack.push_back(left[v - n]);
            stack.push_back(right[v - n]);
        }
        std::reverse(order.begin(), order.end());  // children before parents
    }
    for (int pass = 0; pass < passes; ++pass) {
        int applied = 0;
        for (int v : order) {
            int& L = left[v - n];
            int& R = right[v - n];
            // candidates: (which child is rebuilt, which of its children goes up)
            float best = 0.f;
            int pick = -1;
            if (L >= n) {
                const int A = left[L - n], B = right[L - n];
                const float old = area(box[L]);
                const float a0 = area(unite(box[R], box[B])), a1 = area(unite(box[A], box[R]));  // swap R<->A, R<->B
                if (old - a0 > best) { best = old - a0; pick = 0; }
                if (old - a1 > best) { best = old - a1; pick = 1; }
            }
            if (R >= n) {
                const int C = left[R - n], D = right[R - n];
                const float old = area(box[R]);
                const float a2 = area(unite(box[L], box[D])), a3 = area(unite(box[C], box[L]));  // swap L<->C, L<->D
                if (old - a2 > best) { best = old - a2; pick = 2; }
                if (old - a3 > best) { best = old - a3; pick = 3; }
            }
            if (pick < 0) continue;
            ++applied;
            if (pick == 0) {  // L = (R, B), v = (L, A)
                const int A = left[L - n];
                left[L - n] = R;
                R = A;
                box[L] = unite(box[left[L - n]], box[right[L - n]]);
            } else if (pick == 1) {  // L = (A, R), v = (L, B)
                const int B = right[L - n];
                right[L - n] = R;
                R = B;
                box[L] = unite(box[left[L - n]], box[right[L - n]]);
            } else if (pick == 2) {  // R = (L, D), v = (C, R)
                const int C = left[R - n];
                left[R - n] = L;
                L = C;
                box[R] = unite(box[left[R - n]], box[right[R - n]]);
            } else {  // R = (C, L), v = (D, R)
                const int D = right[R - n];
                right[R - n] = L;
                L = D;
                box[R] = unite(box[left[R - n]], box[right[R - n]]);
            }
        }
        if (!applied) break;
        // the post-order of the rotated tree for the next sweep
        order.clear();
        std::vector<int> stack{root};
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            if (v < n) continue;
            order.push_back(v);
            stack.push_back(left[v - n]);
            stack.push_back(right[v - n]);
        }
        std::reverse(order.begin(), order.end());
    }
}

// Tree over the small primitives of one kind (builder 0: LBVH, 1: PLOC, 2: host SAH sweep); writes the emitted nodes to nodes[node_base..].
std::string buildKind(hipStream_t s, const Box6* d_boxes, const std::vector<Box6>& boxes, uint32_t kind, float pad, int builder, BvhNode* d_nodes,
                      uint32_t node_base, KindResult& res) {
    std::vector<uint32_t> small, big;
#ifndef SMALL_SCENE_PRIMS
#define SMALL_SCENE_PRIMS 1024u
#endif
    if (boxes.size() <= SMALL_SCENE_PRIMS && boxes.size() > BVH_LEAF_MAX) {
        // A few primitives (a Cornell box: every wall is as large as the scene): no "large primitive" leaf that
        // every ray would have to test, and the surface-area sweep on the host, which costs nothing at this size
        // and copes with mixed sizes.
        for (uint32_t i = 0; i < boxes.size(); ++i) small.push_back(i);
        builder = 2;
    } else {
        classify(boxes, small, big);
    }
    res.n_small = (uint32_t)small.size();
    res.n_big = (uint32_t)big.size();
    res.small_box = emptyBox();
    res.big_box = emptyBox();
    auto padded = [&](Box6 b) {
        for (int k = 0; k < 3; ++k) {
            b.lo[k] -= pad;
            b.hi[k] += pad;
        }
        return b;
    };
    for (uint32_t i : small) grow(res.small_box, padded(boxes[i]));
    for (uint32_t i : big) grow(res.big_box, padded(boxes[i]));
    const int n = (int)small.size();
    res.slot_prim.clear();
    if (n > BVH_LEAF_MAX) {
        Box6 cb = emptyBox();
        for (uint32_t i : small) {
            Box6 c;
            for (int k = 0; k < 3; ++k) c.lo[k] = c.hi[k] = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]);
            grow(cb, c);
        }
        DevBuf<uint32_t> ids;
        HIPCHK(ids.alloc(n));
        HIPCHK(hipMemcpyAsync(ids.p, small.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        if (builder == 4) {
            // Surface-area sweep, rotations, leaf collapse and node emission on the device; the host sees the
            // primitive order of the finished tree and four words about it.
            DeviceArena arena;
            sweep_reserve(arena, n);
            tree_reserve(arena, n);
            arena.reserve<int32_t>((size_t)n);
            HIPCHK(arena.commit());
            DeviceTree tree;
            std::string err = sweep_build(s, arena, d_boxes, ids.p, n, pad, tree);
            if (!err.empty()) return err;
            err = tree_rotate(s, arena, tree, BVH_ROTATE_PASSES);
            if (!err.empty()) return err;
            int32_t* d_slot = arena.take<int32_t>((size_t)n);
            if (!d_slot) return "BVH build: work space exhausted";
            err = tree_emit(s, arena, tree, kind, node_base, d_nodes, d_slot, res.n_tree_nodes, res.max_depth, res.tree_box);
            if (!err.empty()) return err;
            res.slot_prim.resize(n);
            HIPCHK(hipMemcpyAsync(res.slot_prim.data(), d_slot, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            res.has_tree = true;
            for (uint32_t i : big) res.slot_prim.push_back((int32_t)i);
            return "";
        }
        if (builder == 2) {
            std::vector<int> h_left, h_right, order;
            std::vector<Box6> h_box;
            std::vector<BvhNode> h_nodes;
            int root = 0;
            uint32_t depth = 0;
            {
                HostPass hp(res.host_ms);
                sahBuild(boxes, small, pad, h_left, h_right, h_box, root);
                rotateTree(h_left, h_right, h_box, n, root, BVH_ROTATE_PASSES);
                emitTree(h_left, h_right, h_box, n, root, kind, node_base, h_nodes, order, depth);
            }
            HIPCHK(hipMemcpyAsync(d_nodes + node_base, h_nodes.data(), sizeof(BvhNode) * h_nodes.size(), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            res.max_depth = depth;
            res.n_tree_nodes = (uint32_t)h_nodes.size();
            res.has_tree = true;
            res.tree_box = h_box[root];
            for (int i = 0; i < n; ++i) res.slot_prim.push_back((int32_t)small[order[i]]);
            for (uint32_t i : big) res.slot_prim.push_back((int32_t)i);
            return "";
        }
        float3 cmin = make_float3(cb.lo[0], cb.lo[1], cb.lo[2]);
        float3 cinv;
        cinv.x = cb.hi[0] > cb.lo[0] ? 1.0f / (cb.hi[0] - cb.lo[0]) : 0.f;
        cinv.y = cb.hi[1] > cb.lo[1] ? 1.0f / (cb.hi[1] - cb.lo[1]) : 0.f;
        cinv.z = cb.hi[2] > cb.lo[2] ? 1.0f / (cb.hi[2] - cb.lo[2]) : 0.f;

        DevBuf<uint32_t> vals_in, vals_out, left, right;
        DevBuf<uint64_t> keys_in, keys_out;
        DevBuf<int> first, last, parent, done, newidx;
        DevBuf<Box6> node_box;
        DevBuf<uint32_t> depth;
        HIPCHK(vals_in.alloc(n)); HIPCHK(vals_out.alloc(n));
        HIPCHK(keys_in.alloc(n)); HIPCHK(keys_out.alloc(n));
        HIPCHK(left.alloc(n)); HIPCHK(right.alloc(n)); HIPCHK(first.alloc(n)); HIPCHK(last.alloc(n));
        HIPCHK(parent.alloc(2 * (size_t)n)); HIPCHK(done.alloc(n)); HIPCHK(newidx.alloc(n));
        HIPCHK(node_box.alloc(2 * (size_t)n)); HIPCHK(depth.alloc(1));
        hipLaunchKernelGGL(k_morton, dim3(nblocks(n)), dim3(BB), 0, s, d_boxes, ids.p, (uint32_t)n, cmin, cinv, keys_in.p, vals_in.p);
        size_t temp_bytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in.p, keys_out.p, vals_in.p, vals_out.p, (size_t)n, 0, 63, s));
        DevBuf<char> temp;
        HIPCHK(temp.alloc(temp_bytes));
        HIPCHK(rocprim::radix_sort_pairs(temp.p, temp_bytes, keys_in.p, keys_out.p, vals_in.p, vals_out.p, (size_t)n, 0, 63, s));
        if (builder == 1) {
            // clustering, rotations, leaf collapse and node emission on the device
            DeviceArena arena;
            tree_reserve(arena, n);
            arena.reserve<int>((size_t)n);
            arena.reserve<int>((size_t)n);
            arena.reserve<Box6>(2 * (size_t)n);
            arena.reserve<int>(8);
            arena.reserve<int32_t>((size_t)n);
            HIPCHK(arena.commit());
            DeviceTree tree;
            tree.n = n;
            tree.left = arena.take<int>((size_t)n);
            tree.right = arena.take<int>((size_t)n);
            tree.box = arena.take<Box6>(2 * (size_t)n);
            tree.root = arena.take<int>(8);
            tree.leaf_prim = vals_out.p;
            int32_t* d_slot = arena.take<int32_t>((size_t)n);
            if (!tree.left || !tree.right || !tree.box || !tree.root || !d_slot) return "BVH build: work space exhausted";
            std::string err = plocBuild(s, d_boxes, vals_out.p, n, pad, tree.left, tree.right, tree.box, tree.root);
            if (!err.empty()) return err;
            err = tree_rotate(s, arena, tree, BVH_ROTATE_PASSES);
            if (!err.empty()) return err;
            err = tree_emit(s, arena, tree, kind, node_base, d_nodes, d_slot, res.n_tree_nodes, res.max_depth, res.tree_box);
            if (!err.empty()) return err;
            res.slot_prim.resize(n);
            HIPCHK(hipMemcpyAsync(res.slot_prim.data(), d_slot, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            res.has_tree = true;
            for (uint32_t i : big) res.slot_prim.push_back((int32_t)i);
            return "";
        }
        hipLaunchKernelGGL(k_karras, dim3(nblocks(n - 1)), dim3(BB), 0, s, keys_out.p, n, left.p, right.p, first.p, last.p, parent.p);
        hipLaunchKernelGGL(k_leaf_boxes, dim3(nblocks(n)), dim3(BB), 0, s, d_boxes, vals_out.p, n, pad, node_box.p);
        HIPCHK(hipMemsetAsync(depth.p, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_depth, dim3(nblocks(n)), dim3(BB), 0, s, parent.p, n, depth.p);
        uint32_t h_depth = 0;
        HIPCHK(hipMemcpyAsync(&h_depth, depth.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (h_depth >= 4096u) return "BVH build: malformed radix tree";
        res.max_depth = h_depth;
        HIPCHK(hipMemsetAsync(done.p, 0, sizeof(int) * n, s));
        for (int pass = 1; pass <= (int)h_depth + 1; ++pass)
            hipLaunchKernelGGL(k_refit_pass, dim3(nblocks(n - 1)), dim3(BB), 0, s, left.p, right.p, n, pass, done.p, node_box.p);
        // which internal nodes survive the leaf collapse, renumbered densely (root first)
        std::vector<int> h_first(n - 1), h_last(n - 1), h_done(n - 1), h_new(n - 1);
        std::vector<uint32_t> h_sorted(n);
        HIPCHK(hipMemcpyAsync(h_first.data(), first.p, sizeof(int) * (n - 1), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_last.data(), last.p, sizeof(int) * (n - 1), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_done.data(), done.p, sizeof(int) * (n - 1), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_sorted.data(), vals_out.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s));
        Box6 rootBox;
        HIPCHK(hipMemcpyAsync(&rootBox, node_box.p, sizeof(Box6), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        int emitted = 0;
        for (int i = 0; i < n - 1; ++i) {
            if (h_done[i] == 0) return "BVH build: box pass did not reach every node";
            h_new[i] = (h_last[i] - h_first[i] + 1 > BVH_LEAF_MAX) ? emitted++ : -1;
        }
        HIPCHK(hipMemcpyAsync(newidx.p, h_new.data(), sizeof(int) * (n - 1), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_emit, dim3(nblocks(n - 1)), dim3(BB), 0, s, left.p, right.p, first.p, last.p, newidx.p, node_box.p, n, node_base, kind, d_nodes);
        HIPCHK(hipStreamSynchronize(s));
        res.n_tree_nodes = (uint32_t)emitted;
        res.has_tree = true;
        res.tree_box = rootBox;
        for (int i = 0; i < n; ++i) res.slot_prim.push_back((int32_t)h_sorted[i]);
    } else {
        for (uint32_t i : small) res.slot_prim.push_back((int32_t)i);
    }
    for (uint32_t i : big) res.slot_prim.push_back((int32_t)i);
    return "";
}

struct TopEntry {
    uint32_t ref;
    Box6 box;
};

void setChild(BvhNode& nd, int c, const TopEntry& e) {
    float* lo = c == 0 ? nd.lo0 : nd.lo1;
    float* hi = c == 0 ? nd.hi0 : nd.hi1;
    for (int k = 0; k < 3; ++k) {
        lo[k] = e.box.lo[k];
        hi[k] = e.box.hi[k];
    }
    nd.child[c] = e.ref;
}

// ---- four-wide nodes ----------------------------------------------------------
// The finished binary hierarchy, collapsed: a wide node starts with the two children of a
// binary node and, while it has room, replaces the internal child with the largest box by
// that child's own two children.  Child boxes are then stored on a 256-step grid of the
// wide node's box, rounded outwards, so the wide hierarchy accepts every ray the binary
// one accepts.  Returns false when some box cannot be represented that way (non-finite
// coordinates) or the tree would need more stack than the traversal kernels have.
struct WideChild {
    uint32_t ref;
    Box6 box;
};

float boxArea(const Box6& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

bool collapseWide(const std::vector<BvhNode>& bin, std::vector<WideNode>& wide, uint32_t& depth_out, uint32_t unused_ref = REF_EMPTY_WORD) {
    wide.clear();
    depth_out = 0;
    if (bin.empty()) return false;
    struct Work {
        uint32_t bin, widx, depth;
    };
    std::vector<Work> work;
    wide.emplace_back();
    work.push_back({0u, 0u, 1u});
    auto childrenOf = [&](uint32_t b, std::vector<WideChild>& dst) {
        const BvhNode& nd = bin[b];
        for (int c = 0; c < 2; ++c) {
            if (REF_KIND(nd.child[c]) == REF_EMPTY) continue;
            WideChild w;
            w.ref = nd.child[c];
            const float* lo = c == 0 ? nd.lo0 : nd.lo1;
            const float* hi = c == 0 ? nd.hi0 : nd.hi1;
            for (int k = 0; k < 3; ++k) {
                w.box.lo[k] = lo[k];
                w.box.hi[k] = hi[k];
            }
            dst.push_back(w);
        }
    };
    std::vector<WideChild> ch, tmp;
    while (!work.empty()) {
        const Work it = work.back();
        work.pop_back();
        depth_out = std::max(depth_out, it.depth);
        ch.clear();
        childrenOf(it.bin, ch);
        for (;;) {
            int pick = -1;
            float best = -1.f;
            for (size_t k = 0; k < ch.size(); ++k) {
                if (REF_KIND(ch[k].ref) != REF_INTERNAL) continue;
                tmp.clear();
                childrenOf(REF_NODE(ch[k].ref), tmp);
                if (ch.size() - 1 + tmp.size() > 4) continue;
                const float a = boxArea(ch[k].box);
                if (a > best) {
                    best = a;
                    pick = (int)k;
                }
            }
            if (pick < 0) break;
            tmp.clear();
            childrenOf(REF_NODE(ch[pick].ref), tmp);
            ch.erase(ch.begin() + pick);
            ch.insert(ch.end(), tmp.begin(), tmp.end());
        }
        // Storage order = the order in which any-hit rays visit the children that they hit: smallest box first (an
        // occluder is found sooner in a small box than in a large one; measured against largest first, leaves first
        // and internal nodes first).  Closest-hit rays sort by entry distance and do not care.
        std::stable_sort(ch.begin(), ch.end(), [](const WideChild& a, const WideChild& b) { return boxArea(a.box) < boxArea(b.box); });
        WideNode out;
        std::memset(&out, 0, sizeof(out));
        // Unused child slots.  For the traversal kernels (`unused_ref` = a leaf of one degenerate triangle) their box is
        // inside out on every axis, lower plane 255 and upper plane 0: no ray enters it, so the kernels need not look
        // at the reference to know that a slot is unused -- and should rounding ever let a ray in, the reference
        // leads to a test that cannot succeed.
        for (int c = 0; c < 4; ++c) {
            out.child[c] = unused_ref;
            if (unused_ref != REF_EMPTY_WORD) {
                out.qlo_x[c] = out.qlo_y[c] = out.qlo_z[c] = 255;
                out.qhi_x[c] = out.qhi_y[c] = out.qhi_z[c] = 0;
            }
        }
        if (!ch.empty()) {
            Box6 u = ch[0].box;
            for (size_t k = 1; k < ch.size(); ++k) grow(u, ch[k].box);
            float scale[3];
            for (int k = 0; k < 3; ++k) {
                if (!std::isfinite(u.lo[k]) || !std::isfinite(u.hi[k])) return false;
                const double ext = (double)u.hi[k] - (double)u.lo[k];
                int e = -100;
                if (ext > 0.0) {
                    int ex;
                    const double m = std::frexp(ext / 255.0, &ex);  // ext/255 = m * 2^ex, m in [0.5, 1)
                    e = m == 0.5 ? ex - 1 : ex;
                }
                e = std::min(std::max(e, -100), 100);
                scale[k] = std::ldexp(1.0f, e);
                if ((double)scale[k] * 255.0 < ext) return false;
                out.origin[k] = u.lo[k];
            }
            out.scale_x = scale[0];
            out.scale_y = scale[1];
            out.scale_z = scale[2];
            for (size_t c = 0; c < ch.size(); ++c) {
                uint8_t* qlo[3] = {out.qlo_x, out.qlo_y, out.qlo_z};
                uint8_t* qhi[3] = {out.qhi_x, out.qhi_y, out.qhi_z};
                for (int k = 0; k < 3; ++k) {
                    const double o = (double)u.lo[k], sc = (double)scale[k];
                    if (!std::isfinite(ch[c].box.lo[k]) || !std::isfinite(ch[c].box.hi[k])) return false;
                    double lo = std::floor(((double)ch[c].box.lo[k] - o) / sc), hi = std::ceil(((double)ch[c].box.hi[k] - o) / sc);
                    lo = std::min(std::max(lo, 0.0), 255.0);
                    hi = std::min(std::max(hi, 0.0), 255.0);
                    if (o + lo * sc > (double)ch[c].box.lo[k] || o + hi * sc < (double)ch[c].box.hi[k]) return false;
                    qlo[k][c] = (uint8_t)lo;
                    qhi[k][c] = (uint8_t)hi;
                }
                if (REF_KIND(ch[c].ref) == REF_INTERNAL) {
                    const uint32_t widx = (uint32_t)wide.size();
                    wide.emplace_back();
                    work.push_back({REF_NODE(ch[c].ref), widx, it.depth + 1u});
                    out.child[c] = widx;
                } else {
                    out.child[c] = ch[c].ref;
                }
            }
        }
        wide[it.widx] = out;
    }
    // a ray's stack grows by at most three entries per level
    return 3u * depth_out + 1u <= TRAV_LDS_STACK + TRAV_SPILL_STACK;
}

}  // namespace

#define TOP_NODES 3u

std::string bvh_build(hipStream_t s, const BvhBuildInput& in, BvhBuildOutput& out) {
    const uint32_t T = in.n_tris, S = in.n_spheres;
    // 1. primitive boxes
    DevBuf<Box6> d_tbox, d_sbox;
    HIPCHK(d_tbox.alloc(T));
    HIPCHK(d_sbox.alloc(S));
    if (T) hipLaunchKernelGGL(k_bounds_tri, dim3(nblocks(T)), dim3(BB), 0, s, in.d_tris, T, d_tbox.p);
    if (S) hipLaunchKernelGGL(k_bounds_sphere, dim3(nblocks(S)), dim3(BB), 0, s, in.d_spheres, S, d_sbox.p);
    std::vector<Box6> tbox(T), sbox(S);
    if (T) HIPCHK(hipMemcpyAsync(tbox.data(), d_tbox.p, sizeof(Box6) * T, hipMemcpyDeviceToHost, s));
    if (S) HIPCHK(hipMemcpyAsync(sbox.data(), d_sbox.p, sizeof(Box6) * S, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));

    // 2. padding: a few dozen ulps of the largest coordinate a ray can start from or reach
    float ext = in.extra_extent;
    for (const auto* v : {&tbox, &sbox})
        for (const Box6& b : *v)
            for (int k = 0; k < 3; ++k) {
                if (std::isfinite(b.lo[k])) ext = std::max(ext, fabsf(b.lo[k]));
                if (std::isfinite(b.hi[k])) ext = std::max(ext, fabsf(b.hi[k]));
            }
    const float pad = 2e-6f * std::max(ext, 1.0f);

    // 3. trees (node array: TOP_NODES top-level nodes, then the triangle tree, then the sphere tree)
    const size_t max_nodes = TOP_NODES + (size_t)T + (size_t)S + 2;
    HIPCHK(hipMalloc((void**)&out.nodes, max_nodes * sizeof(BvhNode)));
    HIPCHK(hipMemsetAsync(out.nodes, 0, max_nodes * sizeof(BvhNode), s));
    KindResult kt, ks;
    std::string err = buildKind(s, d_tbox.p, tbox, REF_TRI, pad, in.builder, out.nodes, TOP_NODES, kt);
    if (!err.empty()) return err;
    err = buildKind(s, d_sbox.p, sbox, REF_SPHERE, pad, in.builder, out.nodes, TOP_NODES + kt.n_tree_nodes, ks);
    if (!err.empty()) return err;
    out.n_nodes = TOP_NODES + kt.n_tree_nodes + ks.n_tree_nodes;
    out.max_depth = std::max(kt.max_depth, ks.max_depth) + 2u;
    out.host_ms += kt.host_ms + ks.host_ms;
    {
        Box6 sb = emptyBox();
        for (const KindResult* k : {&kt, &ks}) {
            if (k->has_tree) grow(sb, k->tree_box);
            else if (k->n_small) grow(sb, k->small_box);
        }
        if (!(sb.lo[0] <= sb.hi[0])) {  // only large primitives: their box
            if (kt.n_big) grow(sb, kt.big_box);
            if (ks.n_big) grow(sb, ks.big_box);
        }
        for (int k = 0; k < 3; ++k) {
            out.sort_lo[k] = std::isfinite(sb.lo[k]) ? sb.lo[k] : 0.f;
            out.sort_hi[k] = std::isfinite(sb.hi[k]) ? sb.hi[k] : 1.f;
        }
    }
    if (out.max_depth > TRAV_LDS_STACK + TRAV_SPILL_STACK) return "BVH too deep for the traversal stack";

    // 4. top level
    std::vector<TopEntry> entries;
    auto addKind = [&](const KindResult& k, uint32_t kind, uint32_t base) {
        if (k.has_tree) entries.push_back({base, k.tree_box});
        else if (k.n_small) entries.push_back({MAKE_LEAF(kind, 0u, k.n_small), k.small_box});
        if (k.n_big) entries.push_back({MAKE_LEAF(kind, k.n_small, k.n_big), k.big_box});
    };
    addKind(kt, REF_TRI, TOP_NODES);
    addKind(ks, REF_SPHERE, TOP_NODES + kt.n_tree_nodes);
    BvhNode top[TOP_NODES];
    std::memset(top, 0, sizeof(top));
    for (auto& nd : top) nd.child[0] = nd.child[1] = REF_EMPTY_WORD;
    auto unionBox = [](const TopEntry& a, const TopEntry& b) {
        Box6 u = a.box;
        grow(u, b.box);
        return u;
    };
    if (entries.empty()) {
        out.n_nodes = 0;
    } else if (entries.size() == 1) {
        setChild(top[0], 0, entries[0]);
    } else if (entries.size() == 2) {
        setChild(top[0], 0, entries[0]);
        setChild(top[0], 1, entries[1]);
        // triangles in one entry, all (one or two) spheres in the other: the spheres can be tested up front
        if (S >= 1 && S <= 2 && kt.n_big == 0 && REF_KIND(entries[1].ref) == REF_SPHERE && REF_COUNT(entries[1].ref) == S) out.root_without_spheres = entries[0].ref;
        // a triangle tree and one leaf with the (at most three) large triangles: those can be plane-tested for every ray up front
        if (S == 0 && kt.has_tree && kt.n_big >= 1 && kt.n_big <= 3 && entries[1].ref == MAKE_LEAF(REF_TRI, kt.n_small, kt.n_big)) {
            out.big_first = kt.n_small;
            out.big_count = kt.n_big;
            for (uint32_t k = 0; k < kt.n_big; ++k) out.big_prim[k] = kt.slot_prim[kt.n_small + k];
            out.root_without_big = entries[0].ref;
        }
    } else if (entries.size() == 3) {
        setChild(top[1], 0, entries[1]);
        setChild(top[1], 1, entries[2]);
        setChild(top[0], 0, entries[0]);
        setChild(top[0], 1, TopEntry{1u, unionBox(entries[1], entries[2])});
    } else {
        setChild(top[1], 0, entries[0]);
        setChild(top[1], 1, entries[1]);
        setChild(top[2], 0, entries[2]);
        setChild(top[2], 1, entries[3]);
        setChild(top[0], 0, TopEntry{1u, unionBox(entries[0], entries[1])});
        setChild(top[0], 1, TopEntry{2u, unionBox(entries[2], entries[3])});
    }
    HIPCHK(hipMemcpyAsync(out.nodes, top, sizeof(top), hipMemcpyHostToDevice, s));

    // 5. geometry in leaf order
    // one triangle more than the scene has: all zeros (no ray hits it), what the unused child slots of the wide nodes refer to
    HIPCHK(hipMalloc((void**)&out.tri_q, ((size_t)T + 1) * 3 * sizeof(float4)));
    HIPCHK(hipMalloc((void**)&out.tri_mat, ((size_t)T + 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void**)&out.tri_nm, ((size_t)T + 1) * sizeof(float4)));
    HIPCHK(hipMalloc((void**)&out.tri_prim, ((size_t)T + 1) * sizeof(int32_t)));
    HIPCHK(hipMemsetAsync(out.tri_q + 3 * (size_t)T, 0, 3 * sizeof(float4), s));
    HIPCHK(hipMemsetAsync(out.tri_mat + T, 0, sizeof(uint32_t), s));
    HIPCHK(hipMemsetAsync(out.tri_nm + T, 0, sizeof(float4), s));
    HIPCHK(hipMemsetAsync(out.tri_prim + T, 0, sizeof(int32_t), s));
    HIPCHK(hipMalloc((void**)&out.spheres, std::max<size_t>(S, 1) * sizeof(DevSphere)));
    HIPCHK(hipMalloc((void**)&out.sph_mat, std::max<size_t>(S, 1) * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void**)&out.sph_prim, std::max<size_t>(S, 1) * sizeof(int32_t)));
    std::vector<uint32_t> tmat(T), smat(S);
    std::vector<int32_t> sprim(S);
    for (uint32_t i = 0; i < T; ++i) tmat[i] = in.h_tri_mat[kt.slot_prim[i]];
    if (T) {
        HIPCHK(hipMemcpyAsync(out.tri_prim, kt.slot_prim.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(out.tri_mat, tmat.data(), sizeof(uint32_t) * T, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pack_tris, dim3(nblocks(T)), dim3(BB), 0, s, in.d_tris, out.tri_prim, out.tri_mat, T, out.tri_q, out.tri_nm);
    }
    std::vector<DevSphere> hs(S);
    if (S) {
        std::vector<PtcSphere> hsph(S);
        HIPCHK(hipMemcpyAsync(hsph.data(), in.d_spheres, sizeof(PtcSphere) * S, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < S; ++i) {
            const int32_t src = ks.slot_prim[i];
            mat4 m;
            std::memcpy(m.m, hsph[src].trans, sizeof(m.m));
            const mat4 inv = mat4_inverse(m);
            std::memcpy(hs[i].trans, m.m, sizeof(m.m));
            std::memcpy(hs[i].itrans, inv.m, sizeof(inv.m));
            smat[i] = in.h_sph_mat[src];
            sprim[i] = (int32_t)T + src;  // global primitive index: triangles first
        }
        HIPCHK(hipMemcpyAsync(out.spheres, hs.data(), sizeof(DevSphere) * S, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(out.sph_mat, smat.data(), sizeof(uint32_t) * S, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(out.sph_prim, sprim.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipStreamSynchronize(s));

    // 6. the four-wide version of the hierarchy
#ifndef WIDE_HOST_MAX_NODES
#define WIDE_HOST_MAX_NODES 256u   /* hierarchies of a few nodes: the host loop is quicker than two launches per level */
#endif
    if (out.n_nodes > WIDE_HOST_MAX_NODES) {
        HIPCHK(hipMalloc((void**)&out.wnodes, ((size_t)out.n_nodes + 1u) * sizeof(WideNode)));  // + the root without the large triangles, below
        bool wide_ok = false;
        const std::string werr = wide_collapse_device(s, out.nodes, out.n_nodes, out.max_depth, MAKE_LEAF(REF_TRI, T, 1u), out.wnodes, out.n_wnodes, out.wide_depth, wide_ok);
        if (!werr.empty()) return werr;
        if (!wide_ok) {  // not representable: the kernels fall back to the two-box nodes
            (void)hipFree(out.wnodes);
            out.wnodes = nullptr;
            out.n_wnodes = out.wide_depth = 0;
        }
    } else if (out.n_nodes) {
        std::vector<BvhNode> h_bin(out.n_nodes);
        HIPCHK(hipMemcpyAsync(h_bin.data(), out.nodes, sizeof(BvhNode) * out.n_nodes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<WideNode> h_wide;
        uint32_t wdepth = 0;
        bool wide_ok;
        {
            HostPass hp(out.host_ms);
            wide_ok = collapseWide(h_bin, h_wide, wdepth, MAKE_LEAF(REF_TRI, T, 1u));
        }
        if (wide_ok) {  // slot T: the degenerate triangle after the scene's
            HIPCHK(hipMalloc((void**)&out.wnodes, (h_wide.size() + 1u) * sizeof(WideNode)));
            HIPCHK(hipMemcpyAsync(out.wnodes, h_wide.data(), h_wide.size() * sizeof(WideNode), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            out.n_wnodes = (uint32_t)h_wide.size();
            out.wide_depth = wdepth;
        }
    }
    // 7. the four-wide root once more, after the hierarchy's own nodes, with the child slot of the large-triangle leaf emptied:
    // where the kernels that test those triangles up front start
    if (out.wnodes && out.big_count) {
        WideNode root;
        HIPCHK(hipMemcpyAsync(&root, out.wnodes, sizeof(root), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const uint32_t leaf = MAKE_LEAF(REF_TRI, out.big_first, out.big_count);
        for (int c = 0; c < 4; ++c)
            if (root.child[c] == leaf) {  // a leaf child of the binary root is a child of the wide root (the collapse only opens internal children)
                root.child[c] = MAKE_LEAF(REF_TRI, T, 1u);  // like every unused slot: an inside-out box, the all-zero triangle after the scene's
                root.qlo_x[c] = root.qlo_y[c] = root.qlo_z[c] = 255;
                root.qhi_x[c] = root.qhi_y[c] = root.qhi_z[c] = 0;
                HIPCHK(hipMemcpyAsync(out.wnodes + out.n_wnodes, &root, sizeof(root), hipMemcpyHostToDevice, s));
                HIPCHK(hipStreamSynchronize(s));
                out.wide_root_without_big = out.n_wnodes;
                break;
            }
    }
    return "";
}

int bvh_collapse_wide_host(const BvhNode* bin, uint32_t n_nodes, WideNode* out, uint32_t capacity, uint32_t* n_wide, uint32_t* depth) {
    // caller-supplied nodes: every internal child must exist and be reached once (a tree, no cycles)
    {
        std::vector<uint8_t> seen(n_nodes, 0);
        std::vector<uint32_t> todo(1, 0u);
        seen[0] = 1;
        while (!todo.empty()) {
            const uint32_t b = todo.back();
            todo.pop_back();
            for (int c = 0; c < 2; ++c) {
                const uint32_t ref = bin[b].child[c];
                if (REF_KIND(ref) != REF_INTERNAL) continue;
                const uint32_t k = REF_NODE(ref);
                if (k >= n_nodes || seen[k]) return 3;
                seen[k] = 1;
                todo.push_back(k);
            }
        }
    }
    std::vector<BvhNode> b(bin, bin + n_nodes);
    std::vector<WideNode> w;
    uint32_t d = 0;
    if (!collapseWide(b, w, d)) return 1;
    if (w.size() > capacity) return 2;
    std::memcpy(out, w.data(), w.size() * sizeof(WideNode));
    *n_wide = (uint32_t)w.size();
    *depth = d;
    return 0;
}

void bvh_free(BvhBuildOutput& o) {
    for (void* p : {(void*)o.tri_q, (void*)o.tri_nm, (void*)o.tri_mat, (void*)o.tri_prim, (void*)o.spheres, (void*)o.sph_mat, (void*)o.sph_prim, (void*)o.nodes, (void*)o.wnodes})
        if (p) (void)hipFree(p);
    o = BvhBuildOutput();
}
