// bvh_device.h -- the parts of the hierarchy build that work on a binary tree held in device memory
// (bvh_device.hip): surface-area sweep build, tree rotations, emission of the traversal nodes and
// the collapse into four-wide nodes.  Shared with bvh_build.hip, which owns the rest of the build
// (primitive boxes, Morton codes, clustering, top level, geometry repacking).
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <string>

#include "device_scene.h"

struct Box6 {
    float lo[3], hi[3];
};

// One device allocation handed out in 256-byte-aligned slices (a build makes dozens of work arrays; one
// hipMalloc / hipFree pair instead of one per array).  Sizes are added up with reserve() first.
class DeviceArena {
public:
    ~DeviceArena() {
        if (base_) (void)hipFree(base_);
    }
    template <class T>
    void reserve(size_t n) { need_ += slice(n * sizeof(T)); }
    hipError_t commit() {
        if (base_) return hipErrorInvalidValue;
        return hipMalloc((void**)&base_, need_ ? need_ : 256u);
    }
    // nullptr when the reservations did not cover the request
    template <class T>
    T* take(size_t n) {
        const size_t bytes = slice(n * sizeof(T));
        if (!base_ || used_ + bytes > need_) return nullptr;
        T* p = reinterpret_cast<T*>(base_ + used_);
        used_ += bytes;
        return p;
    }

private:
    static size_t slice(size_t bytes) { return ((bytes ? bytes : 1u) + 255u) & ~(size_t)255u; }
    char* base_ = nullptr;
    size_t need_ = 0, used_ = 0;
};

// A binary tree over n leaves in device memory: leaves are nodes 0..n-1, internal nodes n..2n-2;
// left[i] / right[i] are the children of internal node n+i; box[v] is the (padded) box of node v;
// *root is the root's node number; leaf_prim[i] is the primitive (index within its kind) of leaf i.
struct DeviceTree {
    int n = 0;
    int* left = nullptr;
    int* right = nullptr;
    Box6* box = nullptr;
    int* root = nullptr;
    const uint32_t* leaf_prim = nullptr;
};

// Work space of the functions below for a tree of n leaves.
void tree_reserve(DeviceArena& a, int n);
void sweep_reserve(DeviceArena& a, int n);

// Top-down surface-area sweep: every node is split where the sweep over its primitives sorted by centroid, along the
// best of the three axes, minimises area(left) * count(left) + area(right) * count(right); equal costs go to the most
// even split, then to the lower axis, then to the lower position.  Level by level, all nodes of a level at once: three
// arrays keep the leaves of every node contiguous and sorted by centroid along x / y / z, prefix and suffix boxes come
// from segmented scans, the best split from a segmented minimum, the three orders are split in place by a stable
// partition.  `boxes` are the primitive boxes of the kind, `ids` the n primitives to build over (leaf i = ids[i]).
// Fills `t` (arrays taken from `a`).  Returns "" or an error; an error containing "too deep" means the level limit
// was reached.
std::string sweep_build(hipStream_t s, DeviceArena& a, const Box6* boxes, const uint32_t* ids, int n, float pad, DeviceTree& t);

// Bottom-up tree rotations, `passes` sweeps (fewer when a sweep changes nothing): a node swaps a child with a grandchild
// on the other side when that shrinks the box of the child that is rebuilt.  Level by level from the deepest, all nodes
// of a level at once.
std::string tree_rotate(hipStream_t s, DeviceArena& a, DeviceTree& t, int passes);

// Traversal nodes of the tree: subtrees of at most BVH_LEAF_MAX primitives become leaves, the other internal nodes are
// written to nodes[node_base ...] in pre-order; slot_prim[k] receives the primitive of slot k (slots go left to right
// through the leaves).  n_nodes = nodes written, max_depth = levels of written nodes, root_box = box of the root.
std::string tree_emit(hipStream_t s, DeviceArena& a, const DeviceTree& t, uint32_t kind, uint32_t node_base, BvhNode* nodes, int32_t* slot_prim,
                      uint32_t& n_nodes, uint32_t& max_depth, Box6& root_box);

// Four-wide nodes from the finished two-box nodes bin[0 .. n_bin), level by level.  `wide` must hold n_bin nodes.
// ok = false: some box cannot be represented on the 8-bit grid, or the wide tree needs more stack than the traversal
// kernels have.  The same rules as the host-side collapse (a wide node starts with a binary node's two children and,
// while it has room, replaces the internal child with the largest box by that child's children; children stored smallest
// box first; planes rounded outwards and checked in double precision), only the numbering of the nodes differs.
std::string wide_collapse_device(hipStream_t s, const BvhNode* bin, uint32_t n_bin, uint32_t bin_depth, uint32_t unused_ref, WideNode* wide,
                                 uint32_t& n_wide, uint32_t& depth, bool& ok);
