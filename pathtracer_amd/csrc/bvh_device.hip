// bvh_device.hip -- hierarchy construction steps that run on a binary tree in device memory.
//
// Stands in, together with bvh_build.hip, for the closed-source "Trbvh" builder of the reference
// (Renderer.cpp:202-217), which also builds on the device.  Nothing here looks at the tree from the host:
// the host only sequences launches and reads back a handful of words (tree depth, node count, "did this
// pass change anything").
//
//   sweep_build           top-down surface-area sweep, all nodes of a level at once (segmented scans)
//   tree_rotate           bottom-up rotations, all nodes of a level at once
//   tree_emit             leaf collapse, pre-order numbering, 64-byte two-box nodes, primitive slots
//   wide_collapse_device  four-wide nodes with 8-bit child boxes, level by level
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "bvh_device.h"

#define BB 256
#define HIPCHK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) return std::string(#x) + ": " + hipGetErrorString(e_);     \
    } while (0)
#define TAKE(var, type, count)                                                           \
    type* var = a.take<type>(count);                                                     \
    if (!var) return std::string("BVH build: work space exhausted at " #var)

namespace {

inline uint32_t nblocks(size_t n) { return (uint32_t)((n + BB - 1) / BB); }

#define DEPTH_LIMIT 4096   /* a tree this deep is refused long before (traversal stack); the walks stop here */

__host__ __device__ inline float box_area(const Box6& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__host__ __device__ inline Box6 box_union(const Box6& x, const Box6& y) {
    Box6 u;
    for (int k = 0; k < 3; ++k) {
        u.lo[k] = fminf(x.lo[k], y.lo[k]);
        u.hi[k] = fmaxf(x.hi[k], y.hi[k]);
    }
    return u;
}

// ------------------------------------------------------------------ tree walks --
__global__ void k_tree_parents(const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ root, int n, int* __restrict__ parent) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    const int v = n + i;
    parent[left[i]] = v;
    parent[right[i]] = v;
    if (v == *root) parent[v] = -1;
}

// depth[i] of internal node n+i (root 0); the deepest one in max_depth; *bad when a walk does not end
__global__ void k_tree_depths(const int* __restrict__ parent, int n, int* __restrict__ depth, int* __restrict__ max_depth, int* __restrict__ bad) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    int d = 0;
    int a = parent[n + i];
    while (a >= 0 && d < DEPTH_LIMIT) {
        ++d;
        a = parent[a];
    }
    if (d >= DEPTH_LIMIT) *bad = 1;
    depth[i] = d;
    atomicMax(max_depth, d);
}

// One level of the bottom-up size pass: primitives below every node, and how many internal nodes below it (itself
// included) stay nodes after the leaf collapse.
__global__ void k_tree_sizes(const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ depth, int n, int d, int* __restrict__ size,
                             int* __restrict__ kept) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1 || depth[i] != d) return;
    const int l = left[i], r = right[i];
    const int sz = (l < n ? 1 : size[l]) + (r < n ? 1 : size[r]);
    size[n + i] = sz;
    kept[n + i] = (sz > BVH_LEAF_MAX ? 1 : 0) + (l < n ? 0 : kept[l]) + (r < n ? 0 : kept[r]);
}

// Per node: the slot of its leftmost primitive when slots go left to right through the leaves, and its number in a
// pre-order walk over the nodes that stay nodes.  Both are sums over the node's ancestors, taken on the way up.
__global__ void k_tree_positions(const int* __restrict__ parent, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ size,
                                 const int* __restrict__ kept, int n, int* __restrict__ first, int* __restrict__ pre) {
    const int v = blockIdx.x * BB + threadIdx.x;
    if (v >= 2 * n - 1) return;
    int f = 0, p = 0, a = v, steps = 0;
    for (;;) {
        const int q = parent[a];
        if (q < 0 || ++steps > DEPTH_LIMIT) break;
        ++p;  // the ancestor itself comes before everything below it
        if (right[q - n] == a) {
            const int l = left[q - n];
            f += l < n ? 1 : size[l];
            p += l < n ? 0 : kept[l];
        }
        a = q;
    }
    first[v] = f;
    pre[v] = p;
}

__global__ void k_tree_emit(const int* __restrict__ left, const int* __restrict__ right, const Box6* __restrict__ box, const int* __restrict__ size,
                            const int* __restrict__ first, const int* __restrict__ pre, const int* __restrict__ depth, int n, uint32_t kind, uint32_t node_base,
                            BvhNode* __restrict__ nodes, int* __restrict__ max_kept_depth) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    const int v = n + i;
    if (size[v] <= BVH_LEAF_MAX) return;
    BvhNode nd;
    const int ch[2] = {left[i], right[i]};
    for (int c = 0; c < 2; ++c) {
        const int u = ch[c];
        const Box6 b = box[u];
        const int sz = u < n ? 1 : size[u];
        float* lo = c == 0 ? nd.lo0 : nd.lo1;
        float* hi = c == 0 ? nd.hi0 : nd.hi1;
        for (int k = 0; k < 3; ++k) {
            lo[k] = b.lo[k];
            hi[k] = b.hi[k];
        }
        nd.child[c] = sz > BVH_LEAF_MAX ? node_base + (uint32_t)pre[u] : MAKE_LEAF(kind, (uint32_t)first[u], (uint32_t)sz);
    }
    nd.pad[0] = nd.pad[1] = 0u;
    nodes[node_base + (uint32_t)pre[v]] = nd;
    atomicMax(max_kept_depth, depth[i] + 1);
}

__global__ void k_tree_slots(const int* __restrict__ first, const uint32_t* __restrict__ leaf_prim, int n, int32_t* __restrict__ slot_prim) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    slot_prim[first[i]] = (int32_t)leaf_prim[i];
}

// ------------------------------------------------------------------- rotations --
// One level of a rotation sweep (the nodes whose depth at the start of the sweep is d; deeper levels are through).
// Nodes of one level have disjoint subtrees, and a rotation only touches the node, its children and grandchildren.
__global__ void k_tree_rotate_level(int* __restrict__ left, int* __restrict__ right, Box6* __restrict__ box, const int* __restrict__ depth, int n, int d,
                                    int* __restrict__ applied) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1 || depth[i] != d) return;
    int L = left[i], R = right[i];
    // candidates: (which child is rebuilt, which of its children goes up)
    float best = 0.f;
    int pick = -1;
    if (L >= n) {
        const int A = left[L - n], B = right[L - n];
        const float old = box_area(box[L]);
        const float a0 = box_area(box_union(box[R], box[B])), a1 = box_area(box_union(box[A], box[R]));  // swap R<->A, R<->B
        if (old - a0 > best) { best = old - a0; pick = 0; }
        if (old - a1 > best) { best = old - a1; pick = 1; }
    }
    if (R >= n) {
        const int C = left[R - n], D = right[R - n];
        const float old = box_area(box[R]);
        const float a2 = box_area(box_union(box[L], box[D])), a3 = box_area(box_union(box[C], box[L]));  // swap L<->C, L<->D
        if (old - a2 > best) { best = old - a2; pick = 2; }
        if (old - a3 > best) { best = old - a3; pick = 3; }
    }
    if (pick < 0) return;
    atomicAdd(applied, 1);
    if (pick == 0) {  // L = (R, B), v = (L, A)
        const int A = left[L - n];
        left[L - n] = R;
        right[i] = A;
        box[L] = box_union(box[R], box[right[L - n]]);
    } else if (pick == 1) {  // L = (A, R), v = (L, B)
        const int B = right[L - n];
        right[L - n] = R;
        right[i] = B;
        box[L] = box_union(box[left[L - n]], box[R]);
    } else if (pick == 2) {  // R = (L, D), v = (C, R)
        const int C = left[R - n];
        left[R - n] = L;
        left[i] = C;
        box[R] = box_union(box[L], box[right[R - n]]);
    } else {  // R = (C, L), v = (D, R)
        const int D = right[R - n];
        right[R - n] = L;
        left[i] = D;
        box[R] = box_union(box[left[R - n]], box[L]);
    }
}

// ------------------------------------------------------------------ wide nodes --
struct WideChild {
    uint32_t ref;
    Box6 box;
};

__device__ inline int bin_children(const BvhNode& nd, WideChild* dst) {
    int k = 0;
    for (int c = 0; c < 2; ++c) {
        if (REF_KIND(nd.child[c]) == REF_EMPTY) continue;
        dst[k].ref = nd.child[c];
        const float* lo = c == 0 ? nd.lo0 : nd.lo1;
        const float* hi = c == 0 ? nd.hi0 : nd.hi1;
        for (int a = 0; a < 3; ++a) {
            dst[k].box.lo[a] = lo[a];
            dst[k].box.hi[a] = hi[a];
        }
        ++k;
    }
    return k;
}

// The wide nodes range[0] .. range[1]-1 (one level of the wide tree): each takes its binary node from bin_of[], chooses
// its children, reserves the numbers of the internal ones (consecutive) and notes their binary nodes for the next level.
__global__ void k_wide_level(const BvhNode* __restrict__ bin, uint32_t n_bin, uint32_t unused_ref, WideNode* __restrict__ wide, uint32_t* __restrict__ bin_of,
                             const uint32_t* __restrict__ range, uint32_t* __restrict__ n_wide, int* __restrict__ fail) {
    const uint32_t w = range[0] + blockIdx.x * BB + threadIdx.x;
    if (w >= range[1]) return;
    const uint32_t b = bin_of[w];
    WideChild ch[4], tmp[2];
    int nch = bin_children(bin[b], ch);
    for (;;) {
        int pick = -1;
        float best = -1.f;
        for (int k = 0; k < nch; ++k) {
            if (REF_KIND(ch[k].ref) != REF_INTERNAL) continue;
            const uint32_t cb = REF_NODE(ch[k].ref);
            if (cb >= n_bin) {
                *fail = 1;
                return;
            }
            const int grand = (REF_KIND(bin[cb].child[0]) != REF_EMPTY ? 1 : 0) + (REF_KIND(bin[cb].child[1]) != REF_EMPTY ? 1 : 0);
            if (nch - 1 + grand > 4) continue;
            const float area = box_area(ch[k].box);
            if (area > best) {
                best = area;
                pick = k;
            }
        }
        if (pick < 0) break;
        const int nt = bin_children(bin[REF_NODE(ch[pick].ref)], tmp);
        for (int k = pick; k + 1 < nch; ++k) ch[k] = ch[k + 1];
        --nch;
        for (int k = 0; k < nt; ++k) ch[nch++] = tmp[k];
    }
    // storage order: smallest box first (the order in which any-hit rays visit the children they hit); stable
    for (int k = 1; k < nch; ++k) {
        const WideChild c = ch[k];
        const float ac = box_area(c.box);
        int j = k;
        while (j > 0 && ac < box_area(ch[j - 1].box)) {
            ch[j] = ch[j - 1];
            --j;
        }
        ch[j] = c;
    }
    WideNode out;
    memset(&out, 0, sizeof(out));
    // unused child slots: an inside-out box no ray enters, and a reference to a triangle no ray hits
    for (int c = 0; c < 4; ++c) {
        out.child[c] = unused_ref;
        if (unused_ref != REF_EMPTY_WORD) {
            out.qlo_x[c] = out.qlo_y[c] = out.qlo_z[c] = 255;
            out.qhi_x[c] = out.qhi_y[c] = out.qhi_z[c] = 0;
        }
    }
    if (nch > 0) {
        Box6 u = ch[0].box;
        for (int k = 1; k < nch; ++k) u = box_union(u, ch[k].box);
        float scale[3];
        for (int k = 0; k < 3; ++k) {
            if (!isfinite(u.lo[k]) || !isfinite(u.hi[k])) {
                *fail = 1;
                return;
            }
            const double ext = (double)u.hi[k] - (double)u.lo[k];
            int e = -100;
            if (ext > 0.0) {
                int ex;
                const double m = frexp(ext / 255.0, &ex);  // ext/255 = m * 2^ex, m in [0.5, 1)
                e = m == 0.5 ? ex - 1 : ex;
            }
            e = min(max(e, -100), 100);
            scale[k] = ldexpf(1.0f, e);
            if ((double)scale[k] * 255.0 < ext) {
                *fail = 1;
                return;
            }
            out.origin[k] = u.lo[k];
        }
        out.scale_x = scale[0];
        out.scale_y = scale[1];
        out.scale_z = scale[2];
        int n_internal = 0;
        for (int c = 0; c < nch; ++c) n_internal += REF_KIND(ch[c].ref) == REF_INTERNAL ? 1 : 0;
        uint32_t next = n_internal ? atomicAdd(n_wide, (uint32_t)n_internal) : 0u;
        for (int c = 0; c < nch; ++c) {
            uint8_t* qlo[3] = {out.qlo_x, out.qlo_y, out.qlo_z};
            uint8_t* qhi[3] = {out.qhi_x, out.qhi_y, out.qhi_z};
            for (int k = 0; k < 3; ++k) {
                const double o = (double)u.lo[k], sc = (double)scale[k];
                if (!isfinite(ch[c].box.lo[k]) || !isfinite(ch[c].box.hi[k])) {
                    *fail = 1;
                    return;
                }
                double lo = floor(((double)ch[c].box.lo[k] - o) / sc), hi = ceil(((double)ch[c].box.hi[k] - o) / sc);
                lo = fmin(fmax(lo, 0.0), 255.0);
                hi = fmin(fmax(hi, 0.0), 255.0);
                if (o + lo * sc > (double)ch[c].box.lo[k] || o + hi * sc < (double)ch[c].box.hi[k]) {
                    *fail = 1;
                    return;
                }
                qlo[k][c] = (uint8_t)lo;
                qhi[k][c] = (uint8_t)hi;
            }
            if (REF_KIND(ch[c].ref) == REF_INTERNAL) {
                if (next >= n_bin) {  // more wide nodes than binary ones: the input is no tree
                    *fail = 1;
                    return;
                }
                bin_of[next] = REF_NODE(ch[c].ref);
                out.child[c] = next++;
            } else {
                out.child[c] = ch[c].ref;
            }
        }
    }
    wide[w] = out;
}

// Between two levels (one thread): the nodes reserved during the last level are the next one.
__global__ void k_wide_next(uint32_t* __restrict__ range, const uint32_t* __restrict__ n_wide, uint32_t* __restrict__ depth) {
    if (range[1] > range[0]) *depth += 1u;
    range[0] = range[1];
    range[1] = *n_wide;
}

// ------------------------------------------------------------------------ sweep --
struct Box3 {
    Box6 b[3];
};
struct Int3 {
    int v[3];
};
struct UnionBox3 {
    __host__ __device__ Box3 operator()(const Box3& x, const Box3& y) const {
        Box3 u;
        for (int k = 0; k < 3; ++k) u.b[k] = box_union(x.b[k], y.b[k]);
        return u;
    }
};
struct PlusInt3 {
    __host__ __device__ Int3 operator()(const Int3& x, const Int3& y) const {
        Int3 s;
        for (int k = 0; k < 3; ++k) s.v[k] = x.v[k] + y.v[k];
        return s;
    }
};
struct MinU64 {
    __host__ __device__ unsigned long long operator()(unsigned long long x, unsigned long long y) const { return x < y ? x : y; }
};
// element j of a scan's input: position j of the three orders, or position n-1-j (suffix scans run over the reversed arrays)
struct LoadBox3 {
    const int *o0, *o1, *o2;
    const Box6* box;
    int n, rev;
    __host__ __device__ Box3 operator()(int j) const {
        const int p = rev ? n - 1 - j : j;
        Box3 r;
        r.b[0] = box[o0[p]];
        r.b[1] = box[o1[p]];
        r.b[2] = box[o2[p]];
        return r;
    }
};
struct LoadKey {
    const int* seg_first;
    int n, rev;
    __host__ __device__ int operator()(int j) const { return seg_first[rev ? n - 1 - j : j]; }
};
struct LoadSide {
    const int *o0, *o1, *o2;
    const int* side;
    __host__ __device__ Int3 operator()(int p) const {
        Int3 r;
        r.v[0] = side[o0[p]];
        r.v[1] = side[o1[p]];
        r.v[2] = side[o2[p]];
        return r;
    }
};

#define SWEEP_INVALID 0xFFFFFFFFFFFFFFFFull
#define SWEEP_MAX_LEAVES (1 << 20)   /* 20 bits of "how uneven" in the candidate word */
#define SWEEP_MAX_LEVELS 192

// floats -> unsigned integers that sort the same way
__device__ inline uint32_t sortable(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void k_sweep_init(const Box6* __restrict__ boxes, const uint32_t* __restrict__ ids, int n, float pad, Box6* __restrict__ box,
                             uint32_t* __restrict__ key0, uint32_t* __restrict__ key1, uint32_t* __restrict__ key2, int* __restrict__ leaf, int* __restrict__ seg_first,
                             int* __restrict__ cnt, int* __restrict__ par) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    Box6 b = boxes[ids[i]];
    key0[i] = sortable(0.5f * (b.lo[0] + b.hi[0]));
    key1[i] = sortable(0.5f * (b.lo[1] + b.hi[1]));
    key2[i] = sortable(0.5f * (b.lo[2] + b.hi[2]));
    for (int k = 0; k < 3; ++k) {
        b.lo[k] -= pad;
        b.hi[k] += pad;
    }
    box[i] = b;
    leaf[i] = i;
    seg_first[i] = 0;
    if (i == 0) {
        cnt[0] = n;
        par[0] = -1;
    }
}

// The split "left = the first nl leaves of the segment along axis k", as one word that orders like (cost, how uneven,
// axis, position).
__device__ inline unsigned long long sweep_word(float cost, int nl, int count, int axis) {
    const int uneven = abs(2 * nl - count);
    return ((unsigned long long)__float_as_uint(cost) << 23) | ((unsigned long long)uneven << 3) | ((unsigned long long)axis << 1) | (2 * nl > count ? 1ull : 0ull);
}
__device__ inline void sweep_decode(unsigned long long w, int count, int& axis, int& nl) {
    if (w == SWEEP_INVALID) {  // no split has a finite cost: halves, in the order of axis 0
        axis = 0;
        nl = count / 2;
        return;
    }
    const int uneven = (int)((w >> 3) & 0xFFFFFu);
    axis = (int)((w >> 1) & 3u);
    nl = (w & 1ull) ? (count + uneven) / 2 : (count - uneven) / 2;
}

// Candidate at position p: the split of p's segment that puts the leaves before p to the left, best of the three axes.
__global__ void k_sweep_candidates(const int* __restrict__ seg_first, const int* __restrict__ cnt, const Box3* __restrict__ pre, const Box3* __restrict__ suf_rev, int n,
                                   unsigned long long* __restrict__ cand) {
    const int p = blockIdx.x * BB + threadIdx.x;
    if (p >= n) return;
    const int first = seg_first[p], count = cnt[first];
    unsigned long long best = SWEEP_INVALID;
    if (p > first) {
        const int nl = p - first, nr = count - nl;
        const Box3 a = pre[p - 1], b = suf_rev[n - 1 - p];
        for (int k = 0; k < 3; ++k) {
            const float cost = box_area(a.b[k]) * (float)nl + box_area(b.b[k]) * (float)nr;
            if (!(cost < INFINITY)) continue;  // a split is only taken for a finite cost
            const unsigned long long w = sweep_word(cost, nl, count, k);
            best = w < best ? w : best;
        }
    }
    cand[p] = best;
}

// Which side every leaf of a segment that is split goes to: position p speaks for the leaf at p in the order of its
// segment's chosen axis.
__global__ void k_sweep_sides(const int* __restrict__ seg_first, const int* __restrict__ cnt, const unsigned long long* __restrict__ seg_min, const int* __restrict__ o0,
                              const int* __restrict__ o1, const int* __restrict__ o2, int n, int* __restrict__ side) {
    const int p = blockIdx.x * BB + threadIdx.x;
    if (p >= n) return;
    const int first = seg_first[p], count = cnt[first];
    if (count < 2) return;
    int axis, nl;
    sweep_decode(seg_min[first + count - 1], count, axis, nl);
    const int leaf = (axis == 0 ? o0 : axis == 1 ? o1 : o2)[p];
    side[leaf] = p - first < nl ? 1 : 0;
}

// Stable partition of the three orders inside every segment, the segments of the next level, and the node of every
// segment that was split (its number is n + the position of the last leaf of its left part, which no other node shares).
__global__ void k_sweep_split(const int* __restrict__ seg_first, const int* __restrict__ cnt, const int* __restrict__ par, const unsigned long long* __restrict__ seg_min,
                              const Box3* __restrict__ pre, const Int3* __restrict__ lefts_before, const int* __restrict__ side, const int* __restrict__ o0,
                              const int* __restrict__ o1, const int* __restrict__ o2, int n, int* __restrict__ o0_new, int* __restrict__ o1_new, int* __restrict__ o2_new,
                              int* __restrict__ seg_first_new, int* __restrict__ cnt_new, int* __restrict__ par_new, int* __restrict__ left, int* __restrict__ right,
                              Box6* __restrict__ box, int* __restrict__ root, int* __restrict__ open_segments) {
    const int p = blockIdx.x * BB + threadIdx.x;
    if (p >= n) return;
    const int first = seg_first[p], count = cnt[first];
    if (count < 2) {  // a finished leaf
        o0_new[p] = o0[p];
        o1_new[p] = o1[p];
        o2_new[p] = o2[p];
        seg_first_new[p] = first;
        cnt_new[p] = 1;
        return;
    }
    int axis, nl;
    sweep_decode(seg_min[first + count - 1], count, axis, nl);
    const Int3 lb = lefts_before[p];
    const int* const o[3] = {o0, o1, o2};
    int* const o_new[3] = {o0_new, o1_new, o2_new};
    for (int k = 0; k < 3; ++k) {
        const int leaf = o[k][p];
        const int at = side[leaf] ? first + lb.v[k] : first + nl + (p - first - lb.v[k]);
        o_new[k][at] = leaf;
    }
    seg_first_new[p] = p - first < nl ? first : first + nl;
    if (p != first) return;
    const int idx = first + nl - 1, id = n + idx;
    box[id] = pre[first + count - 1].b[0];  // the union of the segment's boxes (the same along every axis)
    const int pr = par[first];
    if (pr < 0) *root = id;
    else if (pr & 1) right[pr >> 1] = id;
    else left[pr >> 1] = id;
    const int* const oa = o[axis];
    int open = 0;
    if (nl == 1) {
        left[idx] = oa[first];
        cnt_new[first] = 1;
    } else {
        cnt_new[first] = nl;
        par_new[first] = idx << 1;
        ++open;
    }
    const int nr = count - nl;
    if (nr == 1) {
        right[idx] = oa[first + count - 1];
        cnt_new[first + nl] = 1;
    } else {
        cnt_new[first + nl] = nr;
        par_new[first + nl] = (idx << 1) | 1;
        ++open;
    }
    if (open) atomicAdd(open_segments, open);
}

}  // namespace

// ----------------------------------------------------------------------- host side --
void tree_reserve(DeviceArena& a, int n) {
    const size_t N = (size_t)std::max(n, 1);
    a.reserve<int>(2 * N);  // parent
    a.reserve<int>(N);      // depth
    a.reserve<int>(2 * N);  // size
    a.reserve<int>(2 * N);  // kept
    a.reserve<int>(2 * N);  // first
    a.reserve<int>(2 * N);  // pre
    a.reserve<int>(8);      // words
    // tree_rotate takes parent / depth / words of its own
    a.reserve<int>(2 * N);
    a.reserve<int>(N);
    a.reserve<int>(8);
}

void sweep_reserve(DeviceArena& a, int n) {
    const size_t N = (size_t)std::max(n, 1);
    a.reserve<int>(N);           // left
    a.reserve<int>(N);           // right
    a.reserve<Box6>(2 * N);      // box
    a.reserve<int>(8);           // root + words
    for (int k = 0; k < 6; ++k) a.reserve<int>(N);       // three orders, twice
    for (int k = 0; k < 3; ++k) a.reserve<uint32_t>(N);  // sort keys
    a.reserve<uint32_t>(N);      // sort keys out
    a.reserve<int>(N);           // leaf numbers
    for (int k = 0; k < 6; ++k) a.reserve<int>(N);       // segment first / count / parent, twice
    a.reserve<Box3>(N);
    a.reserve<Box3>(N);
    a.reserve<unsigned long long>(N);
    a.reserve<unsigned long long>(N);
    a.reserve<int>(N);           // side
    a.reserve<Int3>(N);
}

std::string sweep_build(hipStream_t s, DeviceArena& a, const Box6* boxes, const uint32_t* ids, int n, float pad, DeviceTree& t) {
    if (n < 2) return "BVH build: the sweep needs at least two primitives";
    if (n > SWEEP_MAX_LEAVES) return "BVH build: too many primitives for the sweep builder";
    const size_t N = (size_t)n;
    TAKE(left, int, N);
    TAKE(right, int, N);
    TAKE(box, Box6, 2 * N);
    TAKE(words, int, 8);  // [0] root, [1] segments of the next level that still have to be split
    int* o[2][3];
    for (int b = 0; b < 2; ++b)
        for (int k = 0; k < 3; ++k) {
            o[b][k] = a.take<int>(N);
            if (!o[b][k]) return "BVH build: work space exhausted";
        }
    uint32_t* key[3];
    for (int k = 0; k < 3; ++k) {
        key[k] = a.take<uint32_t>(N);
        if (!key[k]) return "BVH build: work space exhausted";
    }
    TAKE(key_out, uint32_t, N);
    TAKE(leaf, int, N);
    int *seg_first[2], *cnt[2], *par[2];
    for (int b = 0; b < 2; ++b) {
        seg_first[b] = a.take<int>(N);
        cnt[b] = a.take<int>(N);
        par[b] = a.take<int>(N);
        if (!seg_first[b] || !cnt[b] || !par[b]) return "BVH build: work space exhausted";
    }
    TAKE(pre, Box3, N);
    TAKE(suf_rev, Box3, N);
    TAKE(cand, unsigned long long, N);
    TAKE(seg_min, unsigned long long, N);
    TAKE(side, int, N);
    TAKE(lefts_before, Int3, N);

    hipLaunchKernelGGL(k_sweep_init, dim3(nblocks(N)), dim3(BB), 0, s, boxes, ids, n, pad, box, key[0], key[1], key[2], leaf, seg_first[0], cnt[0], par[0]);
    const auto count_it = rocprim::make_counting_iterator<int>(0);
    // work space of the library calls: the largest request
    size_t temp_bytes = 0;
    {
        size_t b = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, b, key[0], key_out, leaf, o[0][0], N, 0, 32, s));
        temp_bytes = std::max(temp_bytes, b);
        const LoadBox3 lb{o[0][0], o[0][1], o[0][2], box, n, 0};
        const LoadKey lk{seg_first[0], n, 0};
        b = 0;
        HIPCHK(rocprim::inclusive_scan_by_key(nullptr, b, rocprim::make_transform_iterator(count_it, lk), rocprim::make_transform_iterator(count_it, lb), pre, N,
                                              UnionBox3(), rocprim::equal_to<int>(), s));
        temp_bytes = std::max(temp_bytes, b);
        b = 0;
        HIPCHK(rocprim::inclusive_scan_by_key(nullptr, b, seg_first[0], cand, seg_min, N, MinU64(), rocprim::equal_to<int>(), s));
        temp_bytes = std::max(temp_bytes, b);
        const LoadSide ls{o[0][0], o[0][1], o[0][2], side};
        b = 0;
        HIPCHK(rocprim::exclusive_scan_by_key(nullptr, b, seg_first[0], rocprim::make_transform_iterator(count_it, ls), lefts_before, Int3{{0, 0, 0}}, N, PlusInt3(),
                                              rocprim::equal_to<int>(), s));
        temp_bytes = std::max(temp_bytes, b);
    }
    void* temp = nullptr;
    HIPCHK(hipMalloc(&temp, std::max<size_t>(temp_bytes, 256)));
    struct FreeTemp {
        void* p;
        ~FreeTemp() { (void)hipFree(p); }
    } free_temp{temp};
    // the three orders: leaves by centroid, equal centroids by leaf number (the sort is stable)
    for (int k = 0; k < 3; ++k) {
        size_t b = temp_bytes;
        HIPCHK(rocprim::radix_sort_pairs(temp, b, key[k], key_out, leaf, o[0][k], N, 0, 32, s));
    }
    int cur = 0;
    for (int level = 0;; ++level) {
        if (level >= SWEEP_MAX_LEVELS) return "BVH build: sweep hierarchy too deep";
        const int nxt = cur ^ 1;
        const LoadBox3 fwd{o[cur][0], o[cur][1], o[cur][2], box, n, 0}, bwd{o[cur][0], o[cur][1], o[cur][2], box, n, 1};
        const LoadKey kf{seg_first[cur], n, 0}, kb{seg_first[cur], n, 1};
        size_t b = temp_bytes;
        HIPCHK(rocprim::inclusive_scan_by_key(temp, b, rocprim::make_transform_iterator(count_it, kf), rocprim::make_transform_iterator(count_it, fwd), pre, N,
                                              UnionBox3(), rocprim::equal_to<int>(), s));
        b = temp_bytes;
        HIPCHK(rocprim::inclusive_scan_by_key(temp, b, rocprim::make_transform_iterator(count_it, kb), rocprim::make_transform_iterator(count_it, bwd), suf_rev, N,
                                              UnionBox3(), rocprim::equal_to<int>(), s));
        hipLaunchKernelGGL(k_sweep_candidates, dim3(nblocks(N)), dim3(BB), 0, s, seg_first[cur], cnt[cur], pre, suf_rev, n, cand);
        b = temp_bytes;
        HIPCHK(rocprim::inclusive_scan_by_key(temp, b, seg_first[cur], cand, seg_min, N, MinU64(), rocprim::equal_to<int>(), s));
        hipLaunchKernelGGL(k_sweep_sides, dim3(nblocks(N)), dim3(BB), 0, s, seg_first[cur], cnt[cur], seg_min, o[cur][0], o[cur][1], o[cur][2], n, side);
        const LoadSide ls{o[cur][0], o[cur][1], o[cur][2], side};
        b = temp_bytes;
        HIPCHK(rocprim::exclusive_scan_by_key(temp, b, seg_first[cur], rocprim::make_transform_iterator(count_it, ls), lefts_before, Int3{{0, 0, 0}}, N, PlusInt3(),
                                              rocprim::equal_to<int>(), s));
        HIPCHK(hipMemsetAsync(words + 1, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_sweep_split, dim3(nblocks(N)), dim3(BB), 0, s, seg_first[cur], cnt[cur], par[cur], seg_min, pre, lefts_before, side, o[cur][0], o[cur][1],
                           o[cur][2], n, o[nxt][0], o[nxt][1], o[nxt][2], seg_first[nxt], cnt[nxt], par[nxt], left, right, box, words, words + 1);
        int open = 0;
        HIPCHK(hipMemcpyAsync(&open, words + 1, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        cur = nxt;
        if (open == 0) break;
    }
    HIPCHK(hipGetLastError());
    t.n = n;
    t.left = left;
    t.right = right;
    t.box = box;
    t.root = words;
    t.leaf_prim = ids;
    return "";
}

std::string tree_rotate(hipStream_t s, DeviceArena& a, DeviceTree& t, int passes) {
    const int n = t.n;
    if (n < 3 || passes <= 0) return "";
    TAKE(parent, int, 2 * (size_t)n);
    TAKE(depth, int, (size_t)n);
    TAKE(words, int, 8);  // [0] deepest level, [1] malformed, [2] rotations applied
    for (int pass = 0; pass < passes; ++pass) {
        HIPCHK(hipMemsetAsync(words, 0, 3 * sizeof(int), s));
        hipLaunchKernelGGL(k_tree_parents, dim3(nblocks(n - 1)), dim3(BB), 0, s, t.left, t.right, t.root, n, parent);
        hipLaunchKernelGGL(k_tree_depths, dim3(nblocks(n - 1)), dim3(BB), 0, s, parent, n, depth, words, words + 1);
        int h[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(h, words, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (h[1]) return "BVH build: malformed tree";
        for (int d = h[0]; d >= 0; --d) hipLaunchKernelGGL(k_tree_rotate_level, dim3(nblocks(n - 1)), dim3(BB), 0, s, t.left, t.right, t.box, depth, n, d, words + 2);
        int applied = 0;
        HIPCHK(hipMemcpyAsync(&applied, words + 2, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (!applied) break;
    }
    HIPCHK(hipGetLastError());
    return "";
}

std::string tree_emit(hipStream_t s, DeviceArena& a, const DeviceTree& t, uint32_t kind, uint32_t node_base, BvhNode* nodes, int32_t* slot_prim, uint32_t& n_nodes,
                      uint32_t& max_depth, Box6& root_box) {
    const int n = t.n;
    if (n < 2) return "BVH build: a tree needs at least two leaves";
    const size_t N = (size_t)n;
    TAKE(parent, int, 2 * N);
    TAKE(depth, int, N);
    TAKE(size, int, 2 * N);
    TAKE(kept, int, 2 * N);
    TAKE(first, int, 2 * N);
    TAKE(pre, int, 2 * N);
    TAKE(words, int, 8);  // [0] deepest level, [1] malformed, [2] levels of kept nodes
    HIPCHK(hipMemsetAsync(words, 0, 3 * sizeof(int), s));
    hipLaunchKernelGGL(k_tree_parents, dim3(nblocks(n - 1)), dim3(BB), 0, s, t.left, t.right, t.root, n, parent);
    hipLaunchKernelGGL(k_tree_depths, dim3(nblocks(n - 1)), dim3(BB), 0, s, parent, n, depth, words, words + 1);
    int h[2] = {0, 0}, root = 0;
    HIPCHK(hipMemcpyAsync(h, words, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&root, t.root, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h[1] || root < n || root > 2 * n - 2) return "BVH build: malformed tree";
    for (int d = h[0]; d >= 0; --d) hipLaunchKernelGGL(k_tree_sizes, dim3(nblocks(n - 1)), dim3(BB), 0, s, t.left, t.right, depth, n, d, size, kept);
    hipLaunchKernelGGL(k_tree_positions, dim3(nblocks(2 * N - 1)), dim3(BB), 0, s, parent, t.left, t.right, size, kept, n, first, pre);
    hipLaunchKernelGGL(k_tree_emit, dim3(nblocks(n - 1)), dim3(BB), 0, s, t.left, t.right, t.box, size, first, pre, depth, n, kind, node_base, nodes, words + 2);
    hipLaunchKernelGGL(k_tree_slots, dim3(nblocks(N)), dim3(BB), 0, s, first, t.leaf_prim, n, slot_prim);
    int kept_root = 0, size_root = 0, kept_depth = 0;
    HIPCHK(hipMemcpyAsync(&kept_root, kept + root, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&size_root, size + root, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&kept_depth, words + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&root_box, t.box + root, sizeof(Box6), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (size_root != n) return "BVH build: the tree does not hold every primitive";
    n_nodes = (uint32_t)kept_root;
    max_depth = (uint32_t)kept_depth;
    return "";
}

std::string wide_collapse_device(hipStream_t s, const BvhNode* bin, uint32_t n_bin, uint32_t bin_depth, uint32_t unused_ref, WideNode* wide, uint32_t& n_wide,
                                 uint32_t& depth, bool& ok) {
    ok = false;
    n_wide = depth = 0;
    if (n_bin == 0) return "";
    uint32_t* bin_of = nullptr;
    HIPCHK(hipMalloc((void**)&bin_of, ((size_t)n_bin + 8) * sizeof(uint32_t)));
    struct FreeIt {
        void* p;
        ~FreeIt() { (void)hipFree(p); }
    } free_it{bin_of};
    uint32_t* words = bin_of + n_bin;  // [0..1] range of the level, [2] wide nodes so far, [3] levels, [4] failure
    const uint32_t init[5] = {0u, 1u, 1u, 0u, 0u};
    HIPCHK(hipMemsetAsync(bin_of, 0, sizeof(uint32_t), s));  // wide node 0 is binary node 0
    HIPCHK(hipMemcpyAsync(words, init, sizeof(init), hipMemcpyHostToDevice, s));
    // a level of the wide tree takes at least one level of the binary tree
    for (uint32_t level = 0; level < bin_depth + 1u; ++level) {
        hipLaunchKernelGGL(k_wide_level, dim3(nblocks(n_bin)), dim3(BB), 0, s, bin, n_bin, unused_ref, wide, bin_of, words, words + 2, (int*)(words + 4));
        hipLaunchKernelGGL(k_wide_next, dim3(1), dim3(1), 0, s, words, words + 2, words + 3);
    }
    uint32_t h[5];
    HIPCHK(hipMemcpyAsync(h, words, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (h[0] != h[1]) return "BVH build: the wide collapse did not reach every node";
    n_wide = h[2];
    depth = h[3];
    // a ray's stack grows by at most three entries per level
    ok = h[4] == 0u && 3u * depth + 1u <= TRAV_LDS_STACK + TRAV_SPILL_STACK;
    return "";
}
