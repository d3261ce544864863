"""The host step that turns the two-child hierarchy into four-wide nodes with 8-bit child boxes
(ptc_collapse_wide; no GPU needed): every leaf of the input appears exactly once, every stored
child box contains everything below that child, and hierarchies that cannot be represented are
refused."""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import _ffi

BIN = np.dtype([("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3), ("child", "<u4", 2), ("pad", "<u4", 2)])
WIDE = np.dtype([("origin", "<f4", 3), ("scale", "<f4", 3), ("qlo", "u1", (3, 4)), ("qhi", "u1", (3, 4)), ("child", "<u4", 4)])
EMPTY = 0xC0000000


def leaf_ref(kind, first, count):
    return (kind << 30) | ((count - 1) << 27) | first


def random_tree(n_leaves, seed, spread=10.0, size=0.3):
    """Median-split tree over random leaf boxes; returns (nodes, {leaf ref: box})."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n_leaves, 3)).astype(np.float32)
    h = rng.uniform(0.0, size, (n_leaves, 3)).astype(np.float32)
    h[rng.random(n_leaves) < 0.1] = 0.0                      # points: zero extent on every axis
    h[rng.random(n_leaves) < 0.1, 1] = 0.0                   # flat boxes
    lo, hi = c - h, c + h
    counts = rng.integers(1, 4, n_leaves)
    firsts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    refs = [leaf_ref(1 if i % 7 else 2, int(firsts[i]), int(counts[i])) for i in range(n_leaves)]
    nodes, boxes = [], {refs[i]: (lo[i], hi[i]) for i in range(n_leaves)}

    def build(ids):
        if len(ids) == 1:
            return refs[ids[0]], lo[ids[0]], hi[ids[0]]
        me = len(nodes)
        nodes.append(None)
        ax = int(np.argmax(c[ids].max(0) - c[ids].min(0)))
        order = sorted(ids, key=lambda i: c[i, ax])
        mid = len(order) // 2
        (r0, l0, h0), (r1, l1, h1) = build(order[:mid]), build(order[mid:])
        rec = np.zeros((), BIN)
        rec["lo0"], rec["hi0"], rec["lo1"], rec["hi1"], rec["child"] = l0, h0, l1, h1, (r0, r1)
        nodes[me] = rec
        return me, np.minimum(l0, l1), np.maximum(h0, h1)

    build(list(range(n_leaves)))
    return np.array(nodes, dtype=BIN), boxes


def collapse(bin_nodes):
    lib = _ffi.core_lib()
    lib.ptc_collapse_wide.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    out = np.zeros(2 * len(bin_nodes) + 4, dtype=WIDE)
    n, depth = C.c_uint32(0), C.c_uint32(0)
    rc = lib.ptc_collapse_wide(bin_nodes.ctypes.data, len(bin_nodes), out.ctypes.data, len(out), C.byref(n), C.byref(depth))
    return rc, out[:n.value], depth.value


def check(wide, boxes):
    seen = []

    def below(i):
        """Union of the original leaf boxes below wide node i, checking every child entry on the way."""
        nd = wide[i]
        o, s = nd["origin"].astype(np.float64), nd["scale"].astype(np.float64)
        assert np.all(np.log2(s) == np.round(np.log2(s))), "scales are powers of two"
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        assert any(int(r) != EMPTY for r in nd["child"])
        for c in range(4):
            ref = int(nd["child"][c])
            if ref == EMPTY:
                continue
            if ref >> 30 == 0:
                lo, hi = below(ref)
            else:
                lo, hi = (b.astype(np.float64) for b in boxes[ref])
                seen.append(ref)
            qlo, qhi = o + nd["qlo"][:, c] * s, o + nd["qhi"][:, c] * s
            assert np.all(qlo <= lo) and np.all(qhi >= hi), (i, c)
            # and not looser than one grid step on either side
            assert np.all(qlo > lo - s) and np.all(qhi < hi + s), (i, c)
            lo_all, hi_all = np.minimum(lo_all, lo), np.maximum(hi_all, hi)
        return lo_all, hi_all

    below(0)
    assert sorted(seen) == sorted(boxes), "every leaf exactly once"


@pytest.mark.parametrize("n_leaves,seed,spread,size", [(2, 1, 1.0, 0.1), (3, 2, 1.0, 0.5), (50, 3, 10.0, 0.3), (2000, 4, 100.0, 0.01), (700, 5, 1e-3, 1e-4), (300, 6, 1e6, 10.0)])
def test_wide_nodes_cover_the_binary_hierarchy(n_leaves, seed, spread, size):
    bin_nodes, boxes = random_tree(n_leaves, seed, spread, size)
    rc, wide, depth = collapse(bin_nodes)
    assert rc == 0
    assert 1 <= len(wide) <= len(bin_nodes)
    assert 1 <= depth <= 64
    check(wide, boxes)


def test_a_single_child_root_and_empty_children():
    rec = np.zeros(1, BIN)
    rec[0]["lo0"], rec[0]["hi0"] = (-1, -2, -3), (1, 2, 3)
    rec[0]["child"] = (leaf_ref(1, 0, 2), EMPTY)
    rc, wide, depth = collapse(rec)
    assert rc == 0 and len(wide) == 1 and depth == 1
    assert [int(r) for r in wide[0]["child"]] == [leaf_ref(1, 0, 2), EMPTY, EMPTY, EMPTY]
    check(wide, {leaf_ref(1, 0, 2): (np.float32([-1, -2, -3]), np.float32([1, 2, 3]))})


@pytest.mark.parametrize("bad", [np.inf, np.nan, 3e38])
def test_unrepresentable_boxes_are_refused(bad):
    bin_nodes, _ = random_tree(20, 9)
    bin_nodes[3]["hi1"][1] = bad
    if bad == 3e38:
        bin_nodes[3]["lo1"][1] = -3e38
    rc, _, _ = collapse(bin_nodes)
    assert rc == _ffi.ERR_UNSUPPORTED


def test_arguments_are_checked():
    lib = _ffi.core_lib()
    lib.ptc_collapse_wide.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    n, d = C.c_uint32(0), C.c_uint32(0)
    bin_nodes, _ = random_tree(40, 10)
    out = np.zeros(2, dtype=WIDE)
    assert lib.ptc_collapse_wide(None, 1, out.ctypes.data, 2, C.byref(n), C.byref(d)) == _ffi.ERR_INVALID
    assert lib.ptc_collapse_wide(bin_nodes.ctypes.data, len(bin_nodes), out.ctypes.data, 2, C.byref(n), C.byref(d)) == _ffi.ERR_INVALID  # too small
