"""The C-ABI shared library loads, exports every entry point include/ptcore.h
declares, and refuses to work without a GPU instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pathtracer_amd import _ffi
from util import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    return sorted(set(re.findall(r"\b(ptc_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_ffi.CORE_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _ffi.core_lib()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_pod_sizes_match_the_header():
    assert C.sizeof(_ffi.MaterialValue) == 84
    assert C.sizeof(_ffi.Triangle) == 136 and _ffi.Triangle.mv.offset == 48 and _ffi.Triangle.objType.offset == 132
    assert C.sizeof(_ffi.Sphere) == 152 and _ffi.Sphere.mv.offset == 64
    assert C.sizeof(_ffi.QuadLight) == 284 and _ffi.QuadLight.color.offset == 272
    assert C.sizeof(_ffi.Config) == 88 and _ffi.Config.hSize.offset == 48 and _ffi.Config.russian_roul.offset == 80
    assert C.sizeof(_ffi.PointLight) == 36 and C.sizeof(_ffi.DirectionalLight) == 24
    assert C.sizeof(_ffi.Stats) == 8 * 6 + 8 * 4 + 8 * 4 + 4 * 2 + 8 + 8 * 2 + 8 * 4 + 8 * 4 and _ffi.Stats.build_host_ms.offset == 168


def test_no_gpu_means_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _ffi.core_lib()
    ctx = C.c_void_p()
    rc = lib.ptc_create(0, C.byref(ctx))
    assert rc == _ffi.ERR_DEVICE and not ctx.value
    assert b"HIP device" in lib.ptc_last_error(None)
    from pathtracer_amd import Renderer, SceneLoader
    scene = SceneLoader().loadText("size 4 4\n")
    with pytest.raises(RuntimeError, match="HIP device"):
        Renderer(scene)


def test_null_context_is_rejected():
    lib = _ffi.core_lib()
    assert lib.ptc_build(None) == _ffi.ERR_INVALID
    assert lib.ptc_set_integrator(None, b"direct") == _ffi.ERR_INVALID


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under pathtracer_amd/ may refer to it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pathtracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"import\s+oracle|from\s+oracle|oracle_py|liboracle|oracle\.h|oracle/|orc_[a-z]+\(", text), os.path.join(dirpath, f)


def test_collapse_wide_rejects_malformed_nodes():
    """Caller-supplied two-child nodes: a child index outside the array, or a node that is reached
    twice (a cycle included), is PTC_ERR_INVALID, not an out-of-bounds read or an endless loop."""
    lib = _ffi.core_lib()
    lib.ptc_collapse_wide.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    node = np.dtype([("box", "<f4", 12), ("child", "<u4", 2), ("pad", "<u4", 2)])
    assert node.itemsize == 64
    LEAF = (1 << 30) | 0  # one triangle, slot 0

    def run(children):
        n = np.zeros(len(children), dtype=node)
        n["box"][:, 0:3] = 0.0
        n["box"][:, 3:6] = 1.0
        n["box"][:, 6:9] = 0.0
        n["box"][:, 9:12] = 1.0
        for i, ch in enumerate(children):
            n["child"][i] = ch
        out = np.zeros(len(children) * 64 + 64, dtype=np.uint8)
        nw, depth = C.c_uint32(), C.c_uint32()
        return lib.ptc_collapse_wide(n.ctypes.data, len(n), out.ctypes.data, len(children) + 1, C.byref(nw), C.byref(depth))

    assert run([(1, LEAF), (LEAF, LEAF)]) == 0
    assert run([(5, LEAF), (LEAF, LEAF)]) == _ffi.ERR_INVALID            # child outside the array
    assert run([(1, LEAF), (0, LEAF)]) == _ffi.ERR_INVALID               # cycle back to the root
    assert run([(1, 1), (LEAF, LEAF)]) == _ffi.ERR_INVALID               # the same node twice
