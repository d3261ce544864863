"""ctypes binding of the CPU oracle (oracle/oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, the smoke check and the
CPU-baseline leg of bench.py.  The product never imports this."""
import ctypes as C
import os

import numpy as np

from pathtracer_amd import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
INTEGRATORS = {"raytracer": 0, "analyticdirect": 1, "direct": 2, "pathtracer": 3, "pathtracer2": 4}


class OrcScene(C.Structure):
    _fields_ = [("config", _ffi.Config),
                ("tris", C.c_void_p), ("n_tris", C.c_uint64),
                ("spheres", C.c_void_p), ("n_spheres", C.c_uint64),
                ("plights", C.c_void_p), ("n_plights", C.c_uint64),
                ("dlights", C.c_void_p), ("n_dlights", C.c_uint64),
                ("qlights", C.c_void_p), ("n_qlights", C.c_uint64),
                ("light_samples", C.c_uint32), ("light_stratify", C.c_uint32),
                ("next_event_est", C.c_uint32), ("sampling_method", C.c_uint32),
                ("samples_per_pixel", C.c_int32), ("integrator", C.c_int32),
                ("background", C.c_float * 3), ("width", C.c_uint32), ("height", C.c_uint32)]


class OrcOptions(C.Structure):
    _fields_ = [("semantics", C.c_int32), ("seeding", C.c_int32), ("bounce_cap", C.c_uint32),
                ("cull_zero", C.c_int32), ("use_bvh", C.c_int32), ("threads", C.c_int32),
                ("frame_id", C.c_int32), ("spp_begin", C.c_int32), ("spp_end", C.c_int32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32), ("stride", C.c_uint32)]


class OrcStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("seconds", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        l = C.CDLL(path)
        l.orc_primary_hits.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcOptions), C.c_void_p]
        l.orc_render.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcOptions), C.c_void_p, C.c_void_p, C.POINTER(OrcStats)]
        l.orc_tea16.argtypes = [C.c_uint32, C.c_uint32]
        l.orc_tea16.restype = C.c_uint32
        l.orc_lcg.argtypes = [C.POINTER(C.c_uint32)]
        l.orc_lcg.restype = C.c_uint32
        l.orc_rnd.argtypes = [C.POINTER(C.c_uint32)]
        l.orc_rnd.restype = C.c_float
        l.orc_to_rgba8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        l.orc_to_rgba8.restype = None
        l.orc_mat4_inverse.argtypes = [C.c_void_p, C.c_void_p]
        l.orc_mat4_inverse.restype = None
        l.orc_last_error.restype = C.c_char_p
        l.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        l.orc_refract.restype = C.c_int
        l.orc_fresnel_schlick.argtypes = [C.c_float]
        l.orc_fresnel_schlick.restype = C.c_float
        l.orc_phase_hg.argtypes = [C.c_float, C.c_float]
        l.orc_phase_hg.restype = C.c_float
        l.orc_probe_segments.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcOptions), C.c_uint32, C.c_void_p, C.c_void_p]
        l.orc_eval_shading.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib = l
    return _lib


def make_scene(scene) -> OrcScene:
    """OrcScene borrowing the arrays of a pathtracer_amd.Scene (keep `scene` alive)."""
    s = OrcScene()
    s.config = scene.config
    for name, arr in (("tris", scene.triangles), ("spheres", scene.spheres), ("plights", scene.plights),
                      ("dlights", scene.dlights), ("qlights", scene.qlights)):
        setattr(s, name, C.cast(arr, C.c_void_p))
        setattr(s, "n_" + name, len(arr))
    s.light_samples, s.light_stratify = scene.light_samples, scene.light_stratify
    s.next_event_est, s.sampling_method = scene.config.next_event_est, scene.sampling_method
    s.samples_per_pixel = scene.samples_per_pixel
    if scene.integratorName not in INTEGRATORS:
        raise RuntimeError("Doesn't support the integrator " + scene.integratorName)
    s.integrator = INTEGRATORS[scene.integratorName]
    s.background[:] = [0.0, 0.0, 0.0]
    s.width, s.height = scene.width, scene.height
    return s


def make_options(semantics=0, seeding=0, bounce_cap=1024, cull_zero=1, use_bvh=0, threads=8, frame_id=1,
                 spp_begin=0, spp_end=0, window=None, stride=1) -> OrcOptions:
    o = OrcOptions(semantics, seeding, bounce_cap, cull_zero, use_bvh, threads, frame_id, spp_begin, spp_end)
    if window:
        o.x0, o.y0, o.x1, o.y1 = window
    o.stride = stride
    return o


HIT_DTYPE = np.dtype([("prim", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])


def primary_hits(scene, **opts) -> np.ndarray:
    s, o = make_scene(scene), make_options(**opts)
    out = np.zeros((scene.height, scene.width), dtype=HIT_DTYPE)
    out["prim"] = -2  # -2 = not computed (outside window / stride)
    rc = lib().orc_primary_hits(C.byref(s), C.byref(o), out.ctypes.data)
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return out


def render(scene, **opts):
    """-> (sum, result, stats): float32 (H, W, 3) arrays, row 0 at the bottom."""
    s, o = make_scene(scene), make_options(**opts)
    sums = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    res = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    st = OrcStats()
    rc = lib().orc_render(C.byref(s), C.byref(o), sums.ctypes.data, res.ctypes.data, C.byref(st))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return sums, res, st


def to_rgba8(result: np.ndarray) -> np.ndarray:
    h, w, _ = result.shape
    out = np.empty((h, w, 4), dtype=np.uint8)
    res = np.ascontiguousarray(result, dtype=np.float32)
    lib().orc_to_rgba8(res.ctypes.data, w, h, out.ctypes.data)
    return out


def refract(i, n, ior):
    """(ok, r) of the restated OptiX `refract(r, i, n, ior)`."""
    i = np.ascontiguousarray(i, dtype=np.float32)
    n = np.ascontiguousarray(n, dtype=np.float32)
    r = np.zeros(3, dtype=np.float32)
    ok = lib().orc_refract(i.ctypes.data, n.ctypes.data, ior, r.ctypes.data)
    return bool(ok), r


def probe_segments(scene, probes: np.ndarray, **opts) -> np.ndarray:
    """One closest-hit program invocation per caller-made segment (dtype _ffi.PROBE_IN -> _ffi.PROBE_OUT)."""
    s, o = make_scene(scene), make_options(**opts)
    probes = np.ascontiguousarray(probes, dtype=_ffi.PROBE_IN)
    out = np.zeros(len(probes), dtype=_ffi.PROBE_OUT)
    rc = lib().orc_probe_segments(C.byref(s), C.byref(o), len(probes), probes.ctypes.data, out.ctypes.data)
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return out


def eval_shading(fn: int, inputs: np.ndarray) -> np.ndarray:
    inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 16)
    out = np.zeros((len(inputs), 4), dtype=np.float32)
    if lib().orc_eval_shading(fn, len(inputs), inputs.ctypes.data, out.ctypes.data):
        raise RuntimeError("orc_eval_shading: bad argument")
    return out
