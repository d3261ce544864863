"""ctypes declarations of the C-ABI PODs and libraries (include/ptcore.h,
csrc/host/pthost.h).  Plumbing only: no arithmetic happens here."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "_lib")


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Float2(C.Structure):
    _pack_ = 8
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class MaterialValue(C.Structure):
    _fields_ = [("ambient", Float3), ("diffuse", Float3), ("specular", Float3), ("emission", Float3),
                ("shininess", C.c_float), ("roughness", C.c_float), ("brdf_type", C.c_int32),
                ("ior", C.c_float), ("matType", C.c_int32),
                ("sigma_a", C.c_float), ("sigma_s", C.c_float), ("sigma_t", C.c_float), ("g", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("v1", Float3), ("v2", Float3), ("v3", Float3), ("normal", Float3),
                ("mv", MaterialValue), ("objType", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("trans", C.c_float * 16), ("mv", MaterialValue), ("objType", C.c_int32)]


class PointLight(C.Structure):
    _fields_ = [("color", Float3), ("attenuation", Float3), ("location", Float3)]


class DirectionalLight(C.Structure):
    _fields_ = [("color", Float3), ("direction", Float3)]


class QuadLight(C.Structure):
    _fields_ = [("tri1", Triangle), ("tri2", Triangle), ("color", Float3)]


class Config(C.Structure):
    _fields_ = [("w", Float3), ("u", Float3), ("v", Float3), ("eye", Float3),
                ("hSize", Float2), ("tanHFov", Float2),
                ("gamma", C.c_float), ("maxDepth", C.c_uint32), ("next_event_est", C.c_uint32),
                ("epsilon", C.c_float), ("russian_roul", C.c_uint32), ("_pad", C.c_uint32)]


class HitRecord(C.Structure):
    _fields_ = [("prim", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("extend_launches", C.c_uint64), ("connect_launches", C.c_uint64), ("shade_launches", C.c_uint64),
                ("extend_ms", C.c_double), ("connect_ms", C.c_double), ("shade_ms", C.c_double), ("other_ms", C.c_double),
                ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("shadow_nodes_visited", C.c_uint64), ("shadow_prims_tested", C.c_uint64),
                ("bvh_nodes", C.c_uint32), ("bvh_max_depth", C.c_uint32), ("build_ms", C.c_double),
                ("path_state_bytes", C.c_uint64), ("batch_paths", C.c_uint64),
                ("surface_ms", C.c_double), ("gather_ms", C.c_double), ("surface_hits", C.c_uint64), ("build_host_ms", C.c_double),
                ("extend_timed_rays", C.c_uint64), ("extend_timed_launches", C.c_uint64), ("surface_timed_hits", C.c_uint64), ("surface_timed_launches", C.c_uint64)]


class SceneView(C.Structure):
    _fields_ = [("config", Config), ("width", C.c_uint32), ("height", C.c_uint32),
                ("light_samples", C.c_uint32), ("light_stratify", C.c_uint32),
                ("samples_per_pixel", C.c_uint32), ("sampling_method", C.c_uint32),
                ("integrator_name", C.c_char_p), ("output_filename", C.c_char_p),
                ("triangles", C.POINTER(Triangle)), ("n_triangles", C.c_uint64),
                ("spheres", C.POINTER(Sphere)), ("n_spheres", C.c_uint64),
                ("plights", C.POINTER(PointLight)), ("n_plights", C.c_uint64),
                ("dlights", C.POINTER(DirectionalLight)), ("n_dlights", C.c_uint64),
                ("qlights", C.POINTER(QuadLight)), ("n_qlights", C.c_uint64),
                ("vertices", C.POINTER(Float3)), ("n_vertices", C.c_uint64)]


assert C.sizeof(MaterialValue) == 84 and C.sizeof(Triangle) == 136 and C.sizeof(Sphere) == 152
assert C.sizeof(QuadLight) == 284 and C.sizeof(Config) == 88 and C.sizeof(HitRecord) == 16

# PtcSegmentProbe / PtcSegmentResult as numpy record types
import numpy as _np  # noqa: E402
PROBE_IN = _np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("thr", "<f4", 3), ("depth", "<u4"), ("seed", "<u4")])
PROBE_OUT = _np.dtype([("radiance", "<f4", 3), ("thr", "<f4", 3), ("o", "<f4", 3), ("d", "<f4", 3), ("depth", "<u4"), ("alive", "<i4"), ("prim", "<i4"), ("t", "<f4")])
assert PROBE_IN.itemsize == 44 and PROBE_OUT.itemsize == 64

# option ids (include/ptcore.h)
OPT_SEMANTICS, OPT_SEEDING, OPT_BOUNCE_CAP, OPT_CULL_ZERO = 1, 2, 3, 4
OPT_BATCH_PATHS, OPT_PROFILE, OPT_COUNT_TRAVERSAL, OPT_BRUTE_FORCE = 5, 6, 7, 8
OPT_TUNE_TRAVERSAL = 9
OPT_BVH_BUILDER = 10
OPT_GROUP_HITS = 11
OPT_OVERLAP_GATHER = 12
OPT_SORT_RAYS = 13
OPT_DENSE_SURFACE = 14
OPT_MARK_RAYS = 15

ERR_INVALID, ERR_INTEGRATOR, ERR_DEVICE, ERR_UNSUPPORTED, ERR_NOT_BUILT, ERR_HOST, ERR_NOMEM, ERR_BUSY = -1, -2, -3, -4, -5, -6, -7, -8

_host = None
_core = None


def host_lib():
    """libpthost.so: scene loader + PNG writer (pure host code)."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libpthost.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(path)
        lib.pth_scene_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.pth_scene_load_text.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.pth_scene_free.argtypes = [C.c_void_p]
        lib.pth_scene_free.restype = None
        lib.pth_scene_view.argtypes = [C.c_void_p, C.POINTER(SceneView)]
        lib.pth_last_error.restype = C.c_char_p
        lib.pth_result_to_rgba8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.pth_result_to_rgba8.restype = None
        lib.pth_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        _host = lib
    return _host


CORE_SYMBOLS = [
    "ptc_create", "ptc_destroy", "ptc_last_error", "ptc_set_config", "ptc_set_integrator", "ptc_set_params",
    "ptc_set_triangles", "ptc_set_spheres", "ptc_set_lights", "ptc_set_background", "ptc_build", "ptc_render",
    "ptc_render_range", "ptc_resolve", "ptc_primary_hits", "ptc_read_result", "ptc_read_sum",
    "ptc_bind_sum_buffer", "ptc_set_stream", "ptc_synchronize", "ptc_set_option", "ptc_get_stats", "ptc_reset_stats",
    "ptc_collapse_wide", "ptc_render_range_static", "ptc_probe_segments", "ptc_eval_shading", "ptc_debug_counters",
    "ptc_debug_hierarchy",
]


def core_lib():
    """libptcore.so: the HIP path-tracing core.  There is no CPU fallback: a
    missing library is an error."""
    global _core
    if _core is None:
        # PTCORE_LIB: another build of the same library (A/B measurements of compile-time knobs)
        path = os.environ.get("PTCORE_LIB") or os.path.join(LIB_DIR, "libptcore.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP extension has not been built "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        lib = C.CDLL(path)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
        lib.ptc_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.ptc_destroy.argtypes = [vp]
        lib.ptc_last_error.argtypes = [vp]
        lib.ptc_last_error.restype = C.c_char_p
        lib.ptc_set_config.argtypes = [vp, C.POINTER(Config)]
        lib.ptc_set_integrator.argtypes = [vp, C.c_char_p]
        lib.ptc_set_params.argtypes = [vp, u32, u32, u32, u32, i32]
        lib.ptc_set_triangles.argtypes = [vp, vp, C.c_size_t]
        lib.ptc_set_spheres.argtypes = [vp, vp, C.c_size_t]
        lib.ptc_set_lights.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t]
        lib.ptc_set_background.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        lib.ptc_build.argtypes = [vp]
        lib.ptc_render.argtypes = [vp, C.c_int, u32, u32]
        lib.ptc_render_range.argtypes = [vp, C.c_int, u32, u32, i32, i32, C.c_int]
        lib.ptc_render_range_static.argtypes = [vp, C.c_int, u32, u32, i32, i32, C.c_int, vp, C.c_size_t]
        lib.ptc_probe_segments.argtypes = [vp, u32, vp, vp]
        lib.ptc_eval_shading.argtypes = [vp, C.c_int, u32, vp, vp]
        lib.ptc_debug_counters.argtypes = [vp, vp, u32]
        lib.ptc_debug_hierarchy.argtypes = [vp, C.c_int, vp, u32, C.POINTER(u32), vp, u32, C.POINTER(u32)]
        lib.ptc_resolve.argtypes = [vp, C.c_int, i32]
        lib.ptc_primary_hits.argtypes = [vp, u32, u32, vp]
        lib.ptc_read_result.argtypes = [vp, vp]
        lib.ptc_read_sum.argtypes = [vp, vp]
        lib.ptc_bind_sum_buffer.argtypes = [vp, vp, C.c_size_t]
        lib.ptc_set_stream.argtypes = [vp, vp]
        lib.ptc_synchronize.argtypes = [vp]
        lib.ptc_set_option.argtypes = [vp, C.c_int, C.c_int64]
        lib.ptc_get_stats.argtypes = [vp, C.POINTER(Stats)]
        lib.ptc_reset_stats.argtypes = [vp]
        _core = lib
    return _core
