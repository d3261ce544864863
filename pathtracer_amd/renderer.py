"""Renderer: Python mirror of the reference's Renderer class
(OptiXRenderer/Renderer.h:26-87) on top of the C-ABI of libptcore.so.

Same life cycle: Renderer(scene) uploads the scene and builds the acceleration
structure, run() renders the single frame (numFrames = 1), getResult() returns
RGBA8 bytes, top row first.  Errors from the core become RuntimeError with the
core's message, e.g. "Doesn't support the integrator foo" (Renderer.cpp:155)."""
import ctypes as C

import numpy as np

from . import _ffi
from .scene import Scene


class Renderer:
    def __init__(self, scene: Scene, device: int = 0, options=None):
        self._lib = _ffi.core_lib()
        self._ctx = C.c_void_p()
        if self._lib.ptc_create(device, C.byref(self._ctx)) != 0:
            raise RuntimeError(self._lib.ptc_last_error(None).decode())
        self.scene = scene
        self.numFrames = 1
        self.currentFrame = 0
        for k, v in (options or {}).items():
            self.set_option(k, v)
        self.buildScene()

    # -- helpers ---------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.ptc_last_error(self._ctx).decode())

    def set_option(self, option: int, value: int):
        self._check(self._lib.ptc_set_option(self._ctx, option, value))

    def close(self):
        if self._ctx:
            self._lib.ptc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference surface -----------------------------------------------
    def buildScene(self):
        s, lib, ctx = self.scene, self._lib, self._ctx
        self.width, self.height = s.width, s.height
        self.outputFilename = s.outputFilename
        self.currentFrame = 0
        self._check(lib.ptc_set_config(ctx, C.byref(s.config)))
        self._check(lib.ptc_set_integrator(ctx, s.integratorName.encode()))
        self._check(lib.ptc_set_params(ctx, s.light_samples, s.light_stratify, s.config.next_event_est,
                                       s.sampling_method, s.samples_per_pixel))
        self._check(lib.ptc_set_triangles(ctx, C.cast(s.triangles, C.c_void_p), len(s.triangles)))
        self._check(lib.ptc_set_spheres(ctx, C.cast(s.spheres, C.c_void_p), len(s.spheres)))
        self._check(lib.ptc_set_lights(ctx, C.cast(s.plights, C.c_void_p), len(s.plights),
                                       C.cast(s.dlights, C.c_void_p), len(s.dlights),
                                       C.cast(s.qlights, C.c_void_p), len(s.qlights)))
        self._check(lib.ptc_set_background(ctx, 0.0, 0.0, 0.0))
        self._check(lib.ptc_build(ctx))

    def run(self, progressive: bool = True):
        while self.currentFrame != self.numFrames:
            self.currentFrame += 1
            self._check(self._lib.ptc_render(self._ctx, self.currentFrame, self.width, self.height))
            if progressive:
                break

    def getResult(self) -> bytes:
        res = self.result()
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        _ffi.host_lib().pth_result_to_rgba8(res.ctypes.data, self.width, self.height, out.ctypes.data)
        return out.tobytes()

    def getOutputFilename(self): return self.outputFilename
    def getWidth(self): return self.width
    def getHeight(self): return self.height
    def getNumFrames(self): return self.numFrames
    def getCurrentFrame(self): return self.currentFrame
    def getScene(self): return self.scene

    # -- build-only extras -----------------------------------------------
    def result(self) -> np.ndarray:
        """float32 (H, W, 3), row 0 at the bottom (the mapped resultBuffer)."""
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._check(self._lib.ptc_read_result(self._ctx, out.ctypes.data))
        return out

    def sums(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._check(self._lib.ptc_read_sum(self._ctx, out.ctypes.data))
        return out

    def render_range(self, spp_begin, spp_end, clear=True, frame_id=1):
        self._check(self._lib.ptc_render_range(self._ctx, frame_id, self.width, self.height, spp_begin, spp_end, int(clear)))

    def render_range_static(self, spp_begin, spp_end, sums_device_ptr, sums_bytes, clear=True, frame_id=1):
        """ptc_render_range_static: the same samples added to the caller's device buffer, with all
        work arrays in the library's fixed scratch array (no allocation during the call)."""
        self._check(self._lib.ptc_render_range_static(self._ctx, frame_id, self.width, self.height, spp_begin, spp_end, int(clear),
                                                      C.c_void_p(sums_device_ptr), sums_bytes))

    def resolve(self, total_spp, frame_id=1):
        self._check(self._lib.ptc_resolve(self._ctx, frame_id, total_spp))

    def synchronize(self):
        self._check(self._lib.ptc_synchronize(self._ctx))

    def bind_sum_buffer(self, device_ptr, nbytes):
        self._check(self._lib.ptc_bind_sum_buffer(self._ctx, C.c_void_p(device_ptr), nbytes))

    def set_stream(self, stream_handle):
        self._check(self._lib.ptc_set_stream(self._ctx, C.c_void_p(stream_handle)))

    def primary_hits(self) -> np.ndarray:
        rec = np.empty((self.height, self.width), dtype=np.dtype([("prim", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")]))
        self._check(self._lib.ptc_primary_hits(self._ctx, self.width, self.height, rec.ctypes.data))
        return rec

    def probe_segments(self, probes: np.ndarray) -> np.ndarray:
        """ptc_probe_segments: one extension + closest-hit program per caller-made segment."""
        probes = np.ascontiguousarray(probes, dtype=_ffi.PROBE_IN)
        out = np.zeros(len(probes), dtype=_ffi.PROBE_OUT)
        self._check(self._lib.ptc_probe_segments(self._ctx, len(probes), probes.ctypes.data, out.ctypes.data))
        return out

    def eval_shading(self, fn: int, inputs: np.ndarray) -> np.ndarray:
        """ptc_eval_shading: the device shading helpers on raw inputs, (n, 16) floats -> (n, 4)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 16)
        out = np.zeros((len(inputs), 4), dtype=np.float32)
        self._check(self._lib.ptc_eval_shading(self._ctx, fn, len(inputs), inputs.ctypes.data, out.ctypes.data))
        return out

    def hierarchy(self, which: int = 0):
        """ptc_debug_hierarchy: (two-child nodes, four-wide nodes) of the hierarchy the build left on the device, as
        structured arrays in the layouts ptc_collapse_wide documents (the wide array is empty when that form was not built)."""
        BIN = np.dtype([("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3), ("child", "<u4", 2), ("pad", "<u4", 2)])
        WIDE = np.dtype([("origin", "<f4", 3), ("scale", "<f4", 3), ("qlo", "u1", (3, 4)), ("qhi", "u1", (3, 4)), ("child", "<u4", 4)])
        nb, nw = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.ptc_debug_hierarchy(self._ctx, which, None, 0, C.byref(nb), None, 0, C.byref(nw)))
        b, w = np.zeros(nb.value, dtype=BIN), np.zeros(nw.value, dtype=WIDE)
        self._check(self._lib.ptc_debug_hierarchy(self._ctx, which, b.ctypes.data, len(b), C.byref(nb), w.ctypes.data, len(w), C.byref(nw)))
        return b, w

    def stats(self) -> _ffi.Stats:
        st = _ffi.Stats()
        self._check(self._lib.ptc_get_stats(self._ctx, C.byref(st)))
        return st

    def reset_stats(self):
        self._check(self._lib.ptc_reset_stats(self._ctx))


def resultToPNG(outputFilename: str, width: int, height: int, result: bytes):
    """utils.cpp:3-14."""
    lib = _ffi.host_lib()
    buf = (C.c_ubyte * len(result)).from_buffer_copy(result)
    if lib.pth_write_png(outputFilename.encode(), buf, width, height) != 0:
        raise RuntimeError(lib.pth_last_error().decode())
    print(f"{outputFilename} is saved.")
