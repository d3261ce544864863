"""Host-side scene objects: Python mirror of the reference's SceneLoader / Scene
(OptiXRenderer/SceneLoader.h:46, Scene.h:10-46).  Parsing is done by the C++
SceneLoader in libpthost.so; this module only holds the parsed arrays."""
import ctypes as C
import re

import numpy as np

from . import _ffi


class Scene:
    """Parsed scene: Config, image size, integrator name and the POD arrays in the
    wire format the core takes (kept as ctypes arrays)."""

    def __init__(self, view: _ffi.SceneView):
        self.config = _ffi.Config.from_buffer_copy(view.config)
        self.width, self.height = view.width, view.height
        self.light_samples = view.light_samples
        self.light_stratify = view.light_stratify
        self.samples_per_pixel = view.samples_per_pixel
        self.sampling_method = view.sampling_method
        self.integratorName = view.integrator_name.decode()
        self.outputFilename = view.output_filename.decode()

        def copy(ptr, n, typ):
            arr = (typ * n)()
            if n:
                C.memmove(arr, ptr, C.sizeof(typ) * n)
            return arr

        self.triangles = copy(view.triangles, view.n_triangles, _ffi.Triangle)
        self.spheres = copy(view.spheres, view.n_spheres, _ffi.Sphere)
        self.plights = copy(view.plights, view.n_plights, _ffi.PointLight)
        self.dlights = copy(view.dlights, view.n_dlights, _ffi.DirectionalLight)
        self.qlights = copy(view.qlights, view.n_qlights, _ffi.QuadLight)
        self.vertices = copy(view.vertices, view.n_vertices, _ffi.Float3)

    def triangles_np(self):
        """Triangles as a (T, 34) float32 view (136-byte records)."""
        return np.frombuffer(self.triangles, dtype=np.float32).reshape(len(self.triangles), 34)


class SceneLoader:
    """load(filename) -> Scene; raises RuntimeError("Unable to open scene file ...")
    like the reference (SceneLoader.cpp:40-44)."""

    def load(self, sceneFilename: str) -> Scene:
        lib = _ffi.host_lib()
        h = C.c_void_p()
        if lib.pth_scene_load(sceneFilename.encode(), C.byref(h)) != 0:
            raise RuntimeError(lib.pth_last_error().decode())
        return self._finish(lib, h)

    def loadText(self, text: str) -> Scene:
        lib = _ffi.host_lib()
        h = C.c_void_p()
        if lib.pth_scene_load_text(text.encode(), C.byref(h)) != 0:
            raise RuntimeError(lib.pth_last_error().decode())
        return self._finish(lib, h)

    @staticmethod
    def _finish(lib, h):
        try:
            view = _ffi.SceneView()
            if lib.pth_scene_view(h, C.byref(view)) != 0:
                raise RuntimeError(lib.pth_last_error().decode())
            return Scene(view)
        finally:
            lib.pth_scene_free(h)


def override_scene_text(text: str, **cmds) -> str:
    """Return scene text with single-argument-list commands replaced (or added in
    front when absent), e.g. override_scene_text(t, size="512 512", spp="1024").
    `size` must stay ahead of `camera`, so replacements are made in place."""
    out = text
    for cmd, args in cmds.items():
        pat = re.compile(r"^%s[ \t].*$" % re.escape(cmd), re.M)
        line = f"{cmd} {args}"
        if pat.search(out):
            out = pat.sub(line, out, count=1)
        else:
            out = line + "\n" + out
    return out
