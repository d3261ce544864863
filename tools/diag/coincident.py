import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from pathtracer_amd import Renderer, SceneLoader, _ffi
from test_gpu_edges import soup
import ctypes as C
for ncopy in (50, 500, 5000):
    for builder in (0, 1, 2, 3):
        scene = soup(n_tris=2000, n_spheres=0, seed=5)
        n = len(scene.triangles)
        raw = np.frombuffer(scene.triangles, dtype=np.uint8).reshape(n, C.sizeof(_ffi.Triangle)).copy()
        dup = np.tile(raw[:1], (ncopy, 1))
        out = np.concatenate([raw, dup])
        scene.triangles = (_ffi.Triangle * len(out)).from_buffer_copy(out.tobytes())
        try:
            r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder})
            st = r.stats()
            print(ncopy, builder, "ok nodes", st.bvh_nodes, "depth", st.bvh_max_depth, flush=True)
            r.close()
        except RuntimeError as e:
            print(ncopy, builder, "FAIL", e, flush=True)
