"""Build time and traversal rate against scene size: the dragon replicated k times on a grid (same camera).
usage: python tools/diag/build_scaling.py [copies ...]   (GPU box)"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text  # noqa: E402
from util import scene_text  # noqa: E402


def load(rel, **over):
    return SceneLoader().loadText(override_scene_text(scene_text(rel), **over))


def replicate(scene, k):
    n = len(scene.triangles)
    raw = np.frombuffer(scene.triangles, dtype=np.uint8).reshape(n, C.sizeof(_ffi.Triangle))
    f0 = raw.view(np.float32).reshape(n, -1)
    # the dragon's own triangles are replicated on a grid; the floor / walls / lights (huge) stay, once
    small = np.abs(f0[:, 0:9]).max(axis=1) < 5.0
    parts = [raw]
    side = int(np.ceil(k ** 0.5))
    for c in range(1, k):
        blk = raw[small].copy()
        f = blk.view(np.float32).reshape(len(blk), -1)
        for v in range(3):
            f[:, 3 * v + 0] += (c % side) * 0.7
            f[:, 3 * v + 2] -= (c // side) * 0.7
        parts.append(blk)
    out = np.concatenate(parts)
    arr = (_ffi.Triangle * len(out)).from_buffer_copy(out.tobytes())
    scene.triangles = arr
    return scene


for k in [int(a) for a in sys.argv[1:]] or [1, 4, 10]:
    for builder in (1, 3):
        scene = replicate(load("scene4/dragon.test", size="640 480", spp="4"), k)
        t0 = time.time()
        r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder})
        t1 = time.time()
        r.run()
        st = r.stats()
        print(f"copies {k:3d} tris {len(scene.triangles):8d} builder {builder}: build {st.build_ms:8.1f} ms (wall {1e3 * (t1 - t0):8.1f}), nodes {st.bvh_nodes}, "
              f"extend {st.extend_ms:7.2f} ms shadow {st.connect_ms:7.2f} ms for {st.extension_rays} + {st.shadow_rays} rays", flush=True)
        r.close()
