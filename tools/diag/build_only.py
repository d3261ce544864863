#!/usr/bin/env python3
"""Builds the dragon's hierarchies a few times and prints the build times (for profiling the build alone:
rocprofv3 --kernel-trace --stats -- python3 tools/diag/build_only.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import Renderer, SceneLoader, _ffi  # noqa: E402

scene = SceneLoader().load(os.path.join(ROOT, "scenes", "scene4", "dragon.test"))
builder = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for k in range(4):
    t0 = time.perf_counter()
    r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder})
    st = r.stats()
    print(f"builder {builder}: build_ms {st.build_ms:.2f} host_ms {st.build_host_ms:.2f} nodes {st.bvh_nodes} depth {st.bvh_max_depth} (constructor {1e3 * (time.perf_counter() - t0):.1f} ms)", flush=True)
    r.close()
