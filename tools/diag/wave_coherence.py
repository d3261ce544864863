#!/usr/bin/env python3
"""How many lanes of a wave visit the SAME node in a step of the deferred-test traversal kernel (counting
instantiation): the part of the work a descent shared by the whole wave could take over.
usage: wave_coherence.py [maxdepth]   (dragon at 1920x1080, 4 spp; maxdepth 1 = camera rays and their shadow rays only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text  # noqa: E402

depth = sys.argv[1] if len(sys.argv) > 1 else "5"
text = open(os.path.join(ROOT, "scenes", "scene4", "dragon.test")).read()
scene = SceneLoader().loadText(override_scene_text(text, size="1920 1080", spp="4", maxdepth=depth))
r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_COUNT_TRAVERSAL: 1})
r.reset_stats()
r.render_range(0, 4)
r.synchronize()
st = r.stats()
c = np.zeros(40, dtype=np.uint64)
r._check(r._lib.ptc_debug_counters(r._ctx, c.ctypes.data, 40))
r.close()
print(f"dragon 1920x1080 x 4 spp, maxdepth {depth}")
for name, base, sched, rays, nodes in (("extension", 24, 8, st.extension_rays, st.nodes_visited), ("shadow", 32, 16, st.shadow_rays, st.shadow_nodes_visited)):
    distinct, one_it, one_lanes, four_it, four_lanes, top = (float(x) for x in c[base:base + 6])
    iters, node_lanes = float(c[sched]), float(c[sched + 2])
    print(f"{name}: {rays} rays, {nodes / rays:.2f} node visits per ray; {iters:.4g} wave iterations, {node_lanes / iters:.1f} lanes visiting a node in one")
    print(f"  distinct nodes per iteration {distinct / iters:.2f} -> a wave-wide node step serves {node_lanes / distinct:.2f} lanes on average")
    print(f"  iterations with ONE node for all visiting lanes: {one_it / iters:.3f} of the iterations, {one_lanes / node_lanes:.3f} of the node visits")
    print(f"  iterations with at most four distinct nodes: {four_it / iters:.3f} of the iterations, {four_lanes / node_lanes:.3f} of the node visits")
    print(f"  lanes at the most shared node of an iteration: {top / node_lanes:.3f} of the node visits")
