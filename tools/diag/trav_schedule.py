"""Diagnostic: how the lanes of the deferred-test traversal kernel spend their loop iterations
(counting instantiation, dragon at 1080p, 8 spp)."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pathtracer_amd import Renderer, SceneLoader, _ffi

tune = int(sys.argv[1], 0) if len(sys.argv) > 1 else None
scene = SceneLoader().loadText(bench.scene_text_for("dragon", spp_override=8))
opts = {_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_COUNT_TRAVERSAL: 1}
if tune is not None:
    opts[_ffi.OPT_TUNE_TRAVERSAL] = tune
r = Renderer(scene, options=opts)
r.reset_stats()
r.render_range(0, 8)
r.synchronize()
st = r.stats()
c = np.zeros(24, dtype=np.uint64)
r._check(r._lib.ptc_debug_counters(r._ctx, c.ctypes.data, 24))
r.close()
for name, base, rays, nodes, prims in (("extension", 8, st.extension_rays, st.nodes_visited, st.prims_tested), ("shadow", 16, st.shadow_rays, st.shadow_nodes_visited, st.shadow_prims_tested)):
    it, owned, node, leaf, full, part, part_items = (float(x) for x in c[base:base + 7])
    print(f"{name}: rays {rays}, nodes/ray {nodes/rays:.2f}, prims/ray {prims/rays:.2f}")
    print(f"  wave iterations {it:.4g} = {it*64/rays:.2f} lane-iterations per ray; lanes owning a ray {owned/it/64:.3f}, visiting a node {node/it/64:.3f}, at a leaf {leaf/it/64:.3f}")
    print(f"  full batches per iteration {full/it:.3f}; partial batches per iteration {part/it:.4f} with {part_items/max(part,1):.1f} tests each; lanes busy in a test batch {(full*64+part_items)/max(full+part,1)/64:.3f}")
print(f"shadow rays found occluded: {float(c[7]) / max(st.shadow_rays, 1):.3f}")
