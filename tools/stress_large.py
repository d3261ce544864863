#!/usr/bin/env python3
"""One-off scale check: N translated copies of the dragon in one scene (N x 100k triangles).
Prints build time, node count and throughput, and compares the primary hits of a small
image with brute force on the GPU.  usage: tools/stress_large.py [copies]"""
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text  # noqa: E402


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    text = open(os.path.join(ROOT, "scenes", "scene4", "dragon.test")).read()
    lines = text.split("\n")
    first_v = next(i for i, l in enumerate(lines) if l.startswith("vertex"))
    last_t = max(i for i, l in enumerate(lines) if l.startswith("tri "))
    head, geo, tail = lines[:first_v], lines[first_v:last_t + 1], lines[last_t + 1:]
    nverts = sum(1 for l in geo if l.startswith("vertex"))
    # the dragon's vertices / triangles sit inside one pushTransform block of `head`; repeat
    # the geometry with shifted indices under additional translations
    body = []
    for c in range(copies):
        dx, dz = 9.0 * (c % 4) - 13.0, -9.0 * (c // 4)
        body.append("pushTransform")
        body.append(f"translate {dx} 0 {dz}")
        for l in geo:
            if l.startswith("tri "):
                a, b, cc = (int(x) + c * nverts for x in l.split()[1:4])
                body.append(f"tri {a} {b} {cc}")
            else:
                body.append(l)
        body.append("popTransform")
    head = [re.sub(r"^maxverts .*", f"maxverts {nverts * copies}", l) for l in head]
    scene_text = override_scene_text("\n".join(head + body + tail), size="640 360", spp="8", camera="-8 10 30  -1 0 -8  0 1 0  50")
    t0 = time.time()
    scene = SceneLoader().loadText(scene_text)
    print(f"{len(scene.triangles)} triangles parsed in {time.time() - t0:.1f} s", flush=True)
    for builder in (1, 0):
        r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_BVH_BUILDER: builder})
        st = r.stats()
        r.run()
        r.reset_stats()
        t0 = time.time()
        r.currentFrame = 0
        r.run()
        r.synchronize()
        dt = time.time() - t0
        s2 = r.stats()
        print(f"builder {builder}: build {st.build_ms:.1f} ms, {st.bvh_nodes} nodes, depth {st.bvh_max_depth}; "
              f"{s2.extension_rays / dt / 1e6:.0f} Mrays/s ({s2.extension_rays} extension + {s2.shadow_rays} shadow rays in {dt * 1e3:.1f} ms)", flush=True)
        hits = r.primary_hits()
        r.close()
        if builder == 1:
            ref_hits = hits
        else:
            same = all(np.array_equal(hits[f].view(np.uint32), ref_hits[f].view(np.uint32)) for f in ("prim", "t", "u", "v"))
            print("LBVH and PLOC hit records identical:", same, flush=True)
    small = SceneLoader().loadText(override_scene_text(scene_text, size="96 54", spp="1"))
    a = Renderer(small)
    b = Renderer(small, options={_ffi.OPT_BRUTE_FORCE: 1})
    ha, hb = a.primary_hits(), b.primary_hits()
    a.close()
    b.close()
    print("BVH vs brute force (96x54):", all(np.array_equal(ha[f].view(np.uint32), hb[f].view(np.uint32)) for f in ("prim", "t", "u", "v")),
          f"{(ha['prim'] >= 0).mean():.2f} of the pixels hit", flush=True)


if __name__ == "__main__":
    main()
