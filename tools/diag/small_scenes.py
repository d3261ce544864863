"""Frame time of the small shipped scenes at 1920x1080 x 16 spp (the library chosen by PTCORE_LIB, see tools/ab.sh).
usage: python tools/diag/small_scenes.py [scene ...]   (GPU box)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from pathtracer_amd import Renderer, SceneLoader, override_scene_text, _ffi

names = sys.argv[1:] or ["scene1/scene2.test", "scene1/scene3.test", "scene1/scene6.test", "scene2/cornell.test", "scene4/ggx.test", "scene4/mis.test"]
for name in names:
    text = override_scene_text(open(os.path.join(ROOT, "scenes", name)).read(), size="1920 1080", spp="16")
    scene = SceneLoader().loadText(text)
    r = Renderer(scene, options={_ffi.OPT_SEEDING: 1})
    r.render_range(0, 16)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r.render_range(0, 16)
    r.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = r.stats()
    print(f"{name}: {dt * 1e3:.2f} ms per frame, {st.bvh_nodes} nodes", flush=True)
    r.close()
