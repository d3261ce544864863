import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text
from util import scene_text, close_fraction
from oracle import oracle_py as orc
rel = sys.argv[1]
over = dict(kv.split("=", 1) for kv in sys.argv[2:] if not kv.startswith("--"))
sem = int(over.pop("sem", 0)); seed = int(over.pop("seed", 0)); cap = int(over.pop("cap", 16)); bvh = int(over.pop("bvh", 0))
scene = SceneLoader().loadText(override_scene_text(scene_text(rel), **over))
r = Renderer(scene, options={_ffi.OPT_SEMANTICS: sem, _ffi.OPT_SEEDING: seed, _ffi.OPT_BOUNCE_CAP: cap})
r.run()
g = r.sums(); st = r.stats()
w, _, ost = orc.render(scene, semantics=sem, seeding=seed, bounce_cap=cap, use_bvh=bvh)
print("paths", st.paths, ost.paths, "ext", st.extension_rays, ost.extension_rays, "shadow", st.shadow_rays, ost.shadow_rays)
print("close frac", close_fraction(g, w), "nan g/w", np.isnan(g).sum(), np.isnan(w).sum(), "inf", np.isinf(g).sum(), np.isinf(w).sum())
d = np.abs(g - w); d[np.isnan(d)] = 0
idx = np.argsort(d.max(axis=2).ravel())[::-1][:12]
for i in idx:
    y, x = divmod(int(i), scene.width)
    print(y, x, g[y, x], w[y, x])
