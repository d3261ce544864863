"""Diagnostic: agreement of the HIP path with the CPU oracle under reference seeding as the number of
samples per pixel grows (a decision that flips in one sample shifts the seeds of all later ones)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text
from oracle import oracle_py as orc
from util import scene_text, close_fraction

for rel, size in [("scene4/ggx.test", "160 35"), ("scene3/cornellRR.test", "96 96")]:
    for sem in (0, 1):
        for spp in (1, 2, 4, 8):
            scene = SceneLoader().loadText(override_scene_text(scene_text(rel), size=size, spp=str(spp)))
            out = []
            for seeding in (0, 1):
                r = Renderer(scene, options={_ffi.OPT_SEMANTICS: sem, _ffi.OPT_SEEDING: seeding, _ffi.OPT_BOUNCE_CAP: 12})
                r.run(); g = r.sums(); st = r.stats(); r.close()
                w, _, ost = orc.render(scene, semantics=sem, seeding=seeding, bounce_cap=12)
                out.append((close_fraction(g, w), int(st.extension_rays) - int(ost.extension_rays), int(ost.extension_rays)))
            print(rel, "sem", sem, "spp", spp, "seeding0 frac/diff/rays", out[0], "seeding1", out[1], flush=True)
