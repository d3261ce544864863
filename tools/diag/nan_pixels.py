#!/usr/bin/env python3
"""Pixels of the dragon (scene 4, GGX, `intended` semantics) whose radiance sum is NaN -- the reference's
acosf(dot(h, n)) with the product an ulp above 1 -- for the library in use (PTCORE_LIB selects a variant).
usage: nan_pixels.py [W H spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text  # noqa: E402

w, h, spp = (sys.argv[1:4] + ["1920", "1080", "64"][len(sys.argv) - 1:])[:3]
text = open(os.path.join(ROOT, "scenes", "scene4", "dragon.test")).read()
scene = SceneLoader().loadText(override_scene_text(text, size=f"{w} {h}", spp=str(spp)))
r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1})
r.run()
s = r.sums()
r.close()
nan = np.isnan(s).any(axis=2)
print(f"{os.environ.get('PTCORE_LIB', 'default library')}: {w}x{h} x {spp} spp: {int(nan.sum())} NaN pixels of {nan.size}; mean of the finite sums {np.nanmean(s, axis=(0, 1))}")
