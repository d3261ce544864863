#!/usr/bin/env python3
"""Generates tests/golden/fixtures.npz from the CPU oracle (run from the repository root:
python tests/golden/make_fixtures.py).  The reference itself cannot be built in this
environment (its programs need the OptiX SDK), so these vectors pin the oracle -- which is
pinned in turn by the reference's recorded images, tests/test_oracle_golden.py -- against
later changes, and give the GPU tests fixed expected values: RNG known answers, primary-hit
records (primitive, t, u, v), small radiance sums per integrator / semantics / seeding and
the NaN mask of `nexteventestimation mis`.  Data only: inputs are the scene files under
scenes/ plus the overrides listed in CASES."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle_py as orc  # noqa: E402
from pathtracer_amd import SceneLoader, override_scene_text  # noqa: E402
from util import scene_text  # noqa: E402

# name -> (scene file, overrides, oracle options for the render; None = hits only)
CASES = {
    "scene1": ("scene1/scene1.test", dict(size="64 48"), dict()),
    "scene6": ("scene1/scene6.test", dict(size="64 48"), dict()),
    "specular": ("scene1/scene4-specular.test", dict(size="64 48"), dict()),
    "analytic0": ("scene2/analytic.test", dict(size="64 48", lightsamples="0"), dict()),
    "sphere": ("scene2/sphere.test", dict(size="64 48"), dict()),
    "direct9": ("scene2/direct9.test", dict(size="64 48"), dict()),
    "cornellNEE_lit_ref": ("scene3/cornellNEE.test", dict(size="32 32", spp="3"), dict(semantics=0, seeding=0, bounce_cap=12)),
    "cornellRR_int_ctr": ("scene3/cornellRR.test", dict(size="32 32", spp="3"), dict(semantics=1, seeding=1, bounce_cap=12)),
    "cornellRR_int_ref": ("scene3/cornellRR.test", dict(size="32 32", spp="3"), dict(semantics=1, seeding=0, bounce_cap=12)),
    "cornellCosine_int_ctr": ("scene4/cornellCosine.test", dict(size="32 32", spp="3"), dict(semantics=1, seeding=1, bounce_cap=12)),
    "cornellBRDF_int_ctr": ("scene4/cornellBRDF.test", dict(size="32 32", spp="3"), dict(semantics=1, seeding=1, bounce_cap=12)),
    "ggx_int_ctr": ("scene4/ggx.test", dict(size="80 18", spp="3"), dict(semantics=1, seeding=1, bounce_cap=12)),
    "mis_lit_ref": ("scene4/cornellMIS.test", dict(size="32 32", spp="2"), dict(semantics=0, seeding=0, bounce_cap=4)),
    "pathtracer2": ("scene3/cornellNEE.test", dict(size="32 32", spp="3", integrator="pathtracer2"), dict(semantics=0, seeding=1)),
    "refract_int_ctr": ("scene4/cornellRefract.test", dict(size="32 32", spp="3"), dict(semantics=1, seeding=1, bounce_cap=12)),
    "volume_int_ctr": ("scene4/cornellVolume.test", dict(size="32 32", spp="3"), dict(semantics=1, seeding=1, bounce_cap=12)),
    "dragon": ("scene4/dragon.test", dict(size="64 48", spp="1"), None),
}


def load(name):
    rel, over, opts = CASES[name]
    return SceneLoader().loadText(override_scene_text(scene_text(rel), **over)), opts


def rng_vectors():
    lib = orc.lib()
    pairs = np.array([(0, 0), (1, 0), (0, 1), (12345, 678), (0xFFFFFFFF, 0xFFFFFFFF), (307199, 4095), (640 * 480 - 1, (255 << 12) | 7)], dtype=np.uint32)
    tea = np.array([lib.orc_tea16(int(a), int(b)) for a, b in pairs], dtype=np.uint32)
    st = C.c_uint32(int(tea[3]))
    rnd = np.array([lib.orc_rnd(C.byref(st)) for _ in range(16)], dtype=np.float32)
    return pairs, tea, rnd, np.uint32(st.value)


def main():
    out = {}
    out["tea_in"], out["tea_out"], out["rnd_seq"], out["rnd_state"] = rng_vectors()
    for name in CASES:
        scene, opts = load(name)
        out[f"hits_{name}"] = orc.primary_hits(scene, use_bvh=1 if name == "dragon" else 0)
        if opts is not None:
            sums, _, st = orc.render(scene, **opts)
            out[f"sums_{name}"] = sums
            out[f"rays_{name}"] = np.array([st.paths, st.extension_rays, st.shadow_rays], dtype=np.uint64)
    path = os.path.join(ROOT, "tests", "golden", "fixtures.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
