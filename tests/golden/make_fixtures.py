#!/usr/bin/env python3
"""Generates tests/golden/fixtures.npz from the CPU oracle (run from the repository root:
python tests/golden/make_fixtures.py).  The reference itself cannot be built in this
environment (its programs need the OptiX SDK), so these vectors pin the oracle -- which is
pinned in turn by the reference's recorded images, tests/test_oracle_golden.py -- against
later changes, and give the GPU tests fixed expected values: RNG known answers, primary-hit
records (primitive, t, u, v), small radiance sums per integrator / semantics / seeding, the NaN
mask of `nexteventestimation mis`, the shading helpers (phongBRDF, ggxBRDF, ggxPDF, sampled directions)
on 512 random inputs each, and single-segment probes: (ray, throughput, depth, seed) -> the payload
after rtTrace + miss / closest-hit program, for every integrator, sampling method, BRDF and semantics.  Data only: inputs are the scene files under
scenes/ plus the overrides listed in CASES."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle_py as orc  # noqa: E402
from pathtracer_amd import SceneLoader, _ffi, override_scene_text  # noqa: E402
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


# ---- single-function and single-segment probes ---------------------------------------------
def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def hemisphere(rng, nrm, min_cos=0.02):
    """Unit vectors with dot(v, nrm) >= min_cos."""
    v = unit(rng, len(nrm))
    d = (v * nrm).sum(axis=1, keepdims=True)
    v = np.where(d < 0, v - 2 * d * nrm, v)
    d = (v * nrm).sum(axis=1, keepdims=True)
    v = np.where(d < min_cos, v + (2 * min_cos) * nrm, v)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def shading_inputs(fn, n=512):
    """Inputs of orc_eval_shading / ptc_eval_shading (16 floats each), function `fn`."""
    rng = np.random.default_rng(1000 + fn)
    a = np.zeros((n, 16), dtype=np.float32)
    if fn == 0:    # phongBRDF: wi, reflect, Kd, Ks, shininess
        a[:, 0:3], a[:, 3:6] = unit(rng, n), unit(rng, n)
        a[:, 6:12] = rng.random((n, 6))
        a[:, 12] = rng.choice([1.0, 20.0, 30.0, 100.0, 1000.0], size=n)
    elif fn in (1, 2):
        nrm = unit(rng, n)
        wi, wo = hemisphere(rng, nrm), hemisphere(rng, nrm)
        if fn == 1:  # ggxBRDF: wi, wo, n, Ks, roughness
            a[:, 0:3], a[:, 3:6], a[:, 6:9] = wi, wo, nrm
            a[:, 9:12] = rng.random((n, 3))
            a[:, 12] = rng.uniform(0.05, 0.9, n)
        else:        # ggxPDF: wi, n, h, t, roughness
            h = wi + wo
            a[:, 0:3], a[:, 3:6], a[:, 6:9] = wi, nrm, h / np.linalg.norm(h, axis=1, keepdims=True)
            a[:, 12] = rng.uniform(0.25, 1.0, n)
            a[:, 13] = rng.uniform(0.05, 0.9, n)
    elif fn == 3:  # Phong-sampled direction: w, zeta0..2, t, shininess, sampling method
        a[:, 0:3] = unit(rng, n)
        a[:, 3:6] = rng.integers(0, 1 << 24, (n, 3)) / np.float32(16777216.0)
        a[:, 6] = rng.random(n)
        a[:, 7] = rng.choice([1.0, 20.0, 30.0, 1000.0], size=n)
        a[:, 8] = rng.integers(0, 3, n)
    else:          # GGX-sampled direction: n, wo, zeta1, zeta2, roughness
        nrm = unit(rng, n)
        a[:, 0:3], a[:, 3:6] = nrm, hemisphere(rng, nrm)
        a[:, 6:8] = rng.integers(0, 1 << 24, (n, 2)) / np.float32(16777216.0)
        a[:, 8] = rng.uniform(0.05, 0.9, n)
    return a


# name -> (scene file, overrides, text edit, oracle options): one closest-hit program / sampling method / BRDF / semantics each
PROBE_CASES = {
    "raytracer": ("scene1/scene4-specular.test", {}, None, dict()),
    "analytic0": ("scene2/analytic.test", dict(lightsamples="0"), None, dict()),
    "direct": ("scene2/sphere.test", {}, None, dict()),
    "pathtracer2": ("scene3/cornellNEE.test", dict(integrator="pathtracer2"), None, dict()),
    "hemisphere_phong_lit": ("scene3/cornellRR.test", {}, None, dict(semantics=0)),
    "hemisphere_phong_int": ("scene3/cornellRR.test", {}, None, dict(semantics=1)),
    "no_nee_int": ("scene3/cornellSimple.test", {}, None, dict(semantics=1)),
    "cosine_phong_int": ("scene4/cornellCosine.test", {}, None, dict(semantics=1)),
    "brdf_phong_int": ("scene4/cornellBRDF.test", {}, None, dict(semantics=1)),
    "brdf_phong_lit": ("scene4/cornellBRDF.test", {}, None, dict(semantics=0)),
    "brdf_ggx_int": ("scene4/ggx.test", {}, None, dict(semantics=1)),
    "brdf_ggx_lit": ("scene4/ggx.test", {}, None, dict(semantics=0)),
    "cornell_ggx_sphere_int": ("scene4/cornellBRDF.test", {}, ("#sphere\n", "#sphere\nbrdf ggx\nroughness 0.25\n"), dict(semantics=1)),
    "mis_lit": ("scene4/cornellMIS.test", {}, None, dict(semantics=0)),
    "glass_int": ("scene4/cornellRefract.test", {}, None, dict(semantics=1)),
    "volume_int": ("scene4/cornellVolume.test", {}, None, dict(semantics=1)),
}
PROBES_PER_LEVEL = 160


def load_probe_case(name):
    rel, over, edit, opts = PROBE_CASES[name]
    text = scene_text(rel)
    if edit:
        assert edit[0] in text
        text = text.replace(edit[0], edit[1], 1)
    return SceneLoader().loadText(override_scene_text(text, **over)), dict(opts, seeding=1, bounce_cap=64)


def probe_inputs(name):
    """Two generations of segments: camera-like rays through the image (throughput 1, depth 0), then the
    segments the oracle continues them with (its origin / direction / throughput / depth), new seeds."""
    scene, opts = load_probe_case(name)
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31) if False else sum(map(ord, name)))
    n = PROBES_PER_LEVEL
    cf = scene.config
    px, py = rng.random(n) * scene.width, rng.random(n) * scene.height
    f32 = np.float32
    abx = f32(cf.tanHFov.x) * (px.astype(f32) - f32(cf.hSize.x)) / f32(cf.hSize.x)
    aby = f32(cf.tanHFov.y) * (py.astype(f32) - f32(cf.hSize.y)) / f32(cf.hSize.y)
    u, v, w = (np.array(x.tuple(), dtype=f32) for x in (cf.u, cf.v, cf.w))
    d = abx[:, None] * u + aby[:, None] * v - w
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    first = np.zeros(n, dtype=_ffi.PROBE_IN)
    first["o"] = np.array(cf.eye.tuple(), dtype=f32)
    first["d"] = d
    first["thr"] = 1.0
    first["seed"] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    res = orc.probe_segments(scene, first, **opts)
    alive = res["alive"] != 0
    second = np.zeros(int(alive.sum()), dtype=_ffi.PROBE_IN)
    for f in ("o", "d", "thr", "depth"):
        second[f] = res[f][alive]
    second["seed"] = rng.integers(0, 1 << 32, len(second), dtype=np.uint64).astype(np.uint32)
    return scene, opts, np.concatenate([first, second])


def main():
    out = {}
    for fn in range(5):
        a = shading_inputs(fn)
        out[f"shade_in_{fn}"] = a
        out[f"shade_out_{fn}"] = orc.eval_shading(fn, a)
    for name in PROBE_CASES:
        scene, opts, probes = probe_inputs(name)
        out[f"probe_in_{name}"] = probes
        out[f"probe_out_{name}"] = orc.probe_segments(scene, probes, **opts)
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
