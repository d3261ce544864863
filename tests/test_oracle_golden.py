"""The CPU oracle against everything the reference itself recorded for this path.

The reference ships no tests or golden vectors; what it does hold are rendered
images.  images/sphere.png (scene 2/sphere.test, `direct`, 25 stratified light
samples) was produced by code that matches the committed integrator, and the oracle
reproduces it byte for byte on >99.8% of the pixels - the per-pixel random numbers,
camera rays, triangle / sphere intersection, shadow rays, Phong term, area-light
geometry term and the float->byte conversion all line up.  images/cornellRR.png and
images/ggx.png match the `intended` semantics (helpers taking their outputs by
reference) statistically; they pin the path tracer's estimator, not its noise.
(The copies under tests/golden/ref_*.png are the reference's own output files.)
"""
import os

import numpy as np
import pytest

from pathtracer_amd import SceneLoader, override_scene_text
from oracle import oracle_py as orc
from util import ROOT, block_mean, read_png, scene_text

GOLD = os.path.join(ROOT, "tests", "golden")


def py_tea16(v0, v1):
    s0 = 0
    M = 0xFFFFFFFF
    for _ in range(16):
        s0 = (s0 + 0x9E3779B9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) & M ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xC8013EA4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) & M ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7E95761E) & M))) & M
    return v0


def test_rng_known_answers():
    import ctypes as C
    lib = orc.lib()
    for v0, v1 in [(0, 0), (1, 0), (0, 1), (12345, 678), (0xFFFFFFFF, 0xFFFFFFFF), (307199, 4095)]:
        assert lib.orc_tea16(v0, v1) == py_tea16(v0, v1)
    # tea<16>(0,0) of the OptiX SDK samples' random.h
    assert py_tea16(0, 0) == lib.orc_tea16(0, 0)
    st = C.c_uint32(42)
    x = 42
    for _ in range(100):
        x = (1664525 * x + 1013904223) & 0xFFFFFFFF
        assert lib.orc_lcg(C.byref(st)) == (x & 0x00FFFFFF)
        assert st.value == x
    st = C.c_uint32(7)
    f = lib.orc_rnd(C.byref(st))
    assert 0.0 <= f < 1.0 and f == np.float32((st.value & 0xFFFFFF)) / np.float32(16777216.0)


def test_matrix_inverse():
    rng = np.random.default_rng(1)
    for _ in range(20):
        m = rng.normal(size=(4, 4)).astype(np.float32)
        out = np.empty(16, dtype=np.float32)
        orc.lib().orc_mat4_inverse(m.ctypes.data, out.ctypes.data)
        want = np.linalg.inv(m.astype(np.float64))
        assert np.allclose(out.reshape(4, 4), want, rtol=2e-3, atol=2e-4)


def test_sphere_png_is_reproduced():
    """Pins RNG, ray generation, both intersection programs, shadow rays, `direct`
    and Renderer::getResult against the reference's own output."""
    scene = SceneLoader().load(os.path.join(ROOT, "scenes", "scene2", "sphere.test"))
    _, res, st = orc.render(scene, threads=8)
    mine = orc.to_rgba8(res)[..., :3].astype(int)
    ref = read_png(os.path.join(GOLD, "ref_sphere.png"))[..., :3].astype(int)
    assert mine.shape == ref.shape == (480, 640, 3)
    d = np.abs(mine - ref).max(axis=2)
    assert (d == 0).mean() > 0.998      # byte-exact
    assert (d <= 1).mean() > 0.9999     # the rest: one grey level (fast-math build of the reference)
    assert st.paths == 640 * 480 and st.shadow_rays > 75 * 100000



def test_red_dragon_png_is_reproduced():
    """images/red_dragon.png is scene 1/scene7.test (the dragon alone, one point light, the default
    `raytracer` integrator: Blinn-Phong + mirror recursion, RayTracer.cu:34-105) rendered at 1280x1024
    instead of the file's `size 640 480` (:3).  The oracle, `literal` semantics and reference seeding,
    reproduces it byte for byte on 99.94 % of the pixels (99.96 % within one grey level; channel means
    19.527 / 7.124 / 7.124 against 19.531 / 7.126 / 7.126): pins closestHit, its shadow rays and the
    triangle program over the 100 000-triangle mesh against the reference's own output."""
    from util import dragon_variant
    scene = SceneLoader().loadText(dragon_variant("raytracer", size="1280 1024"))
    assert scene.integratorName == "raytracer" and len(scene.triangles) == 100000
    _, res, _ = orc.render(scene, threads=8, semantics=0, seeding=0, use_bvh=1)
    mine = orc.to_rgba8(res)[..., :3].astype(int)
    ref = read_png(os.path.join(GOLD, "ref_red_dragon.png"))[..., :3].astype(int)
    assert mine.shape == ref.shape == (1024, 1280, 3)
    d = np.abs(mine - ref).max(axis=2)
    assert (d == 0).mean() > 0.999
    assert (d <= 1).mean() > 0.9995
    assert np.allclose(mine.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=1e-3)


def test_cornellBRDF_image_holds_cosine_sampling_only_weakly():
    """images/cornellBRDF.png against scene 4/cornellCosine.test (cornellBRDF.test with `importancesampling
    cosine`, the one line that differs) at the file's own 480x480 x 64 spp, `intended` semantics.  The image
    was made by some other state of the code -- ceiling 15-30 % darker, the sphere's surroundings 20-50 %
    brighter than any of the committed estimators -- so it holds the path only loosely.  Measured: cosine
    sampling channel means +5.1 % / +5.9 % / +5.4 %, relative RMSE of 16x16-pixel block means 10.5 %;
    uniform hemisphere +3.1 % / 9.6 %; BRDF sampling (the file's own mode) +10.8 % / 14.2 %.  (At 240x240 x 16
    spp the cosine render's 8-bit means come out within 1.5 % of the image: gamma 1/2.2 of noisier pixels
    averages lower, which hides the offset.)  Bars: means within 7 %, blocks within 12 %, and closer than
    BRDF sampling is."""
    ref = read_png(os.path.join(GOLD, "ref_cornellBRDF.png"))[..., :3].astype(np.float64)
    S = 2
    err = {}
    for rel in ("scene4/cornellCosine.test", "scene4/cornellBRDF.test"):
        scene = SceneLoader().load(os.path.join(ROOT, "scenes", rel))
        assert (scene.width, scene.height, scene.samples_per_pixel) == (480, 480, 64)
        top_rows = (scene.height - 1 - np.arange(0, scene.height, S))[::-1]
        _, res, _ = orc.render(scene, threads=8, semantics=1, seeding=1, stride=S)
        mine_s = orc.to_rgba8(res)[top_rows][:, ::S, :3].astype(np.float64)
        ref_s = ref[top_rows][:, ::S]
        bm, br = block_mean(mine_s, 8), block_mean(ref_s, 8)
        err[rel] = (np.sqrt(((bm - br) ** 2).mean()) / br.mean(), np.abs(mine_s.mean(axis=(0, 1)) / ref_s.mean(axis=(0, 1)) - 1).max())
    cos, brdf = err["scene4/cornellCosine.test"], err["scene4/cornellBRDF.test"]
    assert cos[0] < 0.12 and cos[1] < 0.07, err
    assert brdf[0] > cos[0] + 0.02 and brdf[1] > cos[1] + 0.03, err


def _strided_blocks(img, stride, block):
    return block_mean(img[::stride, ::stride].astype(np.float64), block)


@pytest.mark.parametrize("name,rel,tol", [("cornellRR", "scene3/cornellRR.test", 0.04), ("ggx", "scene4/ggx.test", 0.02)])
def test_intended_semantics_match_reference_images(name, rel, tol):
    """Every other pixel in x and y at the scene's own 64 spp; 8x8 blocks of those
    (16x16 pixel areas) must agree with the reference image within `tol` relative
    RMSE.  Literal semantics are far off (next test).  Measured: cornellRR 3.69 % under reference seeding and
    3.68 % under counter seeding (two independent noise realisations of the same estimator against the image's
    own: the figure is the Monte-Carlo noise of 64-spp block means, not a bias -- channel means agree within
    0.3/255); it was 3.4 % while the oracle still culled dead paths under reference seeding, which gave the pixels
    other seeds, i.e. a third realisation.  The bar is the larger value + 0.3 points.  ggx: 1.04 % / 1.32 %, bar 2 %."""
    scene = SceneLoader().load(os.path.join(ROOT, "scenes", rel))
    _, res, _ = orc.render(scene, threads=8, semantics=1, stride=2)
    mine = orc.to_rgba8(res)[::-1][::2, ::2][::-1]  # rows were flipped by the conversion
    h = scene.height
    rows = np.arange(0, h, 2)           # rendered bottom-up rows
    top_rows = (h - 1 - rows)[::-1]     # the same rows in the top-down PNG
    ref = read_png(os.path.join(GOLD, f"ref_{name}.png"))[..., :3]
    ref_s = ref[top_rows][:, ::2].astype(np.float64)
    mine_s = orc.to_rgba8(res)[top_rows][:, ::2, :3].astype(np.float64)
    bm, br = block_mean(mine_s, 8), block_mean(ref_s, 8)
    relrmse = np.sqrt(((bm - br) ** 2).mean()) / br.mean()
    assert relrmse < tol, relrmse
    assert np.allclose(mine_s.mean(axis=(0, 1)), ref_s.mean(axis=(0, 1)), rtol=0.02)


def test_dragon_image_is_matched_by_intended_semantics():
    """images/dragon2.png is the reference's recorded render of scene 4/dragon.test as shipped (640x480,
    128 spp, GGX, BRDF sampling, NEE + roulette, maxdepth 5) -- the scene of the headline benchmark.
    Every fourth pixel in x and y at the scene's own 128 spp: 8x8 blocks of those agree with the image
    within 3 % relative RMSE and the channel means within 1 % under `intended` semantics (measured at
    stride 2: 1.7 %, means within 0.4 %); `literal` is 43 % off."""
    scene = SceneLoader().load(os.path.join(ROOT, "scenes", "scene4", "dragon.test"))
    assert (scene.width, scene.height, scene.samples_per_pixel) == (640, 480, 128)
    S = 4
    h = scene.height
    top_rows = (h - 1 - np.arange(0, h, S))[::-1]
    ref_s = read_png(os.path.join(GOLD, "ref_dragon2.png"))[..., :3][top_rows][:, ::S].astype(np.float64)
    br = block_mean(ref_s, 8)
    err = {}
    for sem, spp in ((1, 128), (0, 16)):
        scene.samples_per_pixel = spp
        _, res, _ = orc.render(scene, threads=8, semantics=sem, seeding=0, stride=S, use_bvh=1)
        mine_s = orc.to_rgba8(res)[top_rows][:, ::S, :3].astype(np.float64)
        err[sem] = np.sqrt(((block_mean(mine_s, 8) - br) ** 2).mean()) / br.mean()
        if sem == 1:
            assert np.allclose(mine_s.mean(axis=(0, 1)), ref_s.mean(axis=(0, 1)), rtol=0.01)
    assert err[1] < 0.03, err
    assert err[0] > 0.3, err


def test_dragon_refract_variant_is_the_shipped_scene():
    """tests/util.py::dragon_variant("refract") rebuilds scene 4/dragonRefract.test from dragon.test (same
    geometry; 3.7 MB not kept twice): where the reference tree is present, both parse to the same arrays."""
    from util import dragon_variant
    ref = "/root/reference/Scenes/scene 4/dragonRefract.test"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    a = SceneLoader().load(ref)
    b = SceneLoader().loadText(dragon_variant("refract"))
    import ctypes as C
    assert len(a.triangles) == len(b.triangles) == 100006
    assert C.string_at(a.triangles, C.sizeof(a.triangles)) == C.string_at(b.triangles, C.sizeof(b.triangles))
    assert C.string_at(a.qlights, C.sizeof(a.qlights)) == C.string_at(b.qlights, C.sizeof(b.qlights))
    assert bytes(a.config) == bytes(b.config) and a.integratorName == b.integratorName
    assert (a.samples_per_pixel, a.sampling_method, a.light_samples) == (b.samples_per_pixel, b.sampling_method, b.light_samples)
    mat = np.frombuffer(a.triangles, dtype=np.int32).reshape(-1, 34)[:, 12 + 16]  # mv.matType: byte 48 + 64
    assert (mat == 1).sum() == 100000 and (mat == 0).sum() == 6  # the dragon is glass; the 4 light and 2 floor triangles are not


def test_literal_semantics_do_not_match_the_readme_image():
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), spp="8"))
    _, res, _ = orc.render(scene, threads=8, semantics=0, stride=4, bounce_cap=32)
    h = scene.height
    top_rows = (h - 1 - np.arange(0, h, 4))[::-1]
    mine = orc.to_rgba8(res)[top_rows][:, ::4, :3].astype(np.float64)
    ref = read_png(os.path.join(GOLD, "ref_cornellRR.png"))[top_rows][:, ::4, :3].astype(np.float64)
    assert abs(mine.mean() - ref.mean()) / ref.mean() > 0.15


SMALL = [("scene1/scene1.test", {}), ("scene1/scene5.test", dict(size="160 120")), ("scene2/sphere.test", dict(size="160 120")),
         ("scene3/cornellRR.test", dict(size="96 96")), ("scene4/ggx.test", dict(size="160 35"))]


@pytest.mark.parametrize("rel,over", SMALL)
def test_oracle_bvh_equals_brute_force(rel, over):
    scene = SceneLoader().loadText(override_scene_text(scene_text(rel), **over))
    a = orc.primary_hits(scene, use_bvh=0)
    b = orc.primary_hits(scene, use_bvh=1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_oracle_bvh_equals_brute_force_dragon():
    scene = SceneLoader().load(os.path.join(ROOT, "scenes", "scene4", "dragon.test"))
    assert len(scene.triangles) == 100006 and len(scene.qlights) == 2
    a = orc.primary_hits(scene, use_bvh=0, stride=16)
    b = orc.primary_hits(scene, use_bvh=1, stride=16)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (a["prim"] >= 0).sum() > 100


def test_cull_of_zero_throughput_paths_preserves_the_image():
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="48 48", spp="8"))
    for sem in (0, 1):
        a, _, sa = orc.render(scene, semantics=sem, seeding=1, cull_zero=1, bounce_cap=24)
        b, _, sb = orc.render(scene, semantics=sem, seeding=1, cull_zero=0, bounce_cap=24)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert sa.extension_rays <= sb.extension_rays
    assert sa.extension_rays < sb.extension_rays  # intended semantics: roulette does kill paths


def test_reference_seeding_traces_dead_paths_to_their_end():
    """Reference seeding: the per-pixel segment counter `i` (PinholeCamera.cu:34,56) advances for every
    segment the reference traces, dead paths (throughput 0 after Russian roulette, RayTracer.cu:926-928)
    included, and seeds every later sample of the pixel -- so the zero-throughput cull must not take
    effect there: cull on/off give the same image AND the same ray counts, and more rays are traced than
    with counter seeding, where the cull is free to end dead paths."""
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="48 48", spp="6"))
    a, _, sa = orc.render(scene, semantics=1, seeding=0, cull_zero=1, bounce_cap=24)
    b, _, sb = orc.render(scene, semantics=1, seeding=0, cull_zero=0, bounce_cap=24)
    c, _, sc = orc.render(scene, semantics=1, seeding=1, cull_zero=1, bounce_cap=24)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa.extension_rays == sb.extension_rays and sa.shadow_rays == sb.shadow_rays
    assert sc.extension_rays < sa.extension_rays


def test_counter_seeding_equals_reference_seeding_for_one_sample():
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellNEE.test"), size="48 48", spp="1"))
    a, _, _ = orc.render(scene, semantics=1, seeding=0)
    b, _, _ = orc.render(scene, semantics=1, seeding=1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_counter_seeding_partitions():
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="32 32", spp="6"))
    whole, _, _ = orc.render(scene, semantics=1, seeding=1, bounce_cap=8)
    p0, _, _ = orc.render(scene, semantics=1, seeding=1, bounce_cap=8, spp_begin=0, spp_end=2)
    p1, _, _ = orc.render(scene, semantics=1, seeding=1, bounce_cap=8, spp_begin=2, spp_end=6)
    assert np.allclose(whole, p0 + p1, rtol=1e-5, atol=1e-6)


def test_mis_is_nan_as_committed():
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene4/cornellMIS.test"), size="32 32", spp="1"))
    s, _, _ = orc.render(scene, bounce_cap=2)
    assert np.isnan(s).mean() > 0.5


def test_analytic_direct_needs_lightsamples_zero():
    """analyticDirect's closed form sits behind `light_samples == 0` (RayTracer.cu:116);
    as shipped (default 1) the image is ambient + emission only."""
    t = override_scene_text(scene_text("scene2/analytic.test"), size="64 48")
    a, _, _ = orc.render(SceneLoader().loadText(t))
    b, _, _ = orc.render(SceneLoader().loadText(override_scene_text(t, lightsamples="0")))
    assert not np.array_equal(a, b)
    floor = (a == 0).all(axis=2)          # pixels that only see non-emissive surfaces
    assert floor.any() and (b[floor] != 0).any()


def _variant(rel, material, **over):
    """Scene text with every `materialtype` command replaced by `materialtype <material>`."""
    import re
    t = re.sub(r"(?m)^materialtype .*$", "materialtype " + material, scene_text(rel))
    return SceneLoader().loadText(override_scene_text(t, **over))


def test_refract_and_fresnel_known_answers():
    """Snell's law / Schlick's polynomial, the definitions the restated OptiX helpers follow."""
    ok, r = orc.refract([0, 0, -1], [0, 0, 1], 1.5)          # normal incidence: straight through
    assert ok and np.allclose(r, [0, 0, -1])
    s45 = np.float32(np.sqrt(0.5))
    ok, r = orc.refract([s45, 0, -s45], [0, 0, 1], 1.5)       # entering: sin(t) = sin(i) / 1.5
    assert ok and np.isclose(r[0], s45 / 1.5, atol=1e-6) and r[2] < 0 and np.isclose(np.linalg.norm(r), 1, atol=1e-6)
    ok, r = orc.refract([s45, 0, s45], [0, 0, 1], 1.5)        # leaving the denser side at 45 deg: total internal reflection
    assert not ok and not r.any()
    ok, r = orc.refract([0.1, 0, np.sqrt(1 - 0.01)], [0, 0, 1], 1.5)  # leaving, below the critical angle: sin(t) = 1.5 sin(i)
    assert ok and np.isclose(r[0], 0.15, atol=1e-6) and r[2] > 0
    f = orc.lib().orc_fresnel_schlick
    assert f(1.0) == 0.0 and f(0.0) == 1.0 and f(-3.0) == 1.0 and np.isclose(f(0.5), 0.5 ** 5)
    hg = orc.lib().orc_phase_hg
    assert np.isclose(hg(0.3, 0.0), 1 / (4 * np.pi))          # isotropic
    assert hg(-1.0, 0.7) > hg(1.0, 0.7)                       # the reference's sign convention: peak at wi.wo = -1


def test_glass_first_segment():
    """One segment per path (maxdepth 1): a glass hit shades like the same material without
    the glass flag, except that it draws one more random number before the light samples
    (RayTracer.cu:614) -- so only pixels whose camera ray hits glass may differ."""
    over = dict(size="48 48", spp="1", maxdepth="1")
    glass = _variant("scene4/cornellRefract.test", "glass", **over)
    plain = _variant("scene4/cornellRefract.test", "diffuse", **over)
    hits = orc.primary_hits(glass)["prim"].reshape(48, 48)
    on_sphere = hits >= len(glass.triangles)
    assert 50 < on_sphere.sum() < 48 * 48 / 2
    for sem in (0, 1):
        a, _, _ = orc.render(glass, semantics=sem, seeding=1)
        b, _, _ = orc.render(plain, semantics=sem, seeding=1)
        differs = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
        assert not differs[~on_sphere].any()
        assert differs[on_sphere].mean() > 0.5


def test_glass_continues_from_the_origin_with_unchanged_throughput():
    """As committed, a glass hit leaves the throughput alone and restarts the path at (0,0,0)
    (RayTracer.cu:517, 935-952).  In cornellRefract the point (0,0,0) lies in the floor plane,
    so the longer path can only add light, and it stays finite."""
    over = dict(size="48 48", spp="1")  # one sample: the pixel-centre ray the hit mask is made of
    glass = _variant("scene4/cornellRefract.test", "glass", **over)
    hits = orc.primary_hits(glass)["prim"].reshape(48, 48)
    on_sphere = hits >= len(glass.triangles)
    one, _, s1 = orc.render(_variant("scene4/cornellRefract.test", "glass", maxdepth="1", **over), semantics=1, seeding=1)
    many, _, s2 = orc.render(glass, semantics=1, seeding=1, bounce_cap=16)
    assert np.isfinite(many).all()
    assert s2.extension_rays > s1.extension_rays
    assert (many[on_sphere] >= one[on_sphere]).all()


def test_volume_hit_kills_the_throughput():
    """A volumetric surface keeps the incoming direction (RayTracer.cu:661); with it the
    cosine factor of every sampling method is 0, so nothing after the first hit contributes:
    pixels that see the volume first get the same value at any depth, and with one segment
    the whole image equals the plain-material one bit for bit."""
    over = dict(size="48 48", spp="1")  # one sample: the pixel-centre ray the hit mask is made of
    vol1 = _variant("scene4/cornellVolume.test", "volumetric", maxdepth="1", **over)
    plain1 = _variant("scene4/cornellVolume.test", "diffuse", maxdepth="1", **over)
    vol = _variant("scene4/cornellVolume.test", "volumetric", **over)
    hits = orc.primary_hits(vol)["prim"].reshape(48, 48)
    on_sphere = hits >= len(vol.triangles)
    assert on_sphere.sum() > 30
    for sem in (0, 1):
        a, _, _ = orc.render(vol1, semantics=sem, seeding=1)
        b, _, _ = orc.render(plain1, semantics=sem, seeding=1)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for cull in (0, 1):
            c, _, _ = orc.render(vol, semantics=sem, seeding=1, bounce_cap=6, cull_zero=cull)
            assert np.array_equal(a[on_sphere], c[on_sphere])
