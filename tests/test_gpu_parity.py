"""GPU parity: the HIP wavefront path (through the C ABI of libptcore.so) against
the CPU oracle on the same seeded inputs.

Bars:
  * primary hits (primitive id, t, u, v): bit-exact, BVH and brute force alike;
  * radiance images: every arithmetic step is the same IEEE fp32 sequence except
    the libm calls (ROCm ocml vs glibc: powf/acosf/... differ by an ulp or two), so
    pixels agree to ~1e-5 relative unless such an ulp flips a visibility / Russian
    roulette decision.  Tolerance: >= 99% of the pixels within 1e-3 + 1e-3*|ref|
    and relative RMSE of 8x8 block means <= 2%.
"""
import numpy as np
import pytest

from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text
from util import block_mean, close_fraction, scene_text

pytestmark = pytest.mark.gpu


def orc():
    from oracle import oracle_py
    return oracle_py


def load(rel, **over):
    return SceneLoader().loadText(override_scene_text(scene_text(rel), **over))


def check_hits(scene, brute=False):
    r = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: int(brute)})
    got = r.primary_hits()
    r.close()
    want = orc().primary_hits(scene)
    assert np.array_equal(got["prim"], want["prim"]), f"{(got['prim'] != want['prim']).sum()} primitive ids differ"
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f"{f} not bit-exact"


HIT_SCENES = [("scene1/scene1.test", {}), ("scene1/scene4-specular.test", {}), ("scene1/scene5.test", dict(size="160 120")),
              ("scene1/scene6.test", dict(size="160 120")), ("scene2/sphere.test", {}), ("scene2/cornell.test", {}),
              ("scene3/cornellRR.test", {}), ("scene4/ggx.test", dict(size="640 140")), ("scene4/mis.test", dict(size="384 256"))]


@pytest.mark.parametrize("rel,over", HIT_SCENES)
def test_primary_hits_bvh(rel, over):
    check_hits(load(rel, **over))


@pytest.mark.parametrize("rel,over", HIT_SCENES[:5])
def test_primary_hits_brute_force(rel, over):
    check_hits(load(rel, **over), brute=True)


def test_primary_hits_dragon():
    scene = load("scene4/dragon.test", size="320 240")
    r = Renderer(scene)
    got = r.primary_hits()
    st = r.stats()
    r.close()
    want = orc().primary_hits(scene, use_bvh=1)
    assert st.bvh_nodes > 1000
    assert np.array_equal(got["prim"], want["prim"]), f"{(got['prim'] != want['prim']).sum()} primitive ids differ"
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f"{f} not bit-exact"


def compare_render(scene, semantics, seeding, bounce_cap=16, min_close=0.99, max_block_rel=0.02, use_bvh=0):
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: semantics, _ffi.OPT_SEEDING: seeding, _ffi.OPT_BOUNCE_CAP: bounce_cap})
    r.run()
    got, gsum = r.result(), r.sums()
    st = r.stats()
    r.close()
    wsum, want, ost = orc().render(scene, semantics=semantics, seeding=seeding, bounce_cap=bounce_cap, use_bvh=use_bvh)
    assert st.paths == ost.paths
    frac = close_fraction(gsum, wsum)
    fin = np.isfinite(wsum) & np.isfinite(gsum)
    bm_g, bm_w = block_mean(np.where(fin, gsum, 0), 8), block_mean(np.where(fin, wsum, 0), 8)
    rel = float(np.sqrt(((bm_g - bm_w) ** 2).mean()) / max(np.abs(bm_w).mean(), 1e-6))
    assert frac >= min_close, f"only {frac:.4f} of the pixels agree with the oracle"
    assert rel <= max_block_rel, f"block relRMSE {rel:.4f}"
    # rays traced must match up to the few paths whose decisions flipped
    assert abs(int(st.extension_rays) - int(ost.extension_rays)) <= 0.01 * ost.extension_rays + 16
    return frac, rel


@pytest.mark.parametrize("rel,over", [
    ("scene1/scene1.test", dict(size="160 120")),
    ("scene1/scene4-specular.test", dict(size="160 120")),
    ("scene1/scene6.test", dict(size="160 120")),
    ("scene2/analytic.test", dict(size="160 120")),
    ("scene2/analytic.test", dict(size="160 120", lightsamples="0")),
    ("scene2/sphere.test", dict(size="160 120")),
    ("scene2/direct9.test", dict(size="160 120")),
    ("scene2/cornell.test", dict(size="120 120")),
])
def test_deterministic_integrators(rel, over):
    compare_render(load(rel, **over), semantics=0, seeding=0)


PT = [("scene3/cornellSimple.test", dict(size="96 96", spp="4")),
      ("scene3/cornellNEE.test", dict(size="96 96", spp="4")),
      ("scene3/cornellRR.test", dict(size="96 96", spp="4")),
      ("scene4/cornellCosine.test", dict(size="96 96", spp="4")),
      ("scene4/cornellBRDF.test", dict(size="96 96", spp="4")),
      ("scene4/ggx.test", dict(size="160 35", spp="4")),
      ("scene4/mis.test", dict(size="96 64", spp="2"))]


@pytest.mark.parametrize("rel,over", PT)
@pytest.mark.parametrize("semantics", [0, 1])
@pytest.mark.parametrize("seeding", [0, 1])
def test_pathtracer(rel, over, semantics, seeding):
    # Literal GGX reflects exactly about the normal, so dot(h, n) sits at 1 +- 1 ulp and
    # acosf() of it is 0 or NaN on a coin flip that depends on the last bit of cosf():
    # the NaN masks of the two libms differ on ~2% of the pixels there.
    min_close = 0.97 if (semantics == 0 and "ggx" in rel) else 0.99
    # Reference seeding on curved geometry: a path whose throughput Russian roulette has zeroed is still
    # traced to its end (its segments advance the pixel's seed counter, PinholeCamera.cu:56), bouncing
    # off the spheres up to the segment cap; every bounce off a sphere magnifies the ulp-level
    # differences between the two libms, so ~1.3 % of such paths end after a different number of
    # segments -- and from then on every later sample of that pixel draws other seeds than the
    # oracle's.  Measured agreement: 100 % at 1 spp, 98.6 % at 2, 95.5 % at 4 (counter seeding: 99.9 %;
    # flat-walled Cornell boxes: 100 % either way).  The block-mean bar is unchanged.
    if seeding == 0 and semantics == 1 and "ggx" in rel:
        min_close = 0.93
    compare_render(load(rel, **over), semantics=semantics, seeding=seeding, bounce_cap=12, min_close=min_close)


@pytest.mark.parametrize("n_materials", [3, 24, 25, 60])
def test_many_materials(n_materials):
    """The surface kernel reads a hit's material from a copy of the material table in LDS when the table has at
    most 24 records and from memory otherwise: spheres of n distinct materials over a floor, next-event estimation
    with one quad light, against the oracle on both sides of the limit."""
    rng = np.random.default_rng(n_materials)
    lines = ["size 96 72", "integrator pathtracer", "spp 4", "maxdepth 4", "nexteventestimation on", "camera 0 2 6 0 0.5 0 0 1 0 45",
             "quadLight -1 4 -1 0 0 2 2 0 0 12 12 12", "maxverts 4", "vertex -6 0 -6", "vertex 6 0 -6", "vertex 6 0 6", "vertex -6 0 6",
             "diffuse 0.6 0.6 0.6", "tri 0 2 1", "tri 0 3 2"]
    for k in range(n_materials - 2):  # the floor and the light's own material are the other two
        d, sp = rng.uniform(0.05, 0.8, 3), rng.uniform(0.0, 0.3, 3)
        x, z = rng.uniform(-3, 3, 2)
        lines += [f"diffuse {d[0]:.3f} {d[1]:.3f} {d[2]:.3f}", f"specular {sp[0]:.3f} {sp[1]:.3f} {sp[2]:.3f}", f"shininess {5 + 3 * k}",
                  "brdf ggx" if k % 3 == 0 else "brdf phong", f"roughness {0.1 + 0.01 * k:.2f}", f"sphere {x:.3f} 0.4 {z:.3f} 0.4"]
    scene = SceneLoader().loadText("\n".join(lines) + "\n")
    st = compare_render(scene, semantics=1, seeding=1, bounce_cap=4)
    assert st is not None


def test_pathtracer2():
    scene = load("scene3/cornellNEE.test", size="96 96", spp="4", integrator="pathtracer2")
    compare_render(scene, semantics=0, seeding=1)


def test_mis_nan_mask():
    """`nexteventestimation mis` yields NaN radiance wherever a non-light surface is shaded
    (RayTracer.cu:704-708,785-794): the NaN mask must equal the oracle's."""
    scene = load("scene4/cornellMIS.test", size="64 64", spp="2")
    r = Renderer(scene, options={_ffi.OPT_BOUNCE_CAP: 4})
    r.run()
    got = r.sums()
    r.close()
    want, _, _ = orc().render(scene, bounce_cap=4)
    assert np.isnan(want).any()
    assert np.array_equal(np.isnan(got), np.isnan(want))


def test_dragon_render_small():
    scene = load("scene4/dragon.test", size="160 120", spp="2")
    compare_render(scene, semantics=1, seeding=1, use_bvh=1, min_close=0.99, max_block_rel=0.02)


def test_sample_range_partition_is_exact():
    """Counter seeding: rendering [0,8) in one go equals [0,3)+[3,8) accumulated, bit for bit."""
    scene = load("scene3/cornellRR.test", size="64 64", spp="8")
    opts = {_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_BOUNCE_CAP: 8}
    r = Renderer(scene, options=opts)
    r.render_range(0, 8, clear=True)
    whole = r.sums()
    r.render_range(0, 3, clear=True)
    r.render_range(3, 8, clear=False)
    parts = r.sums()
    r.set_option(_ffi.OPT_BATCH_PATHS, 64 * 64 * 2)
    r.render_range(0, 8, clear=True)
    small_batches = r.sums()
    r.close()
    assert np.array_equal(whole.view(np.uint32), parts.view(np.uint32))
    assert np.array_equal(whole.view(np.uint32), small_batches.view(np.uint32))


def test_unknown_integrator_message():
    scene = load("scene2/sphere.test", integrator="photonmap")
    with pytest.raises(RuntimeError, match="Doesn't support the integrator photonmap"):
        Renderer(scene)


def material_variant(rel, material, head="", **over):
    """Scene with every `materialtype` command set to `material`; `head` is put before the
    first vertex so that the walls can be given a material type too."""
    import re
    t = re.sub(r"(?m)^materialtype .*$", "materialtype " + material, scene_text(rel))
    if head:
        t = t.replace("maxverts", head + "\nmaxverts", 1)
    return SceneLoader().loadText(override_scene_text(t, **over))


@pytest.mark.parametrize("semantics", [0, 1])
@pytest.mark.parametrize("seeding", [0, 1])
@pytest.mark.parametrize("rel,material,head", [
    ("scene4/cornellRefract.test", "glass", ""),               # glass spheres (ior 1.07 / 1.02), entered and left
    ("scene4/cornellRefract.test", "glass", "materialtype glass\nior 1.5"),  # glass triangles as well (dragonRefract's case)
    ("scene4/cornellVolume.test", "volumetric", ""),
    ("scene4/cornellVolume.test", "volumetric", "materialtype volumetric\nscattercoeff 0.5 2.0 0.3"),
])
def test_pathtracer_glass_and_volume(rel, material, head, semantics, seeding):
    """Glass and volumetric branches of pathTracer (RayTracer.cu:574-672, 935-957)."""
    scene = material_variant(rel, material, head, size="96 96", spp="4")
    # reference seeding + spheres: dead paths bounce on to the segment cap and a few per cent of them end
    # at a different segment than the oracle's, which shifts the seeds of the pixel's later samples
    # (see test_pathtracer); a volumetric hit zeroes the throughput, so these scenes are full of them
    compare_render(scene, semantics=semantics, seeding=seeding, bounce_cap=12, min_close=0.95 if seeding == 0 else 0.99)


def test_bvh_builders_agree():
    """LBVH, PLOC, the surface-area sweep in host code (2) and on the GPU (4), and the pair of trees of builder 3 (the
    GPU sweep for the closest-hit rays, PLOC for the any-hit rays), give the same hits and the same image, bit for bit.
    The GPU sweep makes the same decisions as the host sweep (same costs, same tie rules): its tree has the same number
    of nodes and costs the rays the same number of node visits and primitive tests (up to the order of the two leaves of
    a two-primitive node, which the host takes from the parent's axis)."""
    scene = load("scene4/dragon.test", size="160 120", spp="2")
    out = {}
    for builder in (0, 1, 2, 4, 3):
        r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder, _ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_COUNT_TRAVERSAL: 1})
        hits = r.primary_hits()
        r.reset_stats()
        r.run()
        st = r.stats()
        out[builder] = (hits, r.sums(), st.bvh_nodes, st.nodes_visited, st.prims_tested, st.build_host_ms)
        r.close()
    assert len({out[b][2] for b in (0, 1, 2)}) == 3  # really different trees
    assert out[3][2] == out[4][2]                    # builder 3 reports its closest-hit tree: builder 4's
    assert out[4][2] == out[2][2], (out[4][2], out[2][2])
    assert abs(out[4][3] - out[2][3]) <= 0.002 * out[2][3] and abs(out[4][4] - out[2][4]) <= 0.002 * out[2][4], (out[4][3:5], out[2][3:5])
    assert out[4][5] == 0.0 and out[1][5] == 0.0 and out[3][5] == 0.0 and out[2][5] > 0.0  # which builds ran host passes over the tree
    for b in (1, 2, 4, 3):
        for f in ("prim", "t", "u", "v"):
            assert np.array_equal(out[0][0][f].view(np.uint32), out[b][0][f].view(np.uint32)), (b, f)
        assert np.array_equal(out[0][1].view(np.uint32), out[b][1].view(np.uint32)), b


@pytest.mark.parametrize("tune", [0x0020, 0x1120, 0x2220, 0x4420, 0x4408, 0x4440, 0x1320, 0x6620, 0x6608, 0x6640])
@pytest.mark.parametrize("rel,over", [("scene4/dragon.test", dict(size="160 120", spp="2")),
                                      ("scene3/cornellRR.test", dict(size="96 96", spp="4")),
                                      ("scene2/sphere.test", dict(size="160 120"))])
def test_traversal_kernels_agree(rel, over, tune):
    """Every traversal kernel / schedule (static, refilling rounds, refilling steps, deferred
    primitive tests; several refill thresholds) finds the same hits: hit records and
    radiance sums are bit-identical to brute force over all primitives."""
    scene = load(rel, **over)
    opts = {_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_BOUNCE_CAP: 6}
    r = Renderer(scene, options={**opts, _ffi.OPT_TUNE_TRAVERSAL: tune})
    hits = r.primary_hits()
    r.run()
    sums = r.sums()
    r.close()
    key = (rel, tuple(sorted(over.items())))
    if key not in _BRUTE:
        b = Renderer(scene, options={**opts, _ffi.OPT_BRUTE_FORCE: 1})
        bh = b.primary_hits()
        b.run()
        _BRUTE[key] = (bh, b.sums())
        b.close()
    want_hits, want_sums = _BRUTE[key]
    for f in ("prim", "t", "u", "v"):
        assert np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)), f
    assert np.array_equal(sums.view(np.uint32), want_sums.view(np.uint32))


_BRUTE = {}


# ---- the HIP path against the reference's own recorded images ---------------------
def test_sphere_png_from_the_gpu():
    """scene 2/sphere.test as shipped (640x480, `direct`, 25 stratified light samples): the
    GPU result, through Renderer::getResult's float->byte conversion, against the image the
    reference recorded -- byte-exact on > 99.8 % of the pixels, the rest one grey level."""
    import os
    from util import ROOT, read_png
    scene = SceneLoader().load(os.path.join(ROOT, "scenes", "scene2", "sphere.test"))
    r = Renderer(scene)
    r.run()
    mine = np.frombuffer(bytes(r.getResult()), dtype=np.uint8).reshape(480, 640, 4)[..., :3].astype(int)
    r.close()
    ref = read_png(os.path.join(ROOT, "tests", "golden", "ref_sphere.png"))[..., :3].astype(int)
    d = np.abs(mine - ref).max(axis=2)
    assert (d == 0).mean() > 0.998
    assert (d <= 1).mean() > 0.9999


def test_red_dragon_png_from_the_gpu():
    """scene 1/scene7.test at the 1280x1024 of images/red_dragon.png (`raytracer`: Blinn-Phong, one point light,
    shadow rays, mirror recursion to maxdepth 5 over the 100 000-triangle dragon; RayTracer.cu:34-105): the GPU
    result through Renderer::getResult against the image the reference recorded -- byte-exact on > 99.9 % of the
    pixels (the oracle: 99.94 %), all but 0.05 % within one grey level."""
    import os
    from util import ROOT, dragon_variant, read_png
    scene = SceneLoader().loadText(dragon_variant("raytracer", size="1280 1024"))
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 0, _ffi.OPT_SEEDING: 0})
    r.run()
    mine = np.frombuffer(bytes(r.getResult()), dtype=np.uint8).reshape(1024, 1280, 4)[..., :3].astype(int)
    r.close()
    ref = read_png(os.path.join(ROOT, "tests", "golden", "ref_red_dragon.png"))[..., :3].astype(int)
    d = np.abs(mine - ref).max(axis=2)
    assert (d == 0).mean() > 0.999, (d == 0).mean()
    assert (d <= 1).mean() > 0.9995, (d <= 1).mean()
    assert np.allclose(mine.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=1e-3)


def test_cosine_sampling_at_full_size():
    """scene 4/cornellCosine.test as shipped (480x480 x 64 spp, `intended` semantics).  Against the oracle under
    counter seeding: per-pixel and block bars of every other render here.  Against images/cornellBRDF.png, the only
    recorded image of this box: the loose bars of tests/test_oracle_golden.py (the image comes from another state
    of the code; channel means within 7 %, 16x16-pixel block means within 12 % relative RMSE -- the oracle
    measures 5.9 % / 10.5 %)."""
    import os
    from util import ROOT, read_png
    scene = SceneLoader().load(os.path.join(ROOT, "scenes", "scene4", "cornellCosine.test"))
    assert (scene.width, scene.height, scene.samples_per_pixel, scene.sampling_method) == (480, 480, 64, 1)
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1})
    r.run()
    gsum = r.sums()
    mine = np.frombuffer(bytes(r.getResult()), dtype=np.uint8).reshape(480, 480, 4)[..., :3].astype(np.float64)
    r.close()
    ref = read_png(os.path.join(ROOT, "tests", "golden", "ref_cornellBRDF.png"))[..., :3].astype(np.float64)
    bm, br = block_mean(mine, 16), block_mean(ref, 16)
    relrmse = np.sqrt(((bm - br) ** 2).mean()) / br.mean()
    assert relrmse < 0.12, relrmse
    assert np.abs(mine.mean(axis=(0, 1)) / ref.mean(axis=(0, 1)) - 1).max() < 0.07
    # every fourth pixel in x and y through the oracle, same seeds
    S = 4
    wsum, _, _ = orc().render(scene, semantics=1, seeding=1, stride=S, threads=8)
    a, b = gsum[::S, ::S], wsum[::S, ::S]
    assert close_fraction(a, b) >= 0.99, close_fraction(a, b)
    bm_g, bm_w = block_mean(a, 8), block_mean(b, 8)
    assert np.sqrt(((bm_g - bm_w) ** 2).mean()) / np.abs(bm_w).mean() <= 0.02


@pytest.mark.parametrize("name,rel,tol", [("cornellRR", "scene3/cornellRR.test", 0.04), ("ggx", "scene4/ggx.test", 0.02)])
def test_converged_images_match_the_reference(name, rel, tol):
    """Full-size renders at the scene's own settings (`intended` semantics, reference
    seeding) against the reference's images: relative RMSE of 16x16-pixel block means
    <= tol, channel means within 2 %.  The figure is the Monte-Carlo noise of two independent 64-spp renders, not a
    bias: the oracle measures 3.69 % (reference seeding) / 3.68 % (counter seeding) on cornellRR against the same
    image and 1.04 % / 1.32 % on ggx (tests/test_oracle_golden.py); bars = those + 0.3 points / 2 %."""
    import os
    from util import ROOT, read_png
    scene = SceneLoader().load(os.path.join(ROOT, "scenes", rel))
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 0})
    r.run()
    mine = np.frombuffer(bytes(r.getResult()), dtype=np.uint8).reshape(scene.height, scene.width, 4)[..., :3].astype(np.float64)
    r.close()
    ref = read_png(os.path.join(ROOT, "tests", "golden", f"ref_{name}.png"))[..., :3].astype(np.float64)
    assert mine.shape == ref.shape
    bm, br = block_mean(mine, 16), block_mean(ref, 16)
    relrmse = np.sqrt(((bm - br) ** 2).mean()) / br.mean()
    assert relrmse < tol, relrmse
    assert np.allclose(mine.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=0.02)


@pytest.mark.parametrize("rel,over", [("scene4/dragon.test", dict(size="160 120", spp="2")), ("scene4/ggx.test", dict(size="160 35", spp="4")),
                                      ("scene3/cornellRR.test", dict(size="96 96", spp="4"))])
def test_hit_grouping_does_not_change_results(rel, over):
    """Grouping a bounce's hits by shading class only reorders work: bit-identical sums."""
    scene = load(rel, **over)
    out = []
    for group in (0, 1):
        r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_BOUNCE_CAP: 8, _ffi.OPT_GROUP_HITS: group})
        r.run()
        out.append(r.sums())
        r.close()
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


@pytest.mark.parametrize("rel,over", [("scene4/dragon.test", dict(size="320 240", spp="8")), ("scene3/cornellRR.test", dict(size="96 96", spp="16")),
                                      ("scene4/ggx.test", dict(size="320 70", spp="8")), ("scene4/cornellBRDF.test", dict(size="128 128", spp="16", maxdepth="6"))])
@pytest.mark.parametrize("seeding", [1, 0])
def test_skipping_the_hit_compaction_does_not_change_results(rel, over, seeding):
    """Bounces whose paths nearly all hit something (closed boxes; decided from the bounce before) go without the hit
    queue -- the surface kernel looks at every live path and ends the misses itself -- or with it: bit-identical sums
    and the same ray and hit counts; several frames in a row so that the camera rays' decision (taken from the previous
    batch) is exercised both ways."""
    scene = load(rel, **over)
    out = []
    for dense in (0, 1):
        r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: seeding, _ffi.OPT_BOUNCE_CAP: 8, _ffi.OPT_DENSE_SURFACE: dense})
        frames = []
        for _ in range(3):
            r.reset_stats()
            r.render_range(0, scene.samples_per_pixel, clear=True)
            st = r.stats()
            frames.append((r.sums().copy(), st.extension_rays, st.shadow_rays, st.surface_hits))
        r.close()
        for f in frames[1:]:
            assert np.array_equal(frames[0][0].view(np.uint32), f[0].view(np.uint32)) and frames[0][1:] == f[1:]
        out.append(frames[0])
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert out[0][1:] == out[1][1:]


@pytest.mark.parametrize("semantics,seeding", [(1, 1), (1, 0), (0, 1)])
@pytest.mark.parametrize("builder", [3, 1, 4])
def test_rays_marked_as_clear_of_the_large_triangles_find_the_same_hits(builder, semantics, seeding):
    """The dragon's floor is two triangles as wide as the scene, kept in a leaf of their own under the root.  The surface
    kernel marks the shadow rays and next segments that cannot hit them (they leave the floor or point away from it) and
    the traversal kernels start those below that leaf: the same sums bit for bit and the same ray counts as without the
    marks and as brute force over all primitives, with a quarter fewer triangle tests; one hierarchy or two."""
    scene = load("scene4/dragon.test", size="200 150", spp="4")
    out = {}
    for name, opts in (("brute", {_ffi.OPT_BRUTE_FORCE: 1}), ("plain", {_ffi.OPT_MARK_RAYS: 0}), ("marked", {_ffi.OPT_MARK_RAYS: 1})):
        r = Renderer(scene, options={_ffi.OPT_SEMANTICS: semantics, _ffi.OPT_SEEDING: seeding, _ffi.OPT_BOUNCE_CAP: 6, _ffi.OPT_BVH_BUILDER: builder,
                                     _ffi.OPT_COUNT_TRAVERSAL: 1, **opts})
        r.reset_stats()
        r.run()
        st = r.stats()
        out[name] = (r.sums().copy(), st.extension_rays, st.shadow_rays, st.prims_tested + st.shadow_prims_tested, st.nodes_visited + st.shadow_nodes_visited)
        r.close()
    for name in ("plain", "marked"):
        assert np.array_equal(out["brute"][0].view(np.uint32), out[name][0].view(np.uint32)), name
        assert out["brute"][1:3] == out[name][1:3], name
    assert out["marked"][3] < 0.9 * out["plain"][3], (out["marked"][3], out["plain"][3])  # fewer triangle tests ...
    assert out["marked"][4] <= 1.01 * out["plain"][4]  # ... and as many node visits (the closest distance lags the queued tests by a step or two, differently now)


@pytest.mark.parametrize("rel,over", [("scene4/dragon.test", dict(size="320 240", spp="8")), ("scene3/cornellRR.test", dict(size="96 96", spp="16")),
                                      ("scene2/sphere.test", dict(size="160 120"))])
def test_gather_on_second_stream_does_not_change_results(rel, over):
    """The radiance gather of a bounce beside the next bounce's traversal (second stream, two
    events per bounce) or everything on one stream: bit-identical sums, also when the same
    context renders several frames in a row."""
    scene = load(rel, **over)
    out = []
    for overlap in (0, 1):
        r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_BOUNCE_CAP: 8, _ffi.OPT_OVERLAP_GATHER: overlap})
        frames = []
        for _ in range(3):
            r.render_range(0, scene.samples_per_pixel, clear=True)
            frames.append(r.sums().copy())
        r.close()
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32)) and np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32))
        out.append(frames[0])
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))


@pytest.mark.parametrize("rel,over", [("scene4/dragon.test", dict(size="480 360", spp="4")), ("scene3/cornellRR.test", dict(size="192 192", spp="8")),
                                      ("scene4/ggx.test", dict(size="640 140", spp="4"))])
def test_ray_sorting_does_not_change_results(rel, over):
    """PTC_OPT_SORT_RAYS reorders the live paths of every bounce after the first (origin cell, direction
    octant); the packed state moves as a whole and radiance is addressed by path number: bit-identical sums
    and the same ray counts."""
    scene = load(rel, **over)
    out = []
    for sort in (0, 1):
        r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_BOUNCE_CAP: 8, _ffi.OPT_SORT_RAYS: sort})
        r.run()
        st = r.stats()
        out.append((r.sums(), st.extension_rays, st.shadow_rays))
        r.close()
    assert out[0][1] > 4 * 65536  # enough live paths for the sort to run
    assert out[0][1:] == out[1][1:]
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))


@pytest.mark.parametrize("kind,semantics", [("direct", 0), ("nee", 0), ("nee", 1), ("raytracer", 0)])
def test_other_dragon_scenes(kind, semantics):
    """scene 2/dragon.test (direct, 50 stratified shadow rays per path), scene 3/dragon.test
    (path tracer, Phong, NEE + roulette) and scene 1/scene7.test (ray tracer, point light) --
    the same geometry as scene 4/dragon.test under the other integrators."""
    from util import dragon_variant
    scene = SceneLoader().loadText(dragon_variant(kind, size="128 96", spp="2"))
    r = Renderer(scene)
    got = r.primary_hits()
    r.close()
    want = orc().primary_hits(scene, use_bvh=1)
    for f in ("prim", "t", "u", "v"):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
    compare_render(scene, semantics=semantics, seeding=1, use_bvh=1, bounce_cap=8, min_close=0.99, max_block_rel=0.02)
