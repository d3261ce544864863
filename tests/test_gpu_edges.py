"""GPU edge cases: ragged image sizes, empty and degenerate scenes, sphere hierarchies,
exact ties, progressive frames, error behaviour, and the C++ command line end to end.
Everything goes through the C ABI of libptcore.so and is checked against the CPU oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text
from util import ROOT, read_png, scene_text

pytestmark = pytest.mark.gpu


def orc():
    from oracle import oracle_py
    return oracle_py


def hits_equal(scene, options=None):
    r = Renderer(scene, options=options or {})
    got = r.primary_hits()
    st = r.stats()
    r.close()
    want = orc().primary_hits(scene)
    assert np.array_equal(got["prim"], want["prim"]), f"{(got['prim'] != want['prim']).sum()} primitive ids differ"
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
    return got, st


def sums_close(scene, **kw):
    r = Renderer(scene, options={_ffi.OPT_SEEDING: kw.get("seeding", 0), _ffi.OPT_SEMANTICS: kw.get("semantics", 0), _ffi.OPT_BOUNCE_CAP: 8})
    r.run()
    got = r.sums()
    st = r.stats()
    r.close()
    want, _, ost = orc().render(scene, seeding=kw.get("seeding", 0), semantics=kw.get("semantics", 0), bounce_cap=8)
    assert st.paths == ost.paths
    ok = np.isclose(got, want, rtol=1e-3, atol=1e-3, equal_nan=True)
    assert ok.all(axis=2).mean() >= 0.99
    return got


@pytest.mark.parametrize("size", ["1 1", "7 5", "37 23", "65 9", "8 64", "64 64", "130 67", "129 128"])
def test_ragged_image_sizes(size):
    """Sizes that are not multiples of the 8x8 tile a wave renders (down to one pixel)."""
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size=size, spp="3"))
    hits_equal(scene)
    sums_close(scene, seeding=1, semantics=1)
    sums_close(scene, seeding=0, semantics=1)


def test_empty_scene_is_all_background():
    scene = SceneLoader().loadText("size 40 24\nintegrator pathtracer\nspp 2\ncamera 0 0 5 0 0 0 0 1 0 45\n")
    r = Renderer(scene)
    hits = r.primary_hits()
    r.run()
    res, st = r.result(), r.stats()
    r.close()
    assert (hits["prim"] == -1).all()
    assert not res.any()
    assert st.paths == 40 * 24 * 2 and st.extension_rays == st.paths and st.shadow_rays == 0


def test_background_colour_is_added_unscaled():
    """Common.cu:13-18: a miss adds the background colour, not multiplied by the throughput."""
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene1/scene4-specular.test"), size="64 48"))
    r = Renderer(scene)
    lib = _ffi.core_lib()
    lib.ptc_set_background(r._ctx, C.c_float(0.25), C.c_float(0.5), C.c_float(1.0))
    r.run()
    with_bg = r.sums()
    lib.ptc_set_background(r._ctx, C.c_float(0.0), C.c_float(0.0), C.c_float(0.0))
    r.currentFrame = 0
    r.run()
    without = r.sums()
    hits = r.primary_hits()
    r.close()
    miss = hits["prim"] == -1
    assert miss.any() and (~miss).any()
    assert np.allclose((with_bg - without)[miss], [0.25, 0.5, 1.0])
    # mirror paths that leave the scene pick the colour up as well, never more than once
    d = (with_bg - without)[~miss]
    assert (d >= -1e-6).all() and (d <= np.array([0.25, 0.5, 1.0]) + 1e-5).all()


def sphere_cloud(n, seed=3, extra=""):
    rng = np.random.default_rng(seed)
    lines = ["size 96 72", "integrator raytracer", "maxdepth 3", "camera 0 0 9 0 0 0 0 1 0 45", "point 4 6 8 1 1 1", "directional 0 -1 -1 0.4 0.4 0.3",
             "ambient 0.05 0.05 0.05", "shininess 20"]
    for i in range(n):
        c = rng.uniform(0.1, 0.9, 3)
        x, y, z = rng.uniform(-3, 3, 3)
        lines += [f"diffuse {c[0]:.3f} {c[1]:.3f} {c[2]:.3f}", f"specular {0.3 * (i % 2)} {0.3 * (i % 2)} {0.3 * (i % 2)}", "pushTransform",
                  f"translate {x:.4f} {y:.4f} {z:.4f}", f"scale {rng.uniform(0.5, 1.5):.3f} {rng.uniform(0.5, 1.5):.3f} {rng.uniform(0.5, 1.5):.3f}",
                  f"rotate 0 1 1 {rng.uniform(0, 90):.2f}", f"sphere 0 0 0 {rng.uniform(0.15, 0.45):.3f}", "popTransform"]
    return SceneLoader().loadText("\n".join(lines) + "\n" + extra)


@pytest.mark.parametrize("builder", [0, 1, 4])
def test_sphere_hierarchy(builder):
    """150 transformed ellipsoids: the sphere tree (not just the <= 8 sphere top-level leaf of
    the shipped scenes), reflections, point + directional lights."""
    scene = sphere_cloud(150)
    _, st = hits_equal(scene, {_ffi.OPT_BVH_BUILDER: builder})
    assert st.bvh_nodes > 20
    sums_close(scene)


def test_spheres_and_triangles_share_rays():
    tri = "maxverts 4\nvertex -4 -3 -4\nvertex 4 -3 -4\nvertex 4 3 -4\nvertex -4 3 -4\ndiffuse 0.6 0.6 0.6\ntri 0 1 2\ntri 0 2 3\n"
    scene = sphere_cloud(40, seed=5, extra=tri)
    hits, _ = hits_equal(scene)
    assert (hits["prim"] < 2).any() and (hits["prim"] >= 2).any()
    sums_close(scene)


def test_exact_ties_go_to_the_lowest_primitive_index():
    """Coincident triangles give bit-equal distances; the closest hit is then the lowest
    global primitive index, on every traversal kernel."""
    text = ("size 48 48\nintegrator raytracer\ncamera 0 0 4 0 0 0 0 1 0 50\npoint 0 0 5 1 1 1\nmaxverts 3\n"
            "vertex -1 -1 0\nvertex 1 -1 0\nvertex 0 1 0\ndiffuse 1 0 0\ntri 0 1 2\ndiffuse 0 1 0\ntri 0 1 2\ndiffuse 0 0 1\ntri 0 1 2\n")
    scene = SceneLoader().loadText(text)
    for tune in (0x0020, 0x1120, 0x4420, 0x5518, 0x6620, 0x6604):
        hits, _ = hits_equal(scene, {_ffi.OPT_TUNE_TRAVERSAL: tune})
        assert set(np.unique(hits["prim"])) == {-1, 0}


def test_progressive_frames_lerp():
    """PinholeCamera.cu:73-80: frame 1 stores, frame n blends with weight 1/n; the seeds of
    frame n are keyed on index * n."""
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellNEE.test"), size="48 48", spp="2"))
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1})
    r.numFrames = 3
    r.run(progressive=False)
    got = r.result()
    assert r.getCurrentFrame() == 3
    r.close()
    want = None
    for f in (1, 2, 3):
        _, res, _ = orc().render(scene, semantics=1, frame_id=f, bounce_cap=1024)
        want = res if f == 1 else want + np.float32(1.0 / f) * (res - want)
    assert np.isclose(got, want, rtol=2e-3, atol=2e-3).all(axis=2).mean() >= 0.99


def test_error_behaviour():
    lib = _ffi.core_lib()
    ctx = C.c_void_p()
    assert lib.ptc_create(0, C.byref(ctx)) == 0
    try:
        assert lib.ptc_render(ctx, 1, 16, 16) == _ffi.ERR_INVALID           # nothing configured
        assert b"ptc_set_config" in lib.ptc_last_error(ctx)
        scene = SceneLoader().loadText(override_scene_text(scene_text("scene2/sphere.test"), size="16 16"))
        assert lib.ptc_set_config(ctx, C.byref(scene.config)) == 0
        assert lib.ptc_set_integrator(ctx, b"bidirectional") == _ffi.ERR_INTEGRATOR
        assert lib.ptc_last_error(ctx) == b"Doesn't support the integrator bidirectional"
        assert lib.ptc_set_integrator(ctx, b"direct") == 0
        assert lib.ptc_set_triangles(ctx, C.cast(scene.triangles, C.c_void_p), len(scene.triangles)) == 0
        assert lib.ptc_render(ctx, 1, 16, 16) == _ffi.ERR_NOT_BUILT          # geometry uploaded, no ptc_build
        assert lib.ptc_build(ctx) == 0
        assert lib.ptc_render(ctx, 1, 0, 16) == _ffi.ERR_INVALID             # empty image
        assert lib.ptc_render_range(ctx, 1, 16, 16, 3, 2, 1) == _ffi.ERR_INVALID  # reversed range
        assert lib.ptc_render_range(ctx, 1, 16, 16, 0, 0, 1) == 0                 # an empty range is no error
        assert lib.ptc_render(ctx, 1, 70000, 70000) == _ffi.ERR_INVALID      # more pixels than 32-bit pixel numbers
        assert b"too large" in lib.ptc_last_error(ctx)
        assert lib.ptc_render_range(ctx, 1, 16, 16, 2, 4, 1) == _ffi.ERR_INVALID  # reference seeding cannot start at sample 2
        assert lib.ptc_set_option(ctx, 12345, 1) == _ffi.ERR_INVALID
        assert lib.ptc_set_option(ctx, _ffi.OPT_BOUNCE_CAP, 0) == _ffi.ERR_INVALID
        assert lib.ptc_render(ctx, 1, 16, 16) == 0                           # and the context still works
    finally:
        lib.ptc_destroy(ctx)


def test_command_line_writes_the_same_png(tmp_path):
    """The C++ Renderer + main (scene file in, PNG out) against the Python mirror."""
    exe = os.path.join(ROOT, "pathtracer_amd", "_lib", "pathtracer")
    assert os.path.exists(exe), "build() has not produced the command-line tool"
    text = override_scene_text(scene_text("scene2/sphere.test"), size="160 120", output=str(tmp_path / "cli.png"))
    path = tmp_path / "sphere.test"
    path.write_text(text)
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "Elapsed time:" in out.stdout and "cli.png is saved." in out.stdout, out.stdout + out.stderr
    scene = SceneLoader().loadText(text)
    r = Renderer(scene)
    r.run()
    mine = np.frombuffer(bytes(r.getResult()), dtype=np.uint8).reshape(120, 160, 4)
    r.close()
    assert np.array_equal(read_png(str(tmp_path / "cli.png")), mine)
    # errors are printed, the exit status stays 0 (main.cpp:47-52)
    bad = subprocess.run([exe, str(tmp_path / "missing.test")], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 0 and bad.stderr.strip()


def test_command_line_multi_gpu_path(tmp_path):
    """`pathtracer <scene> --gpus N`: one context, stream and host thread per device, samples split, ONE
    ncclReduce of the sums to device 0, resolve there.  With the devices this box has (normally one, where the
    communicator has a single rank) the image equals the single-GPU command line's with counter seeds byte for
    byte; `--ordered-sum` (host adds in device order) as well; asking for more GPUs than present is an error."""
    import torch
    exe = os.path.join(ROOT, "pathtracer_amd", "_lib", "pathtracer")
    base = override_scene_text(scene_text("scene3/cornellRR.test"), size="96 80", spp="6", maxdepth="6")
    imgs = {}
    for tag, flags in (("single", []), ("rccl", ["--gpus", "1"]), ("ordered", ["--gpus", "1", "--ordered-sum"])):
        path = tmp_path / f"{tag}.test"
        path.write_text(override_scene_text(base, output=str(tmp_path / f"{tag}.png")))
        out = subprocess.run([exe, str(path), "--intended", "--counter-seeds"] + flags, capture_output=True, text=True, timeout=180)
        assert out.returncode == 0 and f"{tag}.png is saved." in out.stdout, out.stdout + out.stderr
        if flags:
            assert "rendering on 1 GPU(s)" in out.stdout
        imgs[tag] = read_png(str(tmp_path / f"{tag}.png"))
    assert np.array_equal(imgs["single"], imgs["rccl"]) and np.array_equal(imgs["single"], imgs["ordered"])
    assert imgs["single"][..., :3].max() > 0
    n = torch.cuda.device_count()
    if n >= 2:  # a multi-GPU box: the split over all devices against the single-GPU image (summation order differs)
        path = tmp_path / "all.test"
        path.write_text(override_scene_text(base, output=str(tmp_path / "all.png")))
        out = subprocess.run([exe, str(path), "--intended", "--counter-seeds", "--gpus", str(n)], capture_output=True, text=True, timeout=300)
        assert f"rendering on {n} GPU(s)" in out.stdout, out.stdout + out.stderr
        assert np.abs(read_png(str(tmp_path / "all.png")).astype(int) - imgs["single"].astype(int)).max() <= 1
    bad = subprocess.run([exe, str(tmp_path / "single.test"), "--gpus", str(n + 1)], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 0 and f"asked for {n + 1} GPUs, {n} present" in bad.stderr


def test_command_line_several_contexts_on_one_device(tmp_path):
    """`--gpus 3 --ordered-sum --share-devices` on whatever devices are present: three contexts, three host threads,
    the sample ranges [0,2) [2,4) [4,6), host adds in context order.  The image must be the one obtained by
    rendering those ranges one after the other on one context and adding the three sums in that order."""
    import torch
    from pathtracer_amd import _ffi as ffi
    exe = os.path.join(ROOT, "pathtracer_amd", "_lib", "pathtracer")
    text = override_scene_text(scene_text("scene3/cornellRR.test"), size="96 80", spp="6", maxdepth="6", output=str(tmp_path / "three.png"))
    path = tmp_path / "three.test"
    path.write_text(text)
    out = subprocess.run([exe, str(path), "--intended", "--counter-seeds", "--gpus", "3", "--ordered-sum", "--share-devices"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rendering on 3 GPU(s)" in out.stdout and "three.png is saved." in out.stdout, out.stdout + out.stderr
    got = read_png(str(tmp_path / "three.png"))
    r = Renderer(SceneLoader().loadText(text), options={ffi.OPT_SEMANTICS: 1, ffi.OPT_SEEDING: 1})
    buf = torch.zeros(80 * 96 * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    r.bind_sum_buffer(buf.data_ptr(), buf.numel() * 4)
    total = None
    for a, b in ((0, 2), (2, 4), (4, 6)):
        r.render_range(a, b, clear=True)
        r.synchronize()
        total = buf.clone() if total is None else total + buf
        torch.cuda.synchronize()  # torch's stream has read the sums before the next range clears them on the renderer's stream
    buf.copy_(total)
    torch.cuda.synchronize()
    r.resolve(6)
    res = r.result()
    r.close()
    want = np.empty((80, 96, 4), dtype=np.uint8)
    ffi.host_lib().pth_result_to_rgba8(res.ctypes.data, 96, 80, want.ctypes.data)
    assert np.array_equal(got, want)
    # without the ordered sum a device cannot be used twice
    bad = subprocess.run([exe, str(path), "--gpus", "2", "--share-devices"], capture_output=True, text=True, timeout=60)
    assert "ordered sum" in bad.stderr or "present" in bad.stderr


def soup(n_tris, n_spheres, seed, degenerate=False):
    """Random triangle soup (+ ellipsoids) around the origin with the camera inside it."""
    rng = np.random.default_rng(seed)
    lines = ["size 80 60", "integrator raytracer", "maxdepth 2", "camera 0.1 0.05 0.2 0 0 -1 0 1 0 70", "point 0 0 0 1 1 1", "diffuse 0.5 0.5 0.5",
             f"maxverts {3 * n_tris}"]
    verts, tris = [], []
    for i in range(n_tris):
        c = rng.uniform(-2, 2, 3)
        if degenerate and i % 3 == 0:
            c = np.array([0.5, 0.5, -1.0])          # many primitives with the same centroid
        size = 0.06 if not degenerate or i % 5 else 3.0  # a few huge ones among tiny ones
        for _ in range(3):
            v = c + rng.uniform(-size, size, 3)
            verts.append(f"vertex {v[0]:.5f} {v[1]:.5f} {v[2]:.5f}")
        if degenerate and i % 7 == 0:
            verts[-1] = verts[-2]                      # zero-area triangle
        tris.append(f"tri {3 * i} {3 * i + 1} {3 * i + 2}")
    lines += verts + tris
    for _ in range(n_spheres):
        x, y, z = rng.uniform(-2, 2, 3)
        lines += ["pushTransform", f"translate {x:.4f} {y:.4f} {z:.4f}", f"scale {rng.uniform(0.2, 2):.3f} 1 {rng.uniform(0.2, 2):.3f}",
                  f"sphere 0 0 0 {rng.uniform(0.02, 0.2):.3f}", "popTransform"]
    return SceneLoader().loadText("\n".join(lines) + "\n")


@pytest.mark.parametrize("builder", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("case", [dict(n_tris=6000, n_spheres=60, seed=11), dict(n_tris=3000, n_spheres=0, seed=12, degenerate=True)])
def test_random_soups_against_brute_force(case, builder):
    """Camera inside a cloud of small, huge, coincident and zero-area primitives: every
    traversal kernel must agree bit for bit with brute force over all primitives."""
    scene = soup(**case)
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1})
    want_hits = b.primary_hits()
    b.run()
    want = b.sums()
    b.close()
    assert (want_hits["prim"] >= 0).mean() > 0.1
    for tune in (0x0020, 0x1120, 0x2210, 0x4420, 0x4404, 0x6620, 0x6604):
        r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder, _ffi.OPT_TUNE_TRAVERSAL: tune})
        hits = r.primary_hits()
        r.run()
        got = r.sums()
        r.close()
        for f in ("prim", "t", "u", "v"):
            assert np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)), (hex(tune), f)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), hex(tune)


@pytest.mark.parametrize("n_big,tilt,eye_height", [(1, 0.0, 0.3), (2, 0.0, 0.02), (2, 0.37, 0.3), (3, 0.0, 1e-3), (4, 0.0, 0.3)])
@pytest.mark.parametrize("integrator", ["pathtracer", "direct"])
def test_marked_rays_over_large_triangles(n_big, tilt, eye_height, integrator):
    """A cloud of 3000 small triangles over a floor of 1-4 triangles a thousand times their size (with one to three of
    them the surface kernel marks the rays that cannot hit the floor and the traversal starts those below the floor's
    leaf; four are too many and go the ordinary way), the floor level or tilted, the camera well above it or a hair
    above it so that most rays graze it, a quad light overhead: hits, sums and ray counts equal brute force over all
    primitives bit for bit, for both hierarchies and with the marks off as well."""
    rng = np.random.default_rng(40 + n_big)
    n_small = 3000
    lines = ["size 96 72", f"integrator {integrator}", "spp 4", "maxdepth 4", "nexteventestimation on", "russianroulette on", "importancesampling brdf",
             f"camera 0 {eye_height} 2.5 0 0.2 0 0 1 0 60", "quadLight -0.5 3 -0.5 0 0 1 1 0 0 9 9 9", f"maxverts {3 * n_small + 8}", "diffuse 0.5 0.5 0.5",
             "specular 0.2 0.2 0.2", "shininess 10"]
    verts, tris = [], []
    for i in range(n_small):
        c = rng.uniform(-1.5, 1.5, 3) * np.array([1, 0.5, 1]) + np.array([0, 0.8, 0])
        for _ in range(3):
            v = c + rng.uniform(-0.05, 0.05, 3)
            verts.append(f"vertex {v[0]:.5f} {v[1]:.5f} {v[2]:.5f}")
        tris.append(f"tri {3 * i} {3 * i + 1} {3 * i + 2}")
    base = 3 * n_small
    e = 60.0
    corners = [(-e, -e), (e, -e), (e, e), (-e, e), (0.0, 0.0)]
    for x, z in corners:
        verts.append(f"vertex {x} {tilt * x / e:.6f} {z}")
    if n_big <= 2:
        floor = [(0, 1, 2), (0, 2, 3)][:n_big]
    elif n_big == 3:
        floor = [(0, 1, 4), (1, 2, 4), (2, 3, 0)]
    else:
        floor = [(0, 1, 4), (1, 2, 4), (2, 3, 4), (3, 0, 4)]
    tris += [f"tri {base + a} {base + c} {base + b}" for a, b, c in floor]  # facing up
    scene = SceneLoader().loadText("\n".join(lines + verts + tris) + "\n")
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_SEMANTICS: 1})
    want_hits = b.primary_hits()
    b.run()
    want, want_rays = b.sums().copy(), (b.stats().extension_rays, b.stats().shadow_rays)
    b.close()
    assert (want_hits["prim"] >= 3000).mean() > 0.03 and want_rays[1] > 0  # the floor is seen, shadow rays are traced
    tests = {}
    for builder in (3, 1):
        for mark in (1, 0):
            r = Renderer(scene, options={_ffi.OPT_SEEDING: 1, _ffi.OPT_SEMANTICS: 1, _ffi.OPT_BVH_BUILDER: builder, _ffi.OPT_MARK_RAYS: mark,
                                         _ffi.OPT_COUNT_TRAVERSAL: 1, _ffi.OPT_TUNE_TRAVERSAL: 0x6620})
            hits = r.primary_hits()
            r.reset_stats()
            r.run()
            st = r.stats()
            got = r.sums()
            r.close()
            for f in ("prim", "t", "u", "v"):
                assert np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)), (builder, mark, f)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (builder, mark)
            assert (st.extension_rays, st.shadow_rays) == want_rays, (builder, mark)
            tests[builder, mark] = st.prims_tested + st.shadow_prims_tested
        # the marks save triangle tests where they apply (the counts vary by a few in ten thousand from run to run: which queued
        # test sees which closest distance depends on the order in which the waves' batches run)
        if n_big <= 3 and integrator == "pathtracer":  # (the marking variant of the surface kernel exists for the path tracer)
            assert tests[builder, 1] < 0.995 * tests[builder, 0], tests
        else:
            assert abs(tests[builder, 1] - tests[builder, 0]) < 0.01 * tests[builder, 0], tests


@pytest.mark.parametrize("n_tris,n_spheres", [(1, 0), (0, 1), (7, 1), (24, 1), (30, 0), (12, 6), (3, 9), (36, 2), (60, 6), (12, 21), (100, 40), (190, 0), (400, 0)])
def test_small_scenes_against_brute_force(n_tris, n_spheres):
    """Scenes of a few dozen primitives: the refilling kernels traverse a copy of the hierarchy and the primitives
    in LDS when the hierarchy has at most 64 nodes and everything fits 16 KB (with 8 stack entries per ray in LDS
    and the rest in scratch), and the arrays in memory otherwise (the last cases); both, the one-ray-per-thread
    kernels and the automatic choice agree bit for bit with brute force, for both ray classes (closest-hit camera
    and reflection rays, any-hit shadow rays to the point light)."""
    scene = soup(n_tris=n_tris, n_spheres=n_spheres, seed=100 + n_tris + n_spheres)
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1})
    want_hits = b.primary_hits()
    b.run()
    want = b.sums()
    b.close()
    for builder in (0, 3):
        for tune in (0x1118, 0x1104, 0x2218, 0x0018, 0x5518):
            r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder, _ffi.OPT_TUNE_TRAVERSAL: tune})
            hits = r.primary_hits()
            r.run()
            got = r.sums()
            r.close()
            for f in ("prim", "t", "u", "v"):
                assert np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)), (builder, hex(tune), f)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (builder, hex(tune))


@pytest.mark.parametrize("builder", [0, 1, 2, 3, 4])
def test_many_coincident_primitives(builder):
    """600 exact copies of one triangle inside a soup (an instanced mesh exported twice, stacked decals ...): the
    clustering and the sweep must pair the copies up evenly -- merged one pair per round, or split off one at a
    time, they make a chain deeper than the traversal stack and the build is refused."""
    import ctypes as C
    scene = soup(n_tris=1500, n_spheres=0, seed=5)
    n = len(scene.triangles)
    raw = np.frombuffer(scene.triangles, dtype=np.uint8).reshape(n, C.sizeof(_ffi.Triangle)).copy()
    out = np.concatenate([raw, np.tile(raw[:1], (600, 1))])
    scene.triangles = (_ffi.Triangle * len(out)).from_buffer_copy(out.tobytes())
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1})
    want_hits = b.primary_hits()
    b.run()
    want = b.sums()
    b.close()
    r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder})
    assert r.stats().bvh_max_depth <= 40
    hits = r.primary_hits()
    r.run()
    got = r.sums()
    r.close()
    for f in ("prim", "t", "u", "v"):
        assert np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)), f
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_more_primitives_than_get_two_hierarchies():
    """540 000 triangles, more than the 2^19 primitives up to which closest-hit and any-hit rays get a hierarchy
    each (beyond it one clustered hierarchy serves both): the default builder against brute force."""
    import ctypes as C
    scene = soup(n_tris=6000, n_spheres=0, seed=21)
    n, size = len(scene.triangles), C.sizeof(_ffi.Triangle)
    raw = np.frombuffer(scene.triangles, dtype=np.uint8).reshape(n, size)
    copies = 90
    out = np.tile(raw, (copies, 1))
    corners = out[:, :36].copy().view(np.float32).reshape(copies, n, 3, 3)  # v1, v2, v3 lead the record
    shift = np.random.default_rng(3).uniform(-3, 3, (copies, 1, 1, 3)).astype(np.float32)
    shift[0] = 0  # the first copy stays around the camera
    corners += shift
    out[:, :36] = corners.reshape(copies * n, 9).view(np.uint8)
    scene.triangles = (_ffi.Triangle * len(out)).from_buffer_copy(out.tobytes())
    assert len(scene.triangles) > 1 << 19
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1})
    want_hits = b.primary_hits()
    b.run()
    want = b.sums()
    b.close()
    assert (want_hits["prim"] >= 0).mean() > 0.5
    r = Renderer(scene)
    hits = r.primary_hits()
    r.run()
    got = r.sums()
    r.close()
    for f in ("prim", "t", "u", "v"):
        assert np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)), f
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("far", ["1e30", "3e38"])
def test_extreme_coordinates_against_brute_force(far):
    """A small soup in front of a backdrop triangle with astronomically large coordinates.  With
    1e30 the four-wide nodes near the root get grid steps of 2^92; with 3e38 a node's extent cannot
    be represented on its 8-bit grid at all and the build keeps the two-child hierarchy only.
    Either way every traversal kernel agrees bit for bit with brute force."""
    rng = np.random.default_rng(5)
    lines = ["size 64 48", "integrator raytracer", "maxdepth 2", "camera 0 0 3 0 0 -1 0 1 0 60", "point 0 2 3 1 1 1", "diffuse 0.5 0.5 0.5", "maxverts 2000"]
    verts, tris = [], []
    for i in range(400):
        c = rng.uniform(-1.5, 1.5, 3)
        for _ in range(3):
            v = c + rng.uniform(-0.1, 0.1, 3)
            verts.append(f"vertex {v[0]:.5f} {v[1]:.5f} {v[2]:.5f}")
        tris.append(f"tri {3 * i} {3 * i + 1} {3 * i + 2}")
    verts += [f"vertex -{far} -{far} -{far}", f"vertex {far} -{far} -{far}", f"vertex 0 {far} -{far}"]
    tris.append("tri 1200 1201 1202")
    scene = SceneLoader().loadText("\n".join(lines + verts + tris) + "\n")
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1})
    want_hits = b.primary_hits()
    b.run()
    want = b.sums()
    b.close()
    assert (want_hits["prim"] >= 0).mean() > 0.05
    for tune in (0x0020, 0x4420, 0x6620, 0x5518):
        r = Renderer(scene, options={_ffi.OPT_TUNE_TRAVERSAL: tune})
        hits = r.primary_hits()
        r.run()
        got = r.sums()
        r.close()
        for f in ("prim", "t", "u", "v"):
            assert np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)), (hex(tune), f)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), hex(tune)


def test_undersized_sum_buffer_bound_before_the_first_render_is_rejected():
    """ptc_bind_sum_buffer before any image size is known cannot check the size; the first render
    must (instead of writing past the end of the caller's buffer)."""
    import torch
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="64 64", spp="1"))
    r = Renderer(scene, options={_ffi.OPT_SEEDING: 1})
    small = torch.zeros(64 * 64 * 3 - 1, dtype=torch.float32, device="cuda")
    r.bind_sum_buffer(small.data_ptr(), small.numel() * 4)
    with pytest.raises(RuntimeError, match="too small"):
        r.render_range(0, 1)
    big = torch.zeros(64 * 64 * 3, dtype=torch.float32, device="cuda")
    r.bind_sum_buffer(big.data_ptr(), big.numel() * 4)
    r.render_range(0, 1)
    r.synchronize()
    assert int((big != 0).sum()) > 0
    # after a size change the check still holds
    r.width, r.height = 96, 64
    with pytest.raises(RuntimeError, match="too small"):
        r.render_range(0, 1)
    r.close()


def test_stats_report_the_batch_footprint():
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="64 64", spp="4"))
    r = Renderer(scene, options={_ffi.OPT_SEEDING: 1})
    r.run()
    st = r.stats()
    r.close()
    assert st.batch_paths == 64 * 64 * 4
    # one quad light -> one light sample per segment: 196 B per path plus alignment slack
    assert 196 * st.batch_paths <= st.path_state_bytes <= 196 * st.batch_paths + 17 * 256


@pytest.mark.parametrize("which", [0, 1])
def test_device_built_wide_nodes(which):
    """The four-wide nodes the build makes on the GPU, read back: every stored child box contains everything below
    that child and is no looser than one grid step, every leaf of the two-child hierarchy appears exactly once
    (the checks of tests/test_wide_nodes.py), and node for node they are what the host-side collapse makes of the same
    two-child nodes -- same children in the same order, same origin, scale and byte planes; only the numbering of the
    nodes differs (the device numbers level by level)."""
    import ctypes as C
    from test_wide_nodes import EMPTY, WIDE, check
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene4/dragon.test"), size="64 48", spp="1"))
    r = Renderer(scene)
    bin_nodes, wide = r.hierarchy(which)
    st = r.stats()
    r.close()
    assert st.build_host_ms == 0.0 and len(bin_nodes) > 40000 and 0 < len(wide) < len(bin_nodes)
    unused = (1 << 30) | len(scene.triangles)  # a leaf of one triangle: the all-zero one after the scene's
    # the same two-child nodes through the host-side collapse
    lib = _ffi.core_lib()
    lib.ptc_collapse_wide.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    host = np.zeros(len(bin_nodes), dtype=WIDE)
    n, depth = C.c_uint32(0), C.c_uint32(0)
    assert lib.ptc_collapse_wide(bin_nodes.ctypes.data, len(bin_nodes), host.ctypes.data, len(host), C.byref(n), C.byref(depth)) == 0
    assert n.value == len(wide)
    todo, visited = [(0, 0)], 0
    while todo:
        d, h = todo.pop()
        visited += 1
        a, b = wide[d], host[h]
        assert np.array_equal(a["origin"], b["origin"]) and np.array_equal(a["scale"], b["scale"])
        for c in range(4):
            rd, rh = int(a["child"][c]), int(b["child"][c])
            if rh == EMPTY:
                assert rd == unused and (a["qlo"][:, c] == 255).all() and (a["qhi"][:, c] == 0).all()
                continue
            assert np.array_equal(a["qlo"][:, c], b["qlo"][:, c]) and np.array_equal(a["qhi"][:, c], b["qhi"][:, c])
            if rh >> 30 == 0:
                assert rd >> 30 == 0
                todo.append((rd, rh))
            else:
                assert rd == rh
    assert visited == len(wide)
    # containment and tightness against the leaf boxes of the two-child nodes
    boxes = {}
    for nd in bin_nodes:
        for c, (lo, hi) in enumerate((("lo0", "hi0"), ("lo1", "hi1"))):
            ref = int(nd["child"][c])
            if ref != EMPTY and ref >> 30 != 0:
                boxes[ref] = (nd[lo].copy(), nd[hi].copy())
    plain = wide.copy()
    plain["child"][plain["child"] == unused] = EMPTY
    import sys
    sys.setrecursionlimit(10000)
    check(plain, boxes)
