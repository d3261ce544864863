"""Host-side scene loader (C++ SceneLoader in libpthost.so) against the reference's
command grammar (OptiXRenderer/SceneLoader.cpp:36-379) and defaults
(Scene.h:37-45, Config.h:22-35)."""
import math
import os

import numpy as np
import pytest

from pathtracer_amd import SceneLoader, _ffi
from util import ROOT, scene_text

f32 = np.float32


def v(p):
    return np.array([p.x, p.y, p.z], dtype=np.float32)


def test_defaults():
    s = SceneLoader().loadText("# nothing\n\n")
    assert s.integratorName == "raytracer" and s.outputFilename == "raytrace.png"
    assert (s.light_samples, s.light_stratify, s.samples_per_pixel, s.sampling_method) == (1, 0, 1, 0)
    c = s.config
    assert c.gamma == 1.0 and c.maxDepth == 5 and c.next_event_est == 0 and c.russian_roul == 0
    assert c.epsilon == f32(0.0001)
    assert tuple(v(c.w)) == (1, 0, 0) and tuple(v(c.u)) == (0, 1, 0) and tuple(v(c.v)) == (0, 0, 1)
    assert c.hSize.x == 100 and c.tanHFov.x == f32(math.tan(f32(0.25) * f32(math.pi)))
    assert len(s.triangles) == len(s.spheres) == len(s.qlights) == 0


def test_missing_file_message():
    with pytest.raises(RuntimeError, match="Unable to open scene file /nonexistent/x.test"):
        SceneLoader().load("/nonexistent/x.test")


def test_camera_and_size():
    s = SceneLoader().loadText("size 640 480\ncamera 0 0 4 0 0 0 0 1 0 90\n")
    c = s.config
    assert (s.width, s.height) == (640, 480) and (c.hSize.x, c.hSize.y) == (320, 240)
    assert np.allclose(v(c.w), [0, 0, 1]) and np.allclose(v(c.u), [1, 0, 0]) and np.allclose(v(c.v), [0, 1, 0])
    assert np.isclose(c.tanHFov.y, 1.0, rtol=1e-6) and np.isclose(c.tanHFov.x, 640 / 480, rtol=1e-6)
    # camera before size uses the default half size 100x100 (SceneLoader.cpp:120)
    s2 = SceneLoader().loadText("camera 0 0 4 0 0 0 0 1 0 90\nsize 640 480\n")
    assert np.isclose(s2.config.tanHFov.x, s2.config.tanHFov.y)


def test_maxdepth_minus_one_and_parsing_quirks(capfd):
    s = SceneLoader().loadText("maxdepth -1\nspp 7\nlightsamples 9\nlightstratify on\nvertex 1 2\n1.4\n  # indented comment\nfoo bar\n")
    assert s.config.maxDepth == 0xFFFFFFFF
    assert s.samples_per_pixel == 7 and s.light_samples == 9 and s.light_stratify == 1
    assert len(s.vertices) == 0  # `vertex 1 2` fails to read value 2 and is skipped
    assert "Failed reading value 2 will skip" in capfd.readouterr().out


def test_flags():
    s = SceneLoader().loadText("nexteventestimation mis\nrussianroulette on\nimportancesampling brdf\ngamma 2.2\nintegrator pathtracer\noutput x.png\n")
    assert s.config.next_event_est == 2 and s.config.russian_roul == 1 and s.sampling_method == 2
    assert s.config.gamma == f32(2.2) and s.integratorName == "pathtracer" and s.outputFilename == "x.png"
    s = SceneLoader().loadText("nexteventestimation on\nimportancesampling cosine\nimportancesampling nonsense\n")
    assert s.config.next_event_est == 1 and s.sampling_method == 1
    s = SceneLoader().loadText("nexteventestimation whatever\nrussianroulette off\n")
    assert s.config.next_event_est == 0 and s.config.russian_roul == 0


def test_triangle_transform_and_material_state():
    text = """
maxverts 3
vertex 0 0 0
vertex 1 0 0
vertex 0 1 0
diffuse 0.1 0.2 0.3
specular 0.4 0.5 0.6
shininess 7
brdf ggx
roughness 0.25
pushTransform
translate 1 2 3
scale 2 2 2
tri 0 1 2
popTransform
brdf phong
emission 1 1 1
tri 0 2 1
popTransform
"""
    s = SceneLoader().loadText(text)
    t0, t1 = s.triangles[0], s.triangles[1]
    assert np.allclose(v(t0.v1), [1, 2, 3]) and np.allclose(v(t0.v2), [3, 2, 3]) and np.allclose(v(t0.v3), [1, 4, 3])
    assert np.allclose(v(t0.normal), [0, 0, 1])
    assert np.allclose(v(t0.mv.diffuse), [0.1, 0.2, 0.3]) and t0.mv.shininess == 7 and t0.mv.brdf_type == 1
    assert t0.mv.roughness == f32(0.25) and t0.objType == 0 and t0.mv.matType == 0
    assert np.allclose(v(t1.v1), [0, 0, 0]) and np.allclose(v(t1.normal), [0, 0, -1])
    assert t1.mv.brdf_type == 0 and tuple(v(t1.mv.emission)) == (1, 1, 1)
    # material defaults (SceneLoader.cpp:49-64)
    assert t0.mv.ior == 1 and t0.mv.sigma_a == 1 and t0.mv.sigma_t == 1 and t0.mv.g == 5


def test_rotate():
    s = SceneLoader().loadText("vertex 1 0 0\nvertex 0 1 0\nvertex 0 0 1\nrotate 0 0 1 90\ntri 0 1 2\n")
    t = s.triangles[0]
    assert np.allclose(v(t.v1), [0, 1, 0], atol=1e-6) and np.allclose(v(t.v2), [-1, 0, 0], atol=1e-6) and np.allclose(v(t.v3), [0, 0, 1], atol=1e-6)


def test_sphere_matrix():
    s = SceneLoader().loadText("translate 1 0 0\nscale 2 2 2\nsphere 0 1 0 0.5\n")
    m = np.array(list(s.spheres[0].trans), dtype=np.float32).reshape(4, 4)
    # top * T(0,1,0) * S(0.5): scale 1 overall, centre (1, 2, 0)
    assert np.allclose(m, [[1, 0, 0, 1], [0, 1, 0, 2], [0, 0, 1, 0], [0, 0, 0, 1]])
    assert s.spheres[0].objType == 0


def test_quad_light_layout():
    s = SceneLoader().loadText("translate 5 5 5\nquadLight 0 0 0  1 0 0  0 0 2  3 4 5\n")
    assert len(s.triangles) == 2 and len(s.qlights) == 1
    q = s.qlights[0]
    # not transformed by the stack; tri1 = (a, a+ab, a+ac), tri2 = (a+ab, a+ab+ac, a+ac)
    assert tuple(v(q.tri1.v1)) == (0, 0, 0) and tuple(v(q.tri1.v2)) == (1, 0, 0) and tuple(v(q.tri1.v3)) == (0, 0, 2)
    assert tuple(v(q.tri2.v1)) == (1, 0, 0) and tuple(v(q.tri2.v2)) == (1, 0, 2) and tuple(v(q.tri2.v3)) == (0, 0, 2)
    assert tuple(v(q.tri1.normal)) == (0, -2, 0)  # ab x ac, NOT normalised
    assert q.tri1.objType == 1 and q.tri2.objType == 1
    assert tuple(v(q.color)) == (3, 4, 5) and tuple(v(q.tri1.mv.emission)) == (3, 4, 5)
    assert bytes(s.triangles[0]) == bytes(q.tri1) and bytes(s.triangles[1]) == bytes(q.tri2)


def test_point_and_directional_lights():
    s = SceneLoader().loadText("attenuation 0 1 0\npoint 1 2 3 0.5 0.5 0.5\ndirectional 0 0 2 1 1 1\n")
    p, d = s.plights[0], s.dlights[0]
    assert tuple(v(p.location)) == (1, 2, 3) and tuple(v(p.attenuation)) == (0, 1, 0) and tuple(v(p.color)) == (0.5, 0.5, 0.5)
    assert np.allclose(v(d.direction), [0, 0, 1]) and tuple(v(d.color)) == (1, 1, 1)


def test_glass_and_volume_commands():
    s = SceneLoader().loadText("vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nmaterialtype glass\nior 1.5\ntri 0 1 2\nmaterialtype volumetric\nscattercoeff 0.1 0.2 0.3\ntri 0 1 2\nmaterialtype x\ntri 0 1 2\n")
    assert [t.mv.matType for t in s.triangles] == [1, 2, 0]
    assert s.triangles[0].mv.ior == f32(1.5)
    m = s.triangles[1].mv
    assert (m.sigma_a, m.sigma_s, m.g) == (f32(0.1), f32(0.2), f32(0.3)) and m.sigma_t == f32(0.1) + f32(0.2)


@pytest.mark.parametrize("rel,tris,spheres,quads,integ,size", [
    ("scene2/sphere.test", 8, 1, 3, "direct", (640, 480)),
    ("scene3/cornellRR.test", 24, 1, 1, "pathtracer", (480, 480)),
    ("scene4/cornellBRDF.test", 12, 1, 1, "pathtracer", (480, 480)),
    ("scene4/ggx.test", 8, 6, 2, "pathtracer", (1280, 280)),
    ("scene4/dragon.test", 100006, 0, 2, "pathtracer", (640, 480)),
    ("scene1/scene5.test", None, 1000, 0, "raytracer", None),
])
def test_shipped_scenes(rel, tris, spheres, quads, integ, size):
    s = SceneLoader().load(os.path.join(ROOT, "scenes", rel))
    if tris is not None:
        assert len(s.triangles) == tris
    assert len(s.spheres) == spheres and len(s.qlights) == quads and s.integratorName == integ
    if size:
        assert (s.width, s.height) == size


def test_dragon_scene_settings():
    s = SceneLoader().load(os.path.join(ROOT, "scenes", "scene4", "dragon.test"))
    assert s.samples_per_pixel == 128 and s.config.maxDepth == 5 and s.config.gamma == f32(2.2)
    assert s.sampling_method == 2 and s.config.next_event_est == 1 and s.config.russian_roul == 1
    mats = {bytes(t.mv) for t in s.triangles}
    assert len(mats) <= 4  # 84-byte material copied into each of the 100006 triangles
    assert s.triangles[4].mv.brdf_type == 1 and s.triangles[4].mv.roughness == f32(0.1)  # 0..3 are the two quad lights


def test_dragon_variants_rebuild_the_other_shipped_dragon_scenes():
    from util import dragon_variant
    d = SceneLoader().loadText(dragon_variant("direct"))
    assert d.integratorName == "direct" and d.light_samples == 25 and d.light_stratify == 1 and len(d.qlights) == 2
    assert len(d.triangles) == 100006 and abs(d.qlights[0].color.x - 10) < 1e-6 and d.triangles[4].mv.brdf_type == 0
    n = SceneLoader().loadText(dragon_variant("nee"))
    assert n.integratorName == "pathtracer" and n.samples_per_pixel == 64 and n.sampling_method == 0 and n.config.next_event_est == 1
    assert abs(n.triangles[4].mv.diffuse.y - 0.6) < 1e-6 and abs(n.triangles[100005].mv.diffuse.x - 0.7) < 1e-6
    r = SceneLoader().loadText(dragon_variant("raytracer"))
    assert r.integratorName == "raytracer" and len(r.triangles) == 100000 and len(r.plights) == 1 and len(r.qlights) == 0
    assert abs(r.triangles[0].mv.specular.x - 0.3) < 1e-6 and abs(r.config.eye.z - 20) < 1e-6
