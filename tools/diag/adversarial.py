"""Odd scenes against brute force (primary hits and radiance sums bit for bit), default options.  GPU box."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from pathtracer_amd import Renderer, SceneLoader, _ffi

HEAD = ["size 96 72", "integrator pathtracer", "spp 2", "maxdepth 3", "nexteventestimation on", "russianroulette on", "importancesampling brdf",
        "camera 0 2 6 0 0 0 0 1 0 50", "quadLight -1 4 -1 2 0 0 0 0 2 9 9 9", "brdf ggx", "roughness 0.3", "diffuse 0.6 0.5 0.4", "specular 0.2 0.2 0.2"]


def tri_scene(tris):
    lines = list(HEAD) + [f"maxverts {3 * len(tris)}"]
    for t in tris:
        for v in t:
            lines.append("vertex %.9g %.9g %.9g" % tuple(v))
    lines += [f"tri {3 * i} {3 * i + 1} {3 * i + 2}" for i in range(len(tris))]
    return SceneLoader().loadText("\n".join(lines) + "\n")


def sphere_scene(n, rng):
    lines = list(HEAD)
    for _ in range(n):
        x, y, z = rng.uniform(-3, 3, 3)
        lines += ["pushTransform", f"translate {x:.4f} {y:.4f} {z:.4f}", f"sphere 0 0 0 {rng.uniform(0.01, 0.15):.3f}", "popTransform"]
    return SceneLoader().loadText("\n".join(lines) + "\n")


def cases():
    rng = np.random.default_rng(1)
    g = 60
    flat = []
    for i in range(g):
        for j in range(g):
            x, z = -3 + 6 * i / g, -3 + 6 * j / g
            d = 6.0 / g
            flat.append([(x, 0, z), (x + d, 0, z), (x, 0, z + d)])
            flat.append([(x + d, 0, z), (x + d, 0, z + d), (x, 0, z + d)])
    yield "flat grid (zero extent in y)", tri_scene(flat)
    line = []
    for i in range(4000):
        s = 10 ** rng.uniform(-3, 1)
        x = -3 + 6 * i / 4000
        line.append([(x, 0, 0), (x + s, 0, 0), (x, s, 0)])
    yield "collinear, sizes over four decades", tri_scene(line)
    yield "one triangle", tri_scene([[(-1, 0, 0), (1, 0, 0), (0, 1, 0)]])
    yield "two triangles", tri_scene([[(-1, 0, 0), (1, 0, 0), (0, 1, 0)], [(-1, 0, -1), (1, 0, -1), (0, 1, -1)]])
    yield "five huge triangles", tri_scene([[(-900 + k, -1, -900), (900, -1, -900 + k), (0, -1 - k, 900)] for k in range(5)])
    yield "5000 spheres", sphere_scene(5000, rng)
    nest = []
    for k in range(3000):
        s = 2.0 * 0.997 ** k
        nest.append([(-s, -s, -k * 1e-3), (s, -s, -k * 1e-3), (0, s, -k * 1e-3)])
    yield "3000 nested triangles", tri_scene(nest)


for name, scene in cases():
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1})
    want_hits = b.primary_hits(); b.run(); want = b.sums(); b.close()
    for builder in (0, 1, 2, 3):
        try:
            r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder})
            hits = r.primary_hits(); r.run(); got = r.sums(); st = r.stats(); r.close()
            ok = all(np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)) for f in ("prim", "t", "u", "v")) and \
                np.array_equal(got.view(np.uint32), want.view(np.uint32))
            print(f"{name}: builder {builder}: {'ok' if ok else 'MISMATCH'} nodes {st.bvh_nodes} depth {st.bvh_max_depth} hit fraction {(want_hits['prim'] >= 0).mean():.2f}", flush=True)
        except RuntimeError as e:
            print(f"{name}: builder {builder}: FAIL {e}", flush=True)
