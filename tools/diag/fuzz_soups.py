"""Soak: random triangle / ellipsoid soups of many sizes and seeds, every hierarchy builder against brute force
(primary hit records and radiance sums bit for bit).  usage: python tools/diag/fuzz_soups.py [seeds]   (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pathtracer_amd import Renderer, _ffi
from test_gpu_edges import soup

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(1234)
bad = 0
for k in range(n_seeds):
    n_tris = int(rng.choice([1, 2, 3, 4, 5, 7, 9, 16, 24, 33, 100, 300, 1025, 1500, 5000, 20000]))
    n_sph = int(rng.choice([0, 0, 1, 3, 9, 40, 200]))
    degenerate = bool(rng.integers(0, 2)) and n_sph == 0
    scene = soup(n_tris=n_tris, n_spheres=n_sph, seed=1000 + k, degenerate=degenerate)
    b = Renderer(scene, options={_ffi.OPT_BRUTE_FORCE: 1})
    want_hits = b.primary_hits()
    b.run()
    want = b.sums()
    b.close()
    for builder in (0, 1, 2, 3):
        for tune in (0x0020, 0x1120, 0x4420, 0x6620, 0x5518):
            r = Renderer(scene, options={_ffi.OPT_BVH_BUILDER: builder, _ffi.OPT_TUNE_TRAVERSAL: tune})
            hits = r.primary_hits()
            r.run()
            got = r.sums()
            r.close()
            ok = all(np.array_equal(hits[f].view(np.uint32), want_hits[f].view(np.uint32)) for f in ("prim", "t", "u", "v")) and \
                np.array_equal(got.view(np.uint32), want.view(np.uint32))
            if not ok:
                bad += 1
                print(f"MISMATCH seed {1000 + k} tris {n_tris} spheres {n_sph} degenerate {degenerate} builder {builder} tune {tune:#x}", flush=True)
    print(f"seed {1000 + k}: {n_tris} tris, {n_sph} spheres, degenerate={degenerate}: done", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
