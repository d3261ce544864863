#!/usr/bin/env python3
"""Soak check: the same frame rendered repeatedly (gather on the second stream) must give bit-identical
sums every time, and the same sums as with everything on one stream.  usage: tools/soak_determinism.py [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_scene  # noqa: E402
from pathtracer_amd import Renderer, _ffi  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for workload, spp in (("dragon", 32), ("cornell", 64), ("sphere", None)):
        scene = build_scene(workload, spp, None)
        ref = None
        for overlap in (0, 1):
            r = Renderer(scene, device=0, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_OVERLAP_GATHER: overlap})
            for k in range(n if overlap else 2):
                r.render_range(0, scene.samples_per_pixel, clear=True)
                s = r.sums().view(np.uint32).copy()
                if ref is None:
                    ref = s
                same = np.array_equal(ref, s)
                print(f"{workload} overlap={overlap} run {k}: {'same' if same else 'DIFFERENT'}", flush=True)
                if not same:
                    sys.exit(1)
            r.close()
    print("ok")


if __name__ == "__main__":
    main()
