#!/usr/bin/env python3
"""Does the speed of a frame depend on where the path state happens to be allocated?  One process builds a renderer for the
Phong Cornell box (512x512, 1024 spp per frame, one batch), times a few frames, closes it (all device memory freed) and
starts over, several times; optionally a block of device memory is held in between, so that the next allocation lands
elsewhere.  usage: realloc_states.py [rounds] [hold_GiB]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hold_gib = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
text = open(os.path.join(ROOT, "scenes", "scene3", "cornellRR.test")).read()
scene = SceneLoader().loadText(override_scene_text(text, size="512 512", spp="1024"))
held = None
for k in range(rounds):
    if hold_gib and k % 2 == 1:
        held = torch.empty(int(hold_gib * 2**30), dtype=torch.uint8, device="cuda")
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_PROFILE: 65536})
    r.render_range(0, scene.samples_per_pixel, clear=True)
    r.synchronize()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(3):
        r.render_range(0, scene.samples_per_pixel, clear=True)
    r.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = r.stats()
    print(f"round {k}: {1e3 * dt:7.2f} ms per frame; surface {st.surface_ms / 3:6.2f} gather {st.gather_ms / 3:6.2f} extend {st.extend_ms / 3:6.2f} ms"
          f"{'  (holding %.0f GiB)' % hold_gib if held is not None else ''}", flush=True)
    r.close()
    held = None
    torch.cuda.empty_cache()
