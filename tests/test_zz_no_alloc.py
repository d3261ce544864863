"""ptc_render_range_static -- the entry point that allocates nothing -- against ptc_render_range.

Both run the same kernels on the same samples; the only difference is where the per-path and
per-pixel work arrays live (the library's fixed scratch array in device memory instead of
allocations).  Integer / float results alike must therefore be bit-identical.  Two calls in a row,
the second on a smaller image, check that nothing of the first call's contents leaks into the
second (the scratch array is never cleared as a whole)."""
import numpy as np
import pytest

from pathtracer_amd import Renderer, SceneLoader, _ffi, override_scene_text
from util import scene_text

pytestmark = pytest.mark.gpu


def render_both(text, size, spp, seeding, j0=0):
    import torch
    scene = SceneLoader().loadText(override_scene_text(text, size=size, spp=str(spp)))
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: seeding, _ffi.OPT_BOUNCE_CAP: 8})
    r.render_range(j0, spp, clear=True)
    want = r.sums()
    out = torch.full((scene.height * scene.width * 3,), 7.0, dtype=torch.float32, device="cuda")  # garbage the call must clear
    torch.cuda.synchronize()
    r.render_range_static(j0, spp, out.data_ptr(), out.numel() * 4, clear=True)
    got = out.cpu().numpy().reshape(scene.height, scene.width, 3)
    # the context's own buffers were left alone
    again = r.sums()
    r.close()
    assert np.array_equal(want.view(np.uint32), again.view(np.uint32))
    assert np.isfinite(want).mean() > 0.9995 and float(np.nansum(np.abs(want))) > 0  # (a GGX grazing-angle NaN pixel or two on the dragon, as in the reference)
    return got, want


@pytest.mark.parametrize("seeding", [1, 0])
def test_static_entry_point_matches_render_range_twice_in_a_row(seeding):
    text = scene_text("scene3/cornellRR.test")
    got, want = render_both(text, "160 120", 5, seeding)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    got, want = render_both(text, "72 40", 3, seeding)  # smaller, ragged, fewer samples
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_static_entry_point_on_the_dragon_and_a_sample_sub_range():
    text = scene_text("scene4/dragon.test")
    got, want = render_both(text, "256 192", 4, 1, j0=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    got, want = render_both(text, "96 64", 2, 1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_static_entry_point_rejects_what_does_not_fit():
    import torch
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="1024 1024", spp="1"))
    r = Renderer(scene, options={_ffi.OPT_SEEDING: 1})
    out = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="too large for the static scratch"):
        r.render_range_static(0, 1, out.data_ptr(), out.numel() * 4)
    with pytest.raises(RuntimeError, match="too small"):
        r.width, r.height = 64, 64
        r.render_range_static(0, 1, out.data_ptr(), 64 * 64 * 3 * 4 - 4)
    r.close()


def test_two_contexts_from_two_threads_get_correct_sums_or_the_busy_error():
    """All contexts on one GPU share that GPU's scratch array.  Two contexts calling from two host threads at the
    same time: every call either returns the sums ptc_render_range gives, bit for bit, or fails with the "in use"
    error having written nothing of its own -- never a success with another call's path state mixed in."""
    import threading
    import torch
    scene = SceneLoader().loadText(override_scene_text(scene_text("scene3/cornellRR.test"), size="192 160", spp="6"))
    opts = {_ffi.OPT_SEMANTICS: 1, _ffi.OPT_SEEDING: 1, _ffi.OPT_BOUNCE_CAP: 8}
    rs = [Renderer(scene, options=opts) for _ in range(2)]
    rs[0].render_range(0, 6, clear=True)
    want = rs[0].sums()
    outs = [torch.zeros(scene.height * scene.width * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    results = [[], []]
    start = threading.Barrier(2)

    def work(k):
        start.wait()
        for _ in range(6):
            try:
                rs[k].render_range_static(0, 6, outs[k].data_ptr(), outs[k].numel() * 4, clear=True)
                got = outs[k].cpu().numpy().reshape(scene.height, scene.width, 3)
                results[k].append(bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))))
            except RuntimeError as e:
                assert "in use" in str(e), e
                results[k].append(None)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in rs:
        r.close()
    flat = results[0] + results[1]
    assert len(flat) == 12 and False not in flat, results
    assert flat.count(True) >= 2  # a call is only ever refused while the other context's call is running
