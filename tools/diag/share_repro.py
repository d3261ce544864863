"""Which side of test_command_line_several_contexts_on_one_device varies from run to run: the command line with three
contexts on one device, or the three ranges rendered one after the other on one context?
usage: python tools/diag/share_repro.py   (GPU box, repository root)"""
import os, subprocess, sys, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from pathtracer_amd import Renderer, SceneLoader, _ffi as ffi, override_scene_text
from util import ROOT, read_png, scene_text

tmp = tempfile.mkdtemp()
exe = os.path.join(ROOT, "pathtracer_amd", "_lib", "pathtracer")
text = override_scene_text(scene_text("scene3/cornellRR.test"), size="96 80", spp="6", maxdepth="6", output=os.path.join(tmp, "three.png"))
path = os.path.join(tmp, "three.test")
open(path, "w").write(text)
cli = []
for k in range(8):
    out = subprocess.run([exe, path, "--intended", "--counter-seeds", "--gpus", "3", "--ordered-sum", "--share-devices"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cli.append(read_png(os.path.join(tmp, "three.png")).copy())
py = []
for k in range(8):
    r = Renderer(SceneLoader().loadText(text), options={ffi.OPT_SEMANTICS: 1, ffi.OPT_SEEDING: 1})
    buf = torch.zeros(80 * 96 * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    r.bind_sum_buffer(buf.data_ptr(), buf.numel() * 4)
    total = None
    for a, b in ((0, 2), (2, 4), (4, 6)):
        r.render_range(a, b, clear=True)
        r.synchronize()
        total = buf.clone() if total is None else total + buf
        torch.cuda.synchronize()
    buf.copy_(total)
    torch.cuda.synchronize()
    r.resolve(6)
    res = r.result()
    r.close()
    want = np.empty((80, 96, 4), dtype=np.uint8)
    ffi.host_lib().pth_result_to_rgba8(res.ctypes.data, 96, 80, want.ctypes.data)
    py.append(want)
for name, imgs in (("cli", cli), ("py", py)):
    print(name, [int((im != imgs[0]).any(axis=2).sum()) for im in imgs])
print("cli0 vs py0", int((cli[0] != py[0]).any(axis=2).sum()))
bad = [im for im in cli + py if (im != py[0]).any()]
if bad:
    d = (bad[0] != py[0]).any(axis=2)
    ys, xs = np.nonzero(d)
    print("differing pixels (y, x):", list(zip(ys.tolist(), xs.tolist()))[:40])
