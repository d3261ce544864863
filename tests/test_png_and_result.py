"""Renderer::getResult's conversion (Renderer.cpp:79-101) and the PNG writer that
stands in for lodepng::encode (utils.cpp:3-14)."""
import numpy as np

from pathtracer_amd import _ffi, resultToPNG
from oracle import oracle_py as orc
from util import read_png


def to_rgba8(res):
    out = np.empty(res.shape[:2] + (4,), dtype=np.uint8)
    _ffi.host_lib().pth_result_to_rgba8(res.ctypes.data, res.shape[1], res.shape[0], out.ctypes.data)
    return out


def test_result_to_rgba8_matches_the_oracle():
    rng = np.random.default_rng(3)
    res = rng.uniform(-0.2, 1.3, size=(37, 53, 3)).astype(np.float32)
    res[0, 0] = [0.0, 1.0, 0.999999]
    res[1, 1] = [1.0 / 255, 2.0 / 255 - 1e-7, 0.5]
    got, want = to_rgba8(res), orc.to_rgba8(res)
    assert np.array_equal(got, want)
    assert (got[..., 3] == 255).all()
    # truncation, clamping and the vertical flip
    assert tuple(got[-1, 0, :3]) == (0, 255, 254)
    assert got[0, 0, 0] == int(min(max(res[-1, 0, 0], 0), 1) * 255)


def test_png_round_trip(tmp_path):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(70, 300, 4), dtype=np.uint8)  # > 65535 bytes: several deflate blocks
    path = str(tmp_path / "a.png")
    resultToPNG(path, 300, 70, img.tobytes())
    back = read_png(path)
    assert np.array_equal(back, img)
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.array(Image.open(path)), img)
