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


def test_png_compresses_and_decodes_identically(tmp_path):
    """Adaptive filters + LZ77 + dynamic Huffman codes: smooth and flat images shrink a lot, noise does not
    grow by more than the container overhead, and every image decodes to the bytes that went in (checked with
    zlib through the test's own PNG reader)."""
    import os
    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:96, 0:160]
    smooth = np.stack([(x * 255 // 159), (y * 255 // 95), ((x + y) * 255 // 254), np.full_like(x, 255)], axis=-1).astype(np.uint8)
    flat = np.zeros((300, 500, 4), dtype=np.uint8)
    flat[..., 3] = 255
    flat[100:200, 100:300, :3] = (10, 200, 30)
    noise = rng.integers(0, 256, size=(64, 64, 4), dtype=np.uint8)
    render_like = np.clip(smooth.astype(int) + rng.integers(-6, 7, size=smooth.shape), 0, 255).astype(np.uint8)
    render_like[..., 3] = 255
    skewed = (rng.exponential(0.7, size=(80, 120, 4)).astype(int) % 256).astype(np.uint8)  # a few very frequent values: long and short codes
    tiny = np.array([[[1, 2, 3, 255]]], dtype=np.uint8)
    for name, img, max_ratio in (("smooth", smooth, 0.15), ("flat", flat, 0.01), ("noise", noise, 1.02), ("render_like", render_like, 0.75),
                                 ("skewed", skewed, 0.5), ("tiny", tiny, 50.0)):
        path = str(tmp_path / (name + ".png"))
        resultToPNG(path, img.shape[1], img.shape[0], img.tobytes())
        assert np.array_equal(read_png(path), img), name
        assert os.path.getsize(path) <= max_ratio * img.nbytes + 100, (name, os.path.getsize(path), img.nbytes)
