"""Single-function and single-segment probes (tests/golden/fixtures.npz, made by make_fixtures.py).

  shade_*   the reference's shading helpers on 512 random inputs each: phongBRDF, ggxBRDF, ggxPDF
            (RayTracer.cu:259-294) and the sampled directions of the Phong / GGX branches (:232-257,
            302-322, 675-696, by-reference helpers).  The HIP versions replace libm chains by algebra
            (powf -> exp2(y*log2 x); tanf(acosf(x))^2 -> (1-x)(1+x)/x^2; powf(cosf(acosf(x)),4) -> x^4;
            powf(1-x,5) -> products; sin/cos(2*pi*zeta) -> sincospi(2*zeta); sin/cos(atanf(a)) ->
            a/sqrt(1+a^2), 1/sqrt(1+a^2)); the bounds below state what that costs per function.
  probe_*   one segment each: (ray, throughput, depth, seed) -> payload after rtTrace + miss /
            closest-hit program, per integrator x sampling method x BRDF x semantics, two generations
            (camera rays, then the segments the oracle continues them with).

The oracle must reproduce its own vectors (CPU); the HIP path is held to per-field bounds (GPU)."""
import json
import os

import numpy as np
import pytest

from test_golden_fixtures import FIX, mk
from util import ROOT

# |gpu - oracle| <= ATOL + RTOL * |oracle| per component, for every input (with margin over the measured maxima:
# the test writes them to gpurun_out/probe_errors.json)
SHADE_BOUNDS = {0: (2e-7, 2e-6), 1: (1e-6, 1e-5), 2: (5e-7, 2e-6), 3: (4e-6, 0.0), 4: (4e-6, 0.0)}
SHADE_NAMES = {0: "phongBRDF", 1: "ggxBRDF", 2: "ggxPDF", 3: "Phong-sampled direction", 4: "GGX-sampled direction"}


@pytest.mark.parametrize("fn", range(5))
def test_oracle_shading_vectors(fn):
    a = mk.shading_inputs(fn)
    assert np.array_equal(a, FIX[f"shade_in_{fn}"])
    got = mk.orc.eval_shading(fn, a)
    assert np.allclose(got, FIX[f"shade_out_{fn}"], rtol=1e-5, atol=1e-7, equal_nan=True)
    assert np.isfinite(got).all()


def fields_equal(a, b, rtol, atol):
    ok = np.ones(len(a), dtype=bool)
    for f in ("radiance", "thr", "o", "d"):
        ok &= np.isclose(a[f], b[f], rtol=rtol, atol=atol, equal_nan=True).all(axis=1)
    return ok & (a["alive"] == b["alive"]) & (a["depth"] == b["depth"]) & (a["prim"] == b["prim"])


@pytest.mark.parametrize("name", list(mk.PROBE_CASES))
def test_oracle_probe_vectors(name):
    scene, opts, probes = mk.probe_inputs(name)
    assert probes.tobytes() == FIX[f"probe_in_{name}"].tobytes()
    got = mk.orc.probe_segments(scene, probes, **opts)
    want = FIX[f"probe_out_{name}"]
    assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
    assert fields_equal(got, want, 1e-5, 1e-7).all()


def record(key, value):
    """Measured errors go to gpurun_out/probe_errors.json when that directory exists."""
    d = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(d):
        return
    path = os.path.join(d, "probe_errors.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = value
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.gpu
@pytest.mark.parametrize("fn", range(5))
def test_gpu_shading_functions(fn):
    from pathtracer_amd import Renderer, SceneLoader
    r = Renderer(SceneLoader().loadText("size 8 8\n"))
    got = r.eval_shading(fn, FIX[f"shade_in_{fn}"])
    r.close()
    want = FIX[f"shade_out_{fn}"]
    atol, rtol = SHADE_BOUNDS[fn]
    diff = np.abs(got - want)
    rel = diff / np.maximum(np.abs(want), 1e-30)
    big = np.abs(want) > 1e-4
    record(f"shade_{fn}_{SHADE_NAMES[fn]}", {"max_abs": float(diff.max()), "max_rel_where_above_1e-4": float(rel[big].max()) if big.any() else 0.0,
                                             "p99_rel_where_above_1e-4": float(np.percentile(rel[big], 99)) if big.any() else 0.0})
    assert (diff <= atol + rtol * np.abs(want)).all(), (float(diff.max()), float(rel[big].max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mk.PROBE_CASES))
def test_gpu_probe_segments(name):
    """Per field: the closest hit (primitive, t) and with it the next origin are bit-exact; depth and
    the alive flag equal; radiance, throughput and direction within 1e-5 + 1e-3 relative on >= 98 % of
    the probes (a light sample's visibility or a roulette draw can flip on an ulp) and their medians
    within 2e-6 relative."""
    from pathtracer_amd import Renderer, _ffi
    scene, opts, probes = mk.load_probe_case(name) + (FIX[f"probe_in_{name}"],)
    want = FIX[f"probe_out_{name}"]
    r = Renderer(scene, options={_ffi.OPT_SEMANTICS: opts.get("semantics", 0), _ffi.OPT_SEEDING: 1, _ffi.OPT_BOUNCE_CAP: opts["bounce_cap"]})
    got = r.probe_segments(probes)
    r.close()
    assert np.array_equal(got["prim"], want["prim"])
    assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
    ok = fields_equal(got, want, 1e-3, 1e-5)
    both = (got["alive"] != 0) & (want["alive"] != 0)
    stats = {"probes": int(len(want)), "agree": float(ok.mean()), "alive_equal": float((got["alive"] == want["alive"]).mean())}
    for f in ("radiance", "thr", "d", "o"):
        sel = both if f != "radiance" else np.ones(len(want), dtype=bool)
        a, b = got[f][sel], want[f][sel]
        fin = np.isfinite(a) & np.isfinite(b)
        rel = np.abs(a - b)[fin] / np.maximum(np.abs(b)[fin], 1e-3)
        stats[f] = {"median_rel": float(np.median(rel)) if rel.size else 0.0, "p99_rel": float(np.percentile(rel, 99)) if rel.size else 0.0,
                    "nan_equal": bool(np.array_equal(np.isnan(a), np.isnan(b)))}
    record(f"probe_{name}", stats)
    assert (got["alive"] == want["alive"]).mean() >= 0.99
    # literal GGX: dot(h, n) sits at 1 +- 1 ulp, NaN or not on the last bit of a cosine (see test_gpu_parity.py)
    assert ok.mean() >= (0.93 if name == "brdf_ggx_lit" else 0.98), stats
    if both.any():
        assert np.array_equal(got["o"][both & ok].view(np.uint32), want["o"][both & ok].view(np.uint32)) or name in ("glass_int", "volume_int")
    for f in ("radiance", "thr", "d"):
        assert stats[f]["median_rel"] <= 2e-6, (f, stats[f])
