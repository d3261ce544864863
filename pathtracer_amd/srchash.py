"""Hash of the device sources the core library is built from.

Hardware-counter figures (instructions and HBM bytes per ray) are collected in rocprofv3 passes of their own and
quoted by later benchmark runs; the hash stored with them says which kernels they were measured on, so that a run
on changed kernels can tell that its quoted figures are stale."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_source_hash() -> str:
    files = sorted(glob.glob(os.path.join(ROOT, "pathtracer_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pathtracer_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "ptcore.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]
