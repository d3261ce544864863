#!/usr/bin/env python3
"""Static instruction mix of one kernel in a device assembly listing (hipcc --cuda-device-only -S).
usage: isa_mix.py <file.s> <symbol prefix>   e.g.  isa_mix.py wavefront.s _Z9k_surfaceILi3ELb1EE"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
ops = collections.Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.split()[0].endswith(":"):
        continue
    ops[l.split()[0]] += 1
cls = collections.Counter()
for k, v in ops.items():
    c = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "vmem" if k.startswith(("global_", "flat_", "buffer_", "scratch_")) else "lds" if k.startswith("ds_") else "other"
    cls[c] += v
print("total", sum(ops.values()), dict(cls))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for k, v in ops.most_common(n):
    print(f"{v:6d} {k}")
