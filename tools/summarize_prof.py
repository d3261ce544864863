#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into one small
markdown summary (per-kernel calls / total / average time and register, LDS and
scratch use).  Usage: summarize_prof.py <dir> <out.md> [title]"""
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(d)
    stats = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)
    trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    res = {}
    for t in trace:
        for r in csv.DictReader(open(t)):
            res.setdefault(r["Kernel_Name"], (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]))
    lines = [f"# {title}", "", "(vgpr as rocprofv3 reports it on gfx950: half of the compiler's NumVgprs, e.g. 40 for the 80 registers of k_trace_deferred)", "", "| kernel | calls | total ms | avg us | % | vgpr | sgpr | lds B | scratch B |", "|---|---|---|---|---|---|---|---|---|"]
    for s in stats:
        for r in csv.DictReader(open(s)):
            name = r["Name"]
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if len(short) > 70:
                short = short[:67] + "..."
            v = res.get(name, ("?",) * 5)
            lines.append(f"| `{short}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} | {v[0]} | {v[2]} | {v[3]} | {v[4]} |")
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
