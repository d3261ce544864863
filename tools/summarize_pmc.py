#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel.  usage: summarize_pmc.py <dir> <out.md> [title]"""
import collections
import csv
import glob
import os
import sys


def short(name):
    s = name.split("(")[0].replace("void ", "")
    return s if len(s) <= 60 else s[:57] + "..."


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    counters = sorted({c for k in tot for c in tot[k]})
    lines = [f"# {title}", "", "Sums over all dispatches of each kernel (dispatch count in parentheses).", ""]
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("FETCH_SIZE", 0))):
        if not any(x in k for x in ("k_extend", "k_trace", "k_shadow", "k_connect", "k_shade", "k_surface", "k_gather", "k_generate", "k_accumulate", "k_compact")):
            continue
        lines.append(f"## `{k}`")
        for c in counters:
            if c in tot[k]:
                lines.append(f"- {c}: {tot[k][c]:.6g} ({len(calls[k][c])} dispatches)")
        lines.append("")
    open(out, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
