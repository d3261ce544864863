#!/usr/bin/env python3
"""Per-ray hardware-counter figures of the two traversal kernel classes from the passes of
tools/pmc_passes.sh -> an entry of pmc_counters.json (which bench.py quotes in its roofline object) and
a markdown summary for profiles/.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane
reads (the access width of every load in these kernels), so it is doubled; WRITE_SIZE is taken as is.
Clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same dispatches (timestamps of the counter pass).
Active lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU).

usage: pmc_summary.py <pass dir> <workload> <semantics> <pmc_counters.json> <out.md> <source label>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd.srchash import device_source_hash  # noqa: E402


def counting_variant(name):
    """k_trace_deferred<ANY, COUNT, WIDE>: second argument; the other traversal kernels: <COUNT> or <COUNT, ...>."""
    if "<" not in name:
        return False
    args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
    if "k_trace_deferred" in name:
        return len(args) > 1 and args[1] == "true"
    return args[0] == "true"


def kernel_class(name):
    if counting_variant(name):  # run outside the timed region
        return None
    if "k_extend" in name or "k_trace_deferred<false" in name:
        return "extend"
    if "k_shadow" in name or "k_trace_deferred<true" in name:
        return "shadow"
    for k in ("k_surface", "k_gather", "k_generate", "k_compact_hits", "k_accumulate"):
        if k in name:
            return k
    return None


def bench_line(log):
    for line in open(log, errors="ignore"):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


def main():
    d, workload, semantics, out_json, out_md, source = sys.argv[1:7]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(dict))  # class -> counter -> dispatch -> (t0, t1)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = kernel_class(r["Kernel_Name"])
            if not key:
                continue
            c = r["Counter_Name"]
            tot[key][c] += float(r["Counter_Value"])
            t0, t1 = r.get("Start_Timestamp"), r.get("End_Timestamp")
            disp[key][c][r["Dispatch_Id"]] = (int(t0), int(t1)) if t0 and t1 else None
    line = None
    for log in sorted(glob.glob(os.path.join(d, "pass*.log"))):
        line = bench_line(log) or line
    if line is None:
        raise SystemExit("no bench line in the pass logs")
    rl = line["roofline"]
    rays = {"extend": rl["rays_per_launch"] * rl["launches"], "shadow": rl["shadow"]["rays"], "surface": rl.get("surface", {}).get("hits", 0)}
    unit = {"extend": "ray", "shadow": "ray", "surface": "hit"}
    # the counters belong to the kernels of THIS tree: bench.py compares the hash with the sources it runs on
    entry = {"workload": line["config"]["workload"], "semantics": int(semantics), "source": source, "code_hash": device_source_hash(),
             "how": "rocprofv3 --pmc, one pass per counter group (tools/pmc_passes.sh), bench.py --steps 1 --warmup 0; FETCH_SIZE doubled (gfx950, 16-B-per-lane reads); "
                    "clock = GRBM_GUI_ACTIVE / 8 / kernel time of the counter pass"}
    md = [f"# {source}: {line['config']['workload']} ({line['config']['semantics']})", "",
          "Sums over the dispatches of the timed step (`bench.py --steps 1 --warmup 0`), counting instantiations excluded.", ""]
    if "k_surface" in tot:
        tot["surface"], disp["surface"] = tot["k_surface"], disp["k_surface"]
    for key in ("extend", "shadow", "surface"):
        t = tot.get(key)
        if not t or "SQ_INSTS_VALU" not in t or not rays[key]:
            continue
        u = unit[key]
        n = len(disp[key]["SQ_INSTS_VALU"])
        spans = [v for v in disp[key]["GRBM_GUI_ACTIVE"].values() if v]
        ksec = sum(b - a for a, b in spans) * 1e-9 if spans else None
        clock = t["GRBM_GUI_ACTIVE"] / 8.0 / ksec / 1e9 if ksec else None
        fetch = t.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
        write = t.get("WRITE_SIZE", 0.0) * 1024.0
        e = {"launches": n, "rays": rays[key],
             f"valu_insts_per_{u}": t["SQ_INSTS_VALU"] / rays[key],
             "salu_per_valu": t["SQ_INSTS_SALU"] / t["SQ_INSTS_VALU"],
             "active_lane_frac": t["SQ_THREAD_CYCLES_VALU"] / (64.0 * t["SQ_ACTIVE_INST_VALU"]),
             f"hbm_bytes_per_{u}": (fetch + write) / rays[key], "fetch_bytes_per_launch": fetch / max(n, 1), "write_bytes_per_launch": write / max(n, 1),
             "clock_ghz": clock, "kernel_ms_in_counter_pass": ksec * 1e3 if ksec else None,
             "valu_insts_per_launch": t["SQ_INSTS_VALU"] / max(n, 1)}
        entry[key] = e
        md += [f"## {key} {'traversal' if u == 'ray' else 'kernel'} ({n} launches, {rays[key]:.4g} {u}s)"]
        for c in sorted(t):
            md.append(f"- {c}: {t[c]:.6g}")
        md += ["", f"- VALU wave instructions per {u}: {e[f'valu_insts_per_{u}']:.1f}; SALU per VALU: {e['salu_per_valu']:.3f}; active lanes: {e['active_lane_frac']:.3f}",
               f"- HBM bytes per {u} (2 x FETCH_SIZE + WRITE_SIZE): {e[f'hbm_bytes_per_{u}']:.1f}; per launch: fetch {e['fetch_bytes_per_launch']:.4g}, write {e['write_bytes_per_launch']:.4g}"]
        if clock:
            rate = t["SQ_INSTS_VALU"] / ksec
            peak = 256 * 4 * clock * 1e9 / 2.0
            md += [f"- kernel time in the counter pass: {ksec*1e3:.2f} ms; clock {clock:.3f} GHz; VALU issue {rate/1e9:.1f} G wave-instructions/s = {rate/peak:.3f} of 256 CUs x 4 SIMDs x clock / 2 cycles",
                   f"- HBM rate in the counter pass: {(fetch+write)/ksec/1e9:.0f} GB/s = {(fetch+write)/ksec/8e12:.3f} of 8 TB/s"]
        md.append("")
    for key in sorted(tot):
        if key in ("extend", "shadow", "surface"):
            continue
        t = tot[key]
        md += [f"## {key}"] + [f"- {c}: {t[c]:.6g} ({len(disp[key][c])} dispatches)" for c in sorted(t)] + [""]
    data = {}
    if os.path.exists(out_json):
        try:
            data = json.load(open(out_json))
        except ValueError:
            data = {}
    data[f"{workload}/semantics{semantics}"] = entry
    json.dump(data, open(out_json, "w"), indent=1, sort_keys=True)
    open(out_md, "w").write("\n".join(md) + "\n")
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
