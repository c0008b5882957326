#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/collect_profiles.sh into small, committed summaries:
   <dir>/<round>_kernel_stats.csv      per-kernel calls / average ns (from --kernel-trace --stats)
   <dir>/<round>_pmc.json              per-kernel averages of the PMC counters + derived HBM bytes per launch
   <dir>/<round>_bench.json            the bench.py line of the same run
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of coalesced streaming reads -> doubled (calibrated here on osc_totals_kernel, whose only large
read is the [B,T,H] amplitude tensor: 2 x FETCH_SIZE = its byte count); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys

src, rnd = sys.argv[1], sys.argv[2]
dst = src
KEYS = ("osc_chunk_synth_kernel", "osc_chunk_totals_kernel", "osc_chunk_scan_kernel", "osc_synth_kernel", "osc_totals_kernel", "osc_supscan_kernel", "noise_wave_kernel", "noise_fft_kernel", "noise_batched_kernel", "noise_frame_kernel")


def short(name):
    for k in KEYS:
        if k in name:
            tail = name[name.index(k):]
            return tail.split("(")[0]
    return None


# the launches of bench.py's timed region inside the traced run (bench.py: clock_settle.timed_launches): the run also holds the
# from-idle pass, the settle steps and the clock probe's launches, so the all-launch average is not the timed region's
timed = {}
try:
    t = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])["clock_settle"]["timed_launches"]
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s:
                per[s].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for s, v in per.items():
        v.sort()
        sel = [d for _, d in v[t["first"]:t["first"] + t["count"]]]
        if len(sel) == t["count"]:
            timed[s] = sum(sel) / len(sel)
except Exception as e:  # noqa: BLE001
    print("no timed-region split:", e)

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        s = short(r["Name"])
        if s:
            rows.append({"kernel": s, "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"]),
                         "max_ns": int(r["MaxNs"]), "pct": float(r["Percentage"]),
                         "timed_region_avg_ns": round(timed[s], 1) if s in timed else ""})
    with open(os.path.join(dst, f"{rnd}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s:
                pmc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in pmc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_read_bytes_per_launch"] = 2.0 * d["FETCH_SIZE"] * 1024.0
        d["hbm_write_bytes_per_launch"] = d["WRITE_SIZE"] * 1024.0
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_VALU"):
        d["cycles_per_valu_inst_per_simd"] = (d["GRBM_GUI_ACTIVE"] / 8.0) / (d["SQ_INSTS_VALU"] / 1024.0)
    if d.get("SQ_WAVE_CYCLES") and d.get("SQ_WAIT_ANY") is not None:
        d["wait_any_share_of_wave_cycles"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    out[k] = d
# stamp: the kernel sources these counters were measured on (bench.py drops `roofline.traffic` when HEAD's sources differ)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import subprocess
    import bench
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out["_meta"] = {"kernel_sources_sha": bench.kernel_sources_stamp(), "round": rnd, "git_head": head or None}
except Exception as e:  # noqa: BLE001
    out["_meta"] = {"error": str(e)}
json.dump(out, open(os.path.join(dst, f"{rnd}_pmc.json"), "w"), indent=1, sort_keys=True)
try:
    line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{rnd}_bench.json"), "w"), indent=1)
    print(line[:400])
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)
for r in rows:
    print(r)
for k, d in out.items():
    print(k, {c: (f"{v:.4g}" if isinstance(v, float) else v) for c, v in d.items()})
