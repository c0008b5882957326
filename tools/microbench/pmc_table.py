#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: pmc_table.py <dir with pmc*_<label>/ subdirectories> [substring]"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "osc"
out = collections.defaultdict(dict)
for d in sorted(glob.glob(root + "/pmc*_*")):
    if d.endswith(".err"):
        continue
    label = d.split("_")[-1]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if want not in k:
                continue
            m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
            name = (m.group(1) + (m.group(2) or "")) if m else k[:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[label + " " + k].update({c: sum(x) / len(x) for c, x in v.items()})
for k, v in out.items():
    if v.get("SQ_INSTS_VALU", 1e9) < 1e6:
        continue
    print(k)
    print("   ", {c: round(x / 1e6, 2) for c, x in sorted(v.items())})
    if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        print("    cycles per VALU instruction per SIMD: %.3f   resident wave slots per SIMD: %.2f" % (
            v["GRBM_GUI_ACTIVE"] / 8 * 1024 / v["SQ_INSTS_VALU"], v["SQ_WAVE_CYCLES"] * 4 / v["GRBM_GUI_ACTIVE"] / 128))
