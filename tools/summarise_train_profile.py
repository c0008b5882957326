#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of `bench.py --mode train` -> per-step table (calls, average ns, ms per step).
Usage: summarise_train_profile.py <rocprof output dir> <traced steps> "<command line traced>" > profiles/<name>.csv"""
import csv
import glob
import os
import sys

src, steps, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
path = glob.glob(os.path.join(src, "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"# {cmd} ({steps} steps traced); GPU-busy {total / steps / 1e6:.2f} ms per step, {calls // steps} launches per step")
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls_per_step", "avg_ns", "ms_per_step", "pct"])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t = float(r["TotalDurationNs"])
    w.writerow([r["Name"][:150], round(int(r["Calls"]) / steps, 2), int(float(r["AverageNs"])), f"{t / steps / 1e6:.3f}", f"{100 * t / total:.2f}"])
