#!/usr/bin/env python3
"""Per-wavefront start / end of the chunked synth kernel (library built with -DDDSP_CHUNK_STAMPS; DDSP_HIP_LIB points at it)."""
import collections
import ctypes
import json
import os
import sys

import numpy as np
import torch

os.environ.setdefault("DDSP_TEST_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

_name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
if _name.startswith("s") and "x" in _name:   # s<rows>x<frames>x<hop>x<harmonics> at 48 kHz
    _b, _t, _h, _k = (int(v) for v in _name[1:].split("x"))
    shape = syn.SynthShape(_name, _b, 48000, _h, _t, _k, 65)
else:
    shape = {"cfg4": syn.CFG4_PER_GPU, "cfg2": syn.CFG2, "cfg3": syn.CFG3}[_name]
ctl = syn.make_controls(shape, 1004, "all_live")
x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items() if k != "H"}
plan = ddsp._lib.osc_plan(shape.batch, shape.frames, shape.n_harmonics, shape.hop, shape.sample_rate)
for _ in range(3):
    ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
torch.cuda.synchronize()
n = plan["chunks_per_row"] * plan["row_blocks"]
L = ddsp._lib.lib()
buf = (ctypes.c_long * (4 * n))()
L.ddsp_osc_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.ddsp_osc_read_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(n, 4)
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 100.0   # microseconds (100 MHz)
end = (a[:, 1] - t0) / 100.0
dur = end - start
hw = a[:, 2]
xcc = a[:, 3] & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
simd = (hw >> 4) & 3
wslot = hw & 0xf
print("wave slots used:", collections.Counter(int(v) for v in wslot))
print(json.dumps({"plan": plan, "tasks": int(n), "start_us": [float(start.min()), float(np.median(start)), float(start.max())],
                  "end_us": [float(end.min()), float(np.median(end)), float(end.max())],
                  "dur_us": [float(dur.min()), float(np.median(dur)), float(dur.max())]}))
print("duration histogram (us):", np.histogram(dur, bins=12))
print("start histogram (us):", np.histogram(start, bins=12))
per = collections.defaultdict(list)
for i in range(n):
    per[int(xcc[i])].append(end[i])
print("per XCC: n, median end, max end:", {k: (len(v), round(float(np.median(v)), 1), round(float(max(v)), 1)) for k, v in sorted(per.items())})
slot = collections.Counter((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i])) for i in range(n))
print("waves per (xcc,se,sh,cu,simd): histogram of counts:", collections.Counter(slot.values()), "distinct SIMDs:", len(slot))
cus = collections.Counter((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(n))
print("waves per CU: histogram:", collections.Counter(cus.values()), "distinct CUs:", len(cus))
for w in sorted(set(int(v) for v in wslot)):
    print("wave slot", w, "n", int((wslot == w).sum()), "median end", round(float(np.median(end[wslot == w])), 1))
full = dur > 0.9 * np.median(dur)                      # (drop the shorter last chunks of every row)
for name, key in (("SIMD id", simd), ("shader array", sh), ("shader engine", se), ("CU id", cu), ("XCC", xcc)):
    print("median / max end by", name, {int(v): (round(float(np.median(end[full & (key == v)])), 1), round(float(end[full & (key == v)].max()), 1))
                                        for v in sorted(set(int(q) for q in key))})
simd_end = collections.defaultdict(float)
for i in range(n):
    kk = (int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]))
    simd_end[kk] = max(simd_end[kk], float(end[i]))
v = np.array(list(simd_end.values()))
print("last end per SIMD: min %.1f  5%% %.1f  median %.1f  95%% %.1f  max %.1f" % (v.min(), np.percentile(v, 5), np.median(v), np.percentile(v, 95), v.max()))
cu_end = collections.defaultdict(list)
for kk, e in simd_end.items():
    cu_end[kk[:4]].append(e)
spread = np.array([max(x) - min(x) for x in cu_end.values()])
print("spread of the four SIMDs' last ends inside a CU: median %.1f  max %.1f us; between CU means: std %.1f us" % (
    np.median(spread), spread.max(), float(np.std([np.mean(x) for x in cu_end.values()]))))
# duration against co-residency
cnt = np.array([slot[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]))] for i in range(n)])
for c in sorted(set(cnt)):
    print("waves on a SIMD with", c, "tasks: n", int((cnt == c).sum()), "median dur", round(float(np.median(dur[cnt == c])), 1),
          "median start", round(float(np.median(start[cnt == c])), 1))
# finishing order inside a SIMD (three co-resident tasks)
order = collections.defaultdict(list)
for i in range(n):
    if full[i]:
        order[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]))].append((float(end[i]), float(start[i])))
trip = np.array([[e for e, _ in sorted(v)] for v in order.values() if len(v) == 3])
if len(trip):
    print("SIMDs with three full tasks: %d; median end of the first / second / last to finish: %.1f / %.1f / %.1f us; mean %.1f / %.1f / %.1f" % (
        len(trip), *np.median(trip, axis=0), *trip.mean(axis=0)))
    print("work rate of a SIMD, samples per us: 3*Lc/last end = %.3f (median)" % float(np.median(3 * plan["chunk_samples"] / trip[:, 2])))
