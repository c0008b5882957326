#!/usr/bin/env python3
"""BASELINE.json configs[2] (batch 512, 48 kHz, 200 harmonics, hop 512) under every harmonics-per-lane tiling the oscillator kernels
are built for: is the cost model's choice (ddsp_osc.hip: pick_tiling) the measured best?  One JSON line."""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("WORLD_SIZE", "1")
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

out = {}
L = ddsp._lib.lib()
for K in (0, 13, 15, 16, 20, 23, 25, 0):
    rc = L.ddsp_osc_set_tiling(K)
    if rc != 0:
        out[str(K)] = f"rc {rc}"
        continue
    try:
        r = bench.time_config(syn.CFG3, 1003, 5, 2)
        out[f"K{K}" if f"K{K}" not in out else f"K{K}_again"] = {"ms_per_step": round(r["ms_per_step"], 3), **{k: round(v, 3) for k, v in r["kernel_ms"].items()}}
    except Exception as e:  # noqa: BLE001
        out[f"K{K}"] = repr(e)[:200]
L.ddsp_osc_set_tiling(0)
print(json.dumps(out))
