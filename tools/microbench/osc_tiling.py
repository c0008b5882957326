#!/usr/bin/env python3
"""Oscillator forward per harmonics-per-lane tiling K (ddsp_osc_set_tiling) at the BASELINE.json shapes: device ms of the
totals and synth kernels from the library's HIP-event hooks.  usage: osc_tiling.py [cfg3|cfg4]"""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")  # kernel-form / tiling hooks (include/ddsp_hip.h)
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


def run(shape, K, reps=5):
    L = ddsp._lib.lib()
    ctl = syn.make_controls(shape, 1003, "all_live")
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items() if k != "H"}
    if L.ddsp_osc_set_tiling(K) != 0:
        return None
    try:
        try:
            for _ in range(2):
                ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
        except ddsp._lib.DdspHipError:
            return None
        ddsp._lib.profile_enable(8 * reps + 8)
        torch.cuda.synchronize()
        for _ in range(reps):
            ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
        torch.cuda.synchronize()
        rec = {}
        for n, ms in ddsp._lib.profile_read():
            rec.setdefault(n, []).append(ms)
        ddsp._lib.profile_enable(0)
    finally:
        L.ddsp_osc_set_tiling(0)
    return {k: round(float(np.mean(v)), 4) for k, v in rec.items()}


if __name__ == "__main__":
    shape = {"cfg3": syn.CFG3, "cfg4": syn.CFG4_PER_GPU}[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
    out = {"shape": shape.name}
    for K in (0, 8, 12, 13, 15, 16, 20, 23, 25):
        r = run(shape, K)
        if r:
            out[f"K{K}"] = r
    print(json.dumps(out))
