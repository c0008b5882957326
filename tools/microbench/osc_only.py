#!/usr/bin/env python3
"""Only the oscillator at a BASELINE shape, N times (for rocprofv3 passes):
osc_only.py <frame|chunk|chunk_any> [steps] [cfg1|cfg2|cfg3|cfg4|b<rows>] [all_live|musical]"""
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

path = sys.argv[1] if len(sys.argv) > 1 else "chunk"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
_name = sys.argv[3] if len(sys.argv) > 3 else "cfg4"
if _name.startswith("s") and "x" in _name:   # s<rows>x<frames>x<hop>x<harmonics>: any shape (48 kHz so that every harmonic is audible)
    _b, _t, _h, _k = (int(v) for v in _name[1:].split("x"))
    shape = syn.SynthShape(_name, _b, 48000, _h, _t, _k, 65)
elif _name.startswith("hop"):    # hop<n>: the headline shape (512 rows, 100 harmonics, 64 000 samples) at another hop
    shape = syn.SynthShape(_name, 512, 16000, int(_name[3:]), 64000 // int(_name[3:]), 100, 65)
elif _name.startswith("b"):    # b<rows>: the headline shape with that many rows and 8x longer clips
    shape = syn.SynthShape(_name, int(_name[1:]), 16000, 128, 4000, 100, 65)
else:
    shape = {"cfg1": syn.CFG1, "cfg4": syn.CFG4_PER_GPU, "cfg2": syn.CFG2, "cfg3": syn.CFG3}[_name]
kind = sys.argv[4] if len(sys.argv) > 4 else "all_live"
assert ddsp._lib.lib().ddsp_osc_set_path({"frame": 1, "chunk": 0, "chunk_any": 2}[path]) == 0
ctl = syn.make_controls(shape, 1004, kind)
x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items() if k != "H"}
plan = ddsp._lib.osc_plan(shape.batch, shape.frames, shape.n_harmonics, shape.hop, shape.sample_rate)


def run():
    return ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)[0]


for _ in range(2):
    run()
ddsp._lib.profile_enable(8 * steps + 8)
torch.cuda.synchronize()
for _ in range(steps):
    y = run()
torch.cuda.synchronize()
rec = {}
for name, ms in ddsp._lib.profile_read():
    rec.setdefault(name, []).append(ms)
print(json.dumps({"path": path, "config": shape.name, "f0": kind, "plan": plan, "env_chunk": os.environ.get("DDSP_OSC_CHUNK_LEN"),
                  "kernel_ms": {k: round(float(np.mean(v)), 4) for k, v in rec.items()}}), flush=True)
