#!/usr/bin/env python3
"""Per-phase cycle counts of the wavefront-private hop-128 noise kernel (a -DDDSP_NOISE_STAMPS build of ddsp_noise_wave.hip:
tools/build_variant.sh ddsp_noise_wave.hip stamps -DDDSP_NOISE_STAMPS; run with DDSP_HIP_LIB=.../libddsp_hip_stamps.so).
The stamp buffer travels as the injected draw (which that build ignores).  Prints the mean cycles per group and phase."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

B, T = 512, 500
rng = np.random.default_rng(1)
H = torch.from_numpy(syn.controller_range(rng.standard_normal((B, T, 65), dtype=np.float32))).cuda()
y = torch.zeros(B, T * 128, device="cuda")
buf = torch.zeros(B, T, 128, device="cuda")                     # reinterpreted by the kernel as uint64 [grid][8]
for acc in (True, False):
    for _ in range(3):
        buf.zero_()
        ddsp.noise_forward(H, 128, uniform=buf, out=y, accumulate=acc)
        torch.cuda.synchronize()
    st = buf.view(torch.int64).reshape(-1)[:2048 * 8].reshape(2048, 8).cpu().numpy().astype(np.float64)
    groups = np.full(2048, B * T / 16 / 2048)                    # groups per wavefront (7 or 8)
    names = ["setup+first group's impulse responses and draw", "convolution pass 0", "convolution pass 1 (+ y read issue)", "staging",
             "H->LDS + operand reads + products", "pending output (add + store)", "taps", "noise draw (+ next H tile issue)"]
    per_group = {n: round(float(st[:, i].sum() / groups.sum()), 1) for i, n in enumerate(names)}
    per_group[names[0]] = round(float(st[:, 0].mean()), 1)       # once per wavefront
    total = float(st.sum(axis=1).mean())
    print(json.dumps({"accumulate": acc, "cycles_per_group": per_group, "mean_cycles_per_wavefront": round(total),
                      "groups_per_wavefront": round(float(groups.mean()), 2), "memtime_clock": "100 MHz x ? (s_memtime ticks)"}))
