#!/usr/bin/env python3
"""Per-launch duration of the oscillator's kernels from an idle GPU, alone and with the noise kernel between launches, and the shader
clock of the LAST synth launch of each series (profiles/r04_clock_ramp.txt).
    python tools/microbench/clock_ramp.py"""
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("DDSP_TEST_HOOKS", "1")
sys.path.insert(0, os.getcwd())
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

IDX = [0, 1, 2, 3, 5, 8, 12, 20, 30, 50, 80, 119, 199, 299]


def series(name, shape, steps, with_noise):
    ctl = syn.make_controls(shape, 1004, "all_live")
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}

    def run(i):
        y, _, _, scratch = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate, return_scratch=True,
                                            keep_frame_scratch=False)
        if with_noise == "write_only":       # the noise kernel writes its own buffer and reads nothing back
            ddsp.noise_forward(x["H"], shape.hop, seed=7, offset=i << 32, out=y2, accumulate=False)
        elif with_noise and i % EVERY == 0:
            ddsp.noise_forward(x["H"], shape.hop, seed=7, offset=i << 32, out=y, accumulate=True)
        return scratch

    y2 = torch.empty(shape.batch, shape.frames * shape.hop, device="cuda")

    run(0)
    torch.cuda.synchronize()
    time.sleep(0.5)
    ddsp._lib.profile_enable(8 * steps + 8)
    for i in range(steps):
        scratch = run(i)
    ghz = ddsp._lib.osc_clock(scratch, shape.batch, shape.frames, shape.n_harmonics, shape.hop, shape.sample_rate,
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rec = {}
    for k, ms in ddsp._lib.profile_read():
        rec.setdefault(k, []).append(ms)
    ddsp._lib.profile_enable(0)
    s, t = np.array(rec["osc_frame_synth"]), np.array(rec["osc_frame_totals"])
    tag = f"{name} {'osc + noise' if with_noise else 'osc only'}"
    print(tag, "synth ms at launch #:", {i: round(float(s[i]), 4) for i in IDX if i < len(s)})
    print(tag, "totals ms:", {i: round(float(t[i]), 4) for i in IDX if i < len(t)})
    hs = shape.batch * shape.frames * shape.hop * shape.n_harmonics
    print(tag, "synth mean of first 12 %.4f, of last 12 %.4f ms = %.4f ps per harmonic-sample; clock of the last synth launch %.3f GHz"
          % (s[:12].mean(), s[-12:].mean(), s[-12:].mean() * 1e9 / hs, ghz), flush=True)


EVERY = int(os.environ.get("NOISE_EVERY", "1"))
if len(sys.argv) > 1 and sys.argv[1] == "osc":
    series("headline", syn.CFG4_PER_GPU, 120, False)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "write_only":
    series("headline", syn.CFG4_PER_GPU, 120, "write_only")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "noise":    # only the headline with the noise kernel (A-B builds of that kernel)
    series("headline", syn.CFG4_PER_GPU, 120, True)
    sys.exit(0)
for rnd in range(2):
    for noise in (False, True):
        series("headline", syn.CFG4_PER_GPU, 200, noise)
series("long", syn.SynthShape("l", 512, 16000, 128, 1500, 100, 65), 100, False)
for noise in (False, True):
    series("cfg3", syn.CFG3, 40, noise)
