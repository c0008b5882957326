#!/usr/bin/env python3
"""Device time of the oscillator / noise backward kernels (torch profiler), per f0 kind: `all_live` (no harmonic above
Nyquist) and `musical` (CREPE-grid f0: many masked harmonics).  A/B two builds with DDSP_HIP_LIB=<path to .so>."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402


class Conf:
    n_harmonics, sample_rate, hop_length = 100, 16000, 128


def run(kind, batch):
    shape = syn.SynthShape("b", batch, 16000, 128, 500, 100, 65)
    ctl = syn.make_controls(shape, 1, kind)
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
    c = x["c"].clone().requires_grad_()
    a = x["a"].clone().requires_grad_()
    H = x["H"].clone().requires_grad_()
    osc = ddsp.OscillatorBank(Conf).cuda()
    noise = ddsp.FilteredNoise(Conf, rng="device")

    def step():
        y = osc({"f0": x["f0"], "c": c, "a": a}) + noise({"H": H})
        y.square().mean().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    out = {}
    for e in prof.key_averages():
        for name in ("osc_bwd_kernel", "osc_bwd_finish_kernel", "noise_bwd_batched_kernel", "osc_synth_kernel", "osc_totals_kernel"):
            if name in e.key:
                out[name] = out.get(name, 0.0) + e.device_time_total / 10 / 1e3
    return {k: round(v, 4) for k, v in out.items()}


if __name__ == "__main__":
    res = {"lib": os.path.basename(os.environ.get("DDSP_HIP_LIB", "libddsp_hip.so"))}
    for kind in ("all_live", "musical"):
        for batch in (32, 256):
            res[f"{kind}_b{batch}_ms"] = run(kind, batch)
    print(json.dumps(res))
