#!/usr/bin/env python3
"""Reverb (reverb.py:24-49) on MI355X: the HIP paths of ddsp-pytorch_amd/reverb.py against the same module's stock
torch.fft formulation on the device (what round 1 shipped).  Device time per call from CUDA events over many calls.

  live     one rt callback: 2048 new samples against the one-second history at 44.1 kHz (config/default.py)
  offline  the training shape: 32 clips of 4 s at 16 kHz, forward and forward + backward
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd.reverb import causal_fft_convolve  # noqa: E402
import torch.nn.functional as F  # noqa: E402


class Conf:
    def __init__(self, sr):
        self.sample_rate, self.n_harmonics, self.hop_length = sr, 1, 64


def timed(fn, reps=200, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


def stock_impulse(rv):
    envelope = torch.exp(-F.softplus(-rv.decay) * rv.t * 500)
    taps = rv.noise * envelope * torch.sigmoid(rv.wet)
    return torch.cat([torch.ones_like(taps[:, :1]), taps[:, 1:]], dim=1)


def stock_forward(rv, x):
    n = x.shape[1]
    imp = stock_impulse(rv)
    imp = imp[:, :n] if n < rv.length else F.pad(imp, (0, n - rv.length))
    return causal_fft_convolve(x, imp)


def stock_live(rv, x):
    n = x.shape[1]
    window = torch.cat([rv.buffer[:, n:], x], dim=1)
    rv.buffer.data.copy_(window)
    return causal_fft_convolve(window, stock_impulse(rv))[:, -n:]


def main():
    out = {}
    rv = ddsp.Reverb(Conf(44100), initial_wet=0.5).cuda()
    x = torch.randn(1, 2048, device="cuda")
    with torch.no_grad():
        out["live_2048_of_44100_hip_ms"] = timed(lambda: rv.live_forward(x))
        out["live_2048_of_44100_stock_fft_ms"] = timed(lambda: stock_live(rv, x))
    rv = ddsp.Reverb(Conf(16000), initial_wet=0.5).cuda()
    x = torch.randn(32, 64000, device="cuda")
    with torch.no_grad():
        out["offline_32x64000_fwd_hip_ms"] = timed(lambda: rv(x), 50)
        out["offline_32x64000_fwd_stock_ms"] = timed(lambda: stock_forward(rv, x), 50)
    xg = x.clone().requires_grad_(True)

    def fb(f):
        for p in (rv.noise, rv.decay, rv.wet, xg):
            p.grad = None
        f(xg).square().mean().backward()

    out["offline_32x64000_fwd_bwd_hip_ms"] = timed(lambda: fb(lambda v: rv(v)), 50)
    out["offline_32x64000_fwd_bwd_stock_ms"] = timed(lambda: fb(lambda v: stock_forward(rv, v)), 50)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
