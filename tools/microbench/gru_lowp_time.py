#!/usr/bin/env python3
"""GRU recurrence, fp32 kernels vs the bf16 matrix-core variants (autocast callers): microseconds per time step."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddsp_pytorch_amd import gru as G  # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


out = []
for (B, T, hd) in [(32, 500, 512), (1, 32, 512), (8, 500, 512), (64, 500, 512), (128, 500, 512), (32, 500, 128)]:
    gi = torch.randn(B, T, 3 * hd, device="cuda")
    w = torch.randn(3 * hd, hd, device="cuda") * 0.05
    b = torch.zeros(3 * hd, device="cuda")
    h0 = torch.zeros(B, hd, device="cuda")
    row = {"recurrence": (B, T, hd)}
    for lowp in (False, True):
        f = timeit(lambda: G.gru_forward(gi, w, b, h0, save=True, lowp=lowp))
        y, hT, gates, hn = G.gru_forward(gi, w, b, h0, save=True, lowp=lowp)
        dy = torch.randn_like(y)
        bw = timeit(lambda: G.gru_backward(dy, None, w, h0, y, gates, hn, lowp=lowp))
        tag = "bf16_mfma" if lowp else "fp32"
        row[tag + "_fwd_us_per_step"] = round(f * 1e3 / T, 2)
        row[tag + "_bwd_us_per_step"] = round(bw * 1e3 / T, 2)
    out.append(row)
    print(json.dumps(row), flush=True)
