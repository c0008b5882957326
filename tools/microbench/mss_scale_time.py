#!/usr/bin/env python3
"""Device time of ONE spectral-loss scale kernel (ddsp_mss_scale, value + gradient frames) per transform size and batch:
does it scale with the work or sit on a fixed cost?  torch.cuda events around 20 launches.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddsp_pytorch_amd import _lib  # noqa: E402

L = _lib.lib()
out = {}
for B in (8, 32, 128):
    n = 64000
    xp, xt = torch.randn(B, n, device="cuda") * 0.1, torch.randn(B, n, device="cuda") * 0.1
    res = torch.empty(3, device="cuda")
    scratch = torch.empty(L.ddsp_mss_scale_scratch_bytes(), device="cuda", dtype=torch.uint8)
    for n_fft in (2048, 1024, 512, 256, 128, 64):
        hop = n_fft // 4
        win = torch.hann_window(n_fft, device="cuda")
        frames = torch.empty((B * (1 + n // hop), n_fft), device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        row = {}
        for grad in (True, False):
            fp = frames.data_ptr() if grad else None
            for _ in range(3):
                _lib.check(L.ddsp_mss_scale(xp.data_ptr(), xt.data_ptr(), win.data_ptr(), fp, scratch.data_ptr(), res.data_ptr(), B, n, n_fft, hop, 1.0, 1e-7, st), "mss")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                L.ddsp_mss_scale(xp.data_ptr(), xt.data_ptr(), win.data_ptr(), fp, scratch.data_ptr(), res.data_ptr(), B, n, n_fft, hop, 1.0, 1e-7, st)
            e1.record()
            torch.cuda.synchronize()
            row["with_grad_us" if grad else "value_only_us"] = round(1e3 * e0.elapsed_time(e1) / 20, 1)
        out[f"b{B}_n{n_fft}"] = row
print(json.dumps(out))
