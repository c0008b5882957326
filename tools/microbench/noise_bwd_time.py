#!/usr/bin/env python3
"""Device time of the filtered-noise BACKWARD at hop 512 (cfg3 shape: batch 512 x 375 frames; also batch 32 = a training shard):
the in-LDS FFT form (default) against the direct time-domain kernels (ddsp_noise_set_generic(2)), same process, interleaved.
torch.cuda events around 10 launches.  One JSON line."""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402


def run(gy, hop, F, mode, reps=10):
    L = ddsp._lib.lib()
    L.ddsp_noise_set_generic(mode)
    try:
        for _ in range(2):
            ddsp.noise_backward(gy, hop, F, seed=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            ddsp.noise_backward(gy, hop, F, seed=1, offset=i << 32)
        e1.record()
        torch.cuda.synchronize()
    finally:
        L.ddsp_noise_set_generic(0)
    return round(e0.elapsed_time(e1) / reps, 4)


if __name__ == "__main__":
    out = {}
    for B in (512, 32):
        gy = torch.randn(B, 375 * 512, device="cuda")
        for F in (257, 195):
            out[f"b{B}_F{F}"] = {"fft_form_ms": [run(gy, 512, F, 0) for _ in range(3)], "direct_form_ms": [run(gy, 512, F, 2) for _ in range(3)]}
    print(json.dumps(out))
