#!/usr/bin/env python3
"""Hop-128 / 65-band filtered noise at the bench shape (batch 512 x 500 frames): the wavefront-private form (default) against the
batched kernel it replaced (ddsp_noise_set_generic(8)), interleaved in one process; plain and accumulating, in-kernel and
resident draws.  HIP-event timing from the library's own profile hooks.  One JSON line."""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")  # kernel-form hooks (include/ddsp_hip.h)
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


def run(H, y, mode, accumulate, uniform, reps=20):
    L = ddsp._lib.lib()
    L.ddsp_noise_set_generic(mode)
    try:
        for _ in range(3):
            ddsp.noise_forward(H, 128, seed=1, uniform=uniform, out=y, accumulate=accumulate)
        ddsp._lib.profile_enable(reps + 4)
        torch.cuda.synchronize()
        for i in range(reps):
            ddsp.noise_forward(H, 128, seed=1, offset=i << 32, uniform=uniform, out=y, accumulate=accumulate)
        torch.cuda.synchronize()
        ms = [m for n, m in ddsp._lib.profile_read() if n == "noise_frame"]
        ddsp._lib.profile_enable(0)
    finally:
        L.ddsp_noise_set_generic(0)
    return float(np.mean(ms)), float(np.min(ms))


if __name__ == "__main__":
    B, T = int(os.environ.get("B", 512)), 500
    rng = np.random.default_rng(1)
    H = torch.from_numpy(syn.controller_range(rng.standard_normal((B, T, 65), dtype=np.float32))).cuda()
    y = torch.zeros(B, T * 128, device="cuda")
    u = torch.rand(B, T, 128, device="cuda")
    out = {"batch": B, "frames": T}
    for rnd in range(3):
        for name, mode in (("wave", 0), ("batched_r2", 8)):
            for acc in (True, False):
                for draw, un in (("philox", None), ("resident", u)):
                    key = f"{name}_{'acc' if acc else 'plain'}_{draw}"
                    mean, mn = run(H, y, mode, acc, un)
                    out.setdefault(key, []).append(round(mean, 4))
    print(json.dumps(out))
