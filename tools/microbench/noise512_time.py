#!/usr/bin/env python3
"""cfg3 filtered noise (batch 512 x 375 frames, hop 512, 257 bands; also 195 bands = the reference's default): device time of the
in-LDS FFT form, plain and accumulating, HIP events from the library's profile hooks.  One JSON line."""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


LAST_IR = []      # the matrix product's own time (cosine operand + product), per run


def run(H, y, acc, reps=10):
    for _ in range(2):
        ddsp.noise_forward(H, 512, seed=1, out=y, accumulate=acc)
    ddsp._lib.profile_enable(reps + 4)
    torch.cuda.synchronize()
    for i in range(reps):
        ddsp.noise_forward(H, 512, seed=1, offset=i << 32, out=y, accumulate=acc)
    torch.cuda.synchronize()
    rec = ddsp._lib.profile_read()
    ddsp._lib.profile_enable(0)
    ms = [m for n, m in rec if n == "noise_frame"]
    ir = [m for n, m in rec if n == "noise_impulse_responses"]      # (195 bands: the matrix product ahead of the FFT form)
    if ir:
        LAST_IR.append(round(float(np.mean(ir)), 4))
    return round(float(np.mean(ms)) + (float(np.mean(ir)) if ir else 0.0), 4)


if __name__ == "__main__":
    out = {}
    rng = np.random.default_rng(1)
    for F in (257, 195):
        H = torch.from_numpy(syn.controller_range(rng.standard_normal((512, 375, F), dtype=np.float32))).cuda()
        y = torch.zeros(512, 375 * 512, device="cuda")
        out[f"F{F}"] = {"acc_ms": [run(H, y, True) for _ in range(3)], "plain_ms": [run(H, y, False) for _ in range(3)]}
        if F == 195:      # the cosine-sum path the matrix product replaced (ddsp_noise_set_generic bit 4)
            ddsp._lib.lib().ddsp_noise_set_generic(16)
            out["F195_cosine_sums"] = {"acc_ms": [run(H, y, True) for _ in range(3)], "plain_ms": [run(H, y, False) for _ in range(3)]}
            ddsp._lib.lib().ddsp_noise_set_generic(0)
        del H, y
    out["F195_product_only_ms"] = LAST_IR
    print(json.dumps(out))
