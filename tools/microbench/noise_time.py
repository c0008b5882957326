#!/usr/bin/env python3
"""Device time of the filtered-noise forward per shape: default path (in-LDS FFT form for hop 256 / 512) against the direct
batched kernels (ddsp_noise_set_generic(2)).  HIP-event timing from the library's own profile hooks."""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")  # kernel-form / tiling hooks (include/ddsp_hip.h)
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


def run(B, T, hop, F, mode, reps=10):
    L = ddsp._lib.lib()
    rng = np.random.default_rng(1)
    H = torch.from_numpy(syn.controller_range(rng.standard_normal((B, T, F), dtype=np.float32))).cuda()
    y = torch.zeros(B, T * hop, device="cuda")
    L.ddsp_noise_set_generic(mode)
    try:
        for _ in range(3):
            ddsp.noise_forward(H, hop, seed=1, out=y, accumulate=True)
        ddsp._lib.profile_enable(reps + 4)
        torch.cuda.synchronize()
        for _ in range(reps):
            ddsp.noise_forward(H, hop, seed=1, out=y, accumulate=True)
        torch.cuda.synchronize()
        ms = [m for n, m in ddsp._lib.profile_read() if n == "noise_frame"]
        ddsp._lib.profile_enable(0)
    finally:
        L.ddsp_noise_set_generic(0)
    return float(np.mean(ms))


if __name__ == "__main__":
    out = {}
    for name, (B, T, hop, F) in {"cfg3_b512_hop512_f257": (512, 375, 512, 257), "rt_default_b64_hop512_f195": (64, 172, 512, 195),
                                 "b512_hop256_f129": (512, 250, 256, 129), "b512_hop512_f65": (512, 375, 512, 65)}.items():
        out[name] = {"fft_form_ms": round(run(B, T, hop, F, 0), 4), "direct_form_ms": round(run(B, T, hop, F, 2), 4),
                     "samples": B * T * hop}
    print(json.dumps(out))
