#!/usr/bin/env python3
"""Which frames of the hop-512 noise forms share rounding error: one loud frame in a quiet clip, error of every frame relative to ITS
OWN level, against the C oracle.  Prints a table per (T, F, loud frame)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402

rng = np.random.default_rng(3)
for hop, F in ((512, 257), (512, 121), (128, 65)):
    for T, loud in ((8, 2), (8, 5), (7, 3)):
        B = 2
        H = syn.controller_range(rng.standard_normal((B, T, F), dtype=np.float32))
        level = np.full((B, T, 1), 1e-3, dtype=np.float32)
        level[0, loud] = 1e3
        H = H * level
        u = rng.random((B, T, hop), dtype=np.float32)
        ref = oracle.noise_forward(H, u, hop)
        y = ddsp.noise_forward(torch.from_numpy(H).cuda(), hop, uniform=torch.from_numpy(u).cuda()).cpu().numpy()
        err = np.abs(y - ref).reshape(B, T, hop).max(axis=2) / np.maximum(level[:, :, 0], np.abs(ref).reshape(B, T, hop).max(axis=2))
        print(f"hop {hop} F {F} T {T} loud frame (0,{loud}):")
        for b in range(B):
            print("   row", b, " ".join(f"{e:8.1e}" for e in err[b]))
