#!/usr/bin/env python3
"""A few launches of the filtered-noise forward at one shape, for rocprofv3 (kernel trace / PMC passes).
usage: noise_prof.py B T hop F [mode]"""
import os as _os; _os.environ.setdefault("DDSP_TEST_HOOKS", "1")  # kernel-form / tiling hooks (include/ddsp_hip.h)
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

B, T, hop, F = (int(v) for v in sys.argv[1:5])
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rng = np.random.default_rng(1)
H = torch.from_numpy(syn.controller_range(rng.standard_normal((B, T, F), dtype=np.float32))).cuda()
y = torch.zeros(B, T * hop, device="cuda")
ddsp._lib.lib().ddsp_noise_set_generic(mode)
for _ in range(6):
    ddsp.noise_forward(H, hop, seed=1, out=y, accumulate=True)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
