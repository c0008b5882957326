#!/usr/bin/env python3
"""Replays case `index` of `tests/fuzz_parity.py <n> <seed> training` (same generators) and prints its gradient error against fp64."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddsp_pytorch_amd.training import SpectralLoss  # noqa: E402

seed, index = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
g = torch.Generator().manual_seed(seed)
for i in range(index + 1):
    n_fft = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
    overlap = float(rng.choice([0.75, 0.75, 0.5, 0.875, 0.0]))
    B = int(rng.integers(1, 6))
    L = int(rng.integers(n_fft // 2 + 1, n_fft // 2 + 1 + int(rng.choice([3, 200, 5000]))))
    x_true = 0.3 * torch.randn(B, L, generator=g)
    x_pred = 0.3 * torch.randn(B, L, generator=g)
    if rng.random() < 0.3:
        x_true[0, : L // 2] = 0.0
    alpha = float(rng.choice([1.0, 0.3]))
    torch.randn(x_pred.shape, generator=g, dtype=torch.float64)      # (the sweep's conditioning probe)
    if i < index:
        # consume what the sweep consumes after this point of an iteration
        sl_tmp = SpectralLoss(n_fft, alpha=alpha, overlap=overlap)
        fa_shape = sl_tmp.stft_ri(x_pred).shape
        torch.randn(fa_shape, generator=g)
        M, N = int(rng.integers(0, 20000)), int(rng.integers(1, 1600))
        int(rng.integers(0, 3))
        torch.randn(M, N, generator=g)
print("case", n_fft, overlap, B, L, alpha, "silent" if float(x_true[0, : L // 2].abs().max()) == 0 else "")
sl = SpectralLoss(n_fft, alpha=alpha, overlap=overlap)
xp = x_pred.double().requires_grad_(True)
ref = sl.double()(xp, x_true.double()); ref.backward()
sg = SpectralLoss(n_fft, alpha=alpha, overlap=overlap).cuda()
xg = x_pred.cuda().requires_grad_(True)
got = sg(xg, x_true.cuda()); got.backward()
gd = xg.grad.cpu().double() - xp.grad
x32 = x_pred.clone().requires_grad_(True)
SpectralLoss(n_fft, alpha=alpha, overlap=overlap)(x32, x_true).backward()
d32 = x32.grad.double() - xp.grad
gq = torch.Generator().manual_seed(1)
xq = (x_pred.double() + 6e-8 * 0.3 * torch.randn(x_pred.shape, generator=gq, dtype=torch.float64)).requires_grad_(True)
sl.double()(xq, x_true.double()).backward()
print(f"conditioning: an fp32-epsilon input perturbation moves the fp64 gradient by L2 {float((xq.grad - xp.grad).norm() / xp.grad.norm()):.2e}")
print(f"hip L2 {float(gd.norm() / xp.grad.norm()):.2e} max {float(gd.abs().max() / xp.grad.abs().max()):.2e} | torch fp32 L2 {float(d32.norm() / xp.grad.norm()):.2e} max {float(d32.abs().max() / xp.grad.abs().max()):.2e}")
