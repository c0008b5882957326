#!/usr/bin/env python3
"""One spectral-loss scale against the fp64 torch formulation for a given shape (debugging aid of the fuzz sweep):
mss_case.py n_fft overlap B L [seed] -> relative L2 / max error of the gradient, the same for torch's own fp32, worst positions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddsp_pytorch_amd.training import SpectralLoss  # noqa: E402

n_fft, overlap, B, L = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seeds = [int(sys.argv[5])] if len(sys.argv) > 5 else range(6)
for seed in seeds:
    g = torch.Generator().manual_seed(seed)
    x_true = 0.3 * torch.randn(B, L, generator=g)
    x_pred = 0.3 * torch.randn(B, L, generator=g)
    sl = SpectralLoss(n_fft, alpha=1.0, overlap=overlap)
    xp = x_pred.double().requires_grad_(True)
    ref = sl.double()(xp, x_true.double())
    ref.backward()
    sg = SpectralLoss(n_fft, alpha=1.0, overlap=overlap).cuda()
    xg = x_pred.cuda().requires_grad_(True)
    got = sg(xg, x_true.cuda())
    got.backward()
    gd = xg.grad.cpu().double() - xp.grad
    x32 = x_pred.clone().requires_grad_(True)
    SpectralLoss(n_fft, alpha=1.0, overlap=overlap)(x32, x_true).backward()
    d32 = x32.grad.double() - xp.grad
    worst = gd.abs().flatten().topk(min(5, gd.numel()))
    print(f"seed {seed}: loss rel {abs(got.item() - ref.item()) / abs(ref.item()):.1e} | hip grad L2 {float(gd.norm() / xp.grad.norm()):.2e} max "
          f"{float(gd.abs().max() / xp.grad.abs().max()):.2e} | torch fp32 L2 {float(d32.norm() / xp.grad.norm()):.2e} | worst idx {worst.indices.tolist()} "
          f"err {[f'{v:.1e}' for v in worst.values.tolist()]} ref {[f'{xp.grad.flatten()[i].item():.1e}' for i in worst.indices.tolist()]}")
