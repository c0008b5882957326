import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from ddsp_pytorch_amd.training import SpectralLoss
res = {}
for seed in range(12):
    g = torch.Generator().manual_seed(1000 + seed)
    n_fft, overlap, B, L = 2048, 0.0, 1, 1100 + 37 * seed
    x_true = 0.3 * torch.randn(B, L, generator=g); x_pred = 0.3 * torch.randn(B, L, generator=g)
    sl = SpectralLoss(n_fft, alpha=1.0, overlap=overlap)
    xp = x_pred.double().requires_grad_(True); ref = sl.double()(xp, x_true.double()); ref.backward()
    sg = SpectralLoss(n_fft, alpha=1.0, overlap=overlap).cuda()
    xg = x_pred.cuda().requires_grad_(True); got = sg(xg, x_true.cuda()); got.backward()
    gd = xg.grad.cpu().double() - xp.grad
    x32 = x_pred.clone().requires_grad_(True); SpectralLoss(n_fft, alpha=1.0, overlap=overlap)(x32, x_true).backward()
    d32 = x32.grad.double() - xp.grad
    print(seed, L, "hip L2 %.2e max %.2e | torch fp32 L2 %.2e max %.2e" % (float(gd.norm()/xp.grad.norm()), float(gd.abs().max()/xp.grad.abs().max()), float(d32.norm()/xp.grad.norm()), float(d32.abs().max()/xp.grad.abs().max())))
