#!/usr/bin/env python3
"""Re-derives ONE case of the randomised loss-side sweep (tests/fuzz_parity.py: sweep_training_kernels) on the CPU by replaying the
sweep's random streams, and stores its inputs as a small fixture.  Used for the case the round-3 sweep flagged
(python tests/fuzz_parity.py 120 777 training: n_fft 2048, overlap 0.0, B 2, L 1081 -- a single frame barely longer than the padding;
gradient error 3.5e-3 of the norm against a conditioning yardstick of 2.5e-4 before the exact twiddle table of commit d59cda0):

    python tools/make_fuzz_case.py 777 2048 0.0 2 1081 tests/golden/g18_mss_fuzz_case.npz
"""
import sys

import numpy as np
import torch


def replay(seed, want, max_cases=400):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    for i in range(max_cases):
        n_fft = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
        overlap = float(rng.choice([0.75, 0.75, 0.5, 0.875, 0.0]))
        B = int(rng.integers(1, 6))
        L = int(rng.integers(n_fft // 2 + 1, n_fft // 2 + 1 + int(rng.choice([3, 200, 5000]))))
        x_true = 0.3 * torch.randn(B, L, generator=g)
        x_pred = 0.3 * torch.randn(B, L, generator=g)
        if rng.random() < 0.3:
            x_true[0, : L // 2] = 0.0
        alpha = float(rng.choice([1.0, 0.3]))
        if (n_fft, overlap, B, L) == want:
            return i, alpha, x_true, x_pred
        torch.randn(x_pred.shape, generator=g, dtype=torch.float64)          # the conditioning probe's perturbation
        hop = int(n_fft * (1 - overlap))
        frames = 1 + L // hop
        torch.randn((B, frames, n_fft // 2 + 1, 2), generator=g)              # the framing test's weights
        M, N = int(rng.integers(0, 20000)), int(rng.integers(1, 1600))
        rng.integers(0, 3)
        torch.randn(M, N, generator=g)                                        # the column-sum test's matrix
    raise SystemExit("case not found")


if __name__ == "__main__":
    seed, n_fft, overlap, B, L, out = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    i, alpha, x_true, x_pred = replay(seed, (n_fft, overlap, B, L))
    np.savez_compressed(out, x_true=x_true.numpy(), x_pred=x_pred.numpy(), n_fft=n_fft, overlap=overlap, alpha=alpha, seed=seed, case=i)
    print(f"case {i} of seed {seed}: n_fft {n_fft} overlap {overlap} B {B} L {L} alpha {alpha} -> {out}")
