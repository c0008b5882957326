#!/usr/bin/env python3
"""Weight-gradient GEMMs of the controller's dense layers, dW[N,K] = gy[M,N]^T x[M,K] with M = 16000 (batch 32 x 500 frames):
one library GEMM (what autograd's Linear backward issues: 16-64 output tiles on 256 CUs) against a manual split-K
(S batched GEMMs over M / S rows each + a sum), fp32 and bf16."""
import json
import sys

import torch


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
M = 16000
for dt in (torch.float32, torch.bfloat16):
    for (N, K) in ((512, 512), (1536, 1024), (512, 1536), (100, 512), (1536, 512)):
        gy = torch.randn(M, N, device="cuda", dtype=dt)
        x = torch.randn(M, K, device="cuda", dtype=dt)
        row = {"plain_ms": round(timed(lambda: gy.t() @ x), 4)}
        for S in (4, 8, 16, 32):
            if M % S:
                continue
            g3, x3 = gy.view(S, M // S, N), x.view(S, M // S, K)
            row[f"splitk{S}_ms"] = round(timed(lambda: torch.bmm(g3.transpose(1, 2), x3).sum(0)), 4)
        out[f"{str(dt).split('.')[-1]}_{N}x{K}"] = row
        print(json.dumps({f"{str(dt).split('.')[-1]}_{N}x{K}": row}), flush=True)
