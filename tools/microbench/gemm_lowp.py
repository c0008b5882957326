#!/usr/bin/env python3
"""Host + device cost of the controller's dense layers in fp32 / bf16 / fp16 through torch (rocBLAS vs hipBLASLt):
[16000, K] x [K, N] Linear forward + backward at the training shape (batch 32 x 500 frames)."""
import json
import sys
import time

import torch
import torch.nn.functional as F


def run(dtype, lib, M=16000, shapes=((512, 512), (1024, 1536), (1536, 512), (1, 512), (512, 100), (512, 1))):
    torch.backends.cuda.preferred_blas_library(lib)
    out = {}
    for K, N in shapes:
        x = torch.randn(M, K, device="cuda", dtype=dtype, requires_grad=True)
        w = torch.randn(N, K, device="cuda", dtype=dtype, requires_grad=True)
        b = torch.randn(N, device="cuda", dtype=dtype, requires_grad=True)
        g = torch.randn(M, N, device="cuda", dtype=dtype)

        def step():
            y = F.linear(x, w, b)
            y.backward(g)
            x.grad = w.grad = b.grad = None

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            step()
        e1.record()
        host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        out[f"{K}x{N}"] = {"host_issue_ms": round(1e3 * host, 4), "device_ms": round(e0.elapsed_time(e1) / 50, 4)}
    return out


if __name__ == "__main__":
    res = {}
    for lib in ("cublas", "cublaslt"):
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            res[f"{lib}/{str(dt).split('.')[-1]}"] = run(dt, lib)
    print(json.dumps(res, indent=1))
