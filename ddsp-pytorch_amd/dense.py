"""Dense layers of the control network (model/autoencoder/decoder.py:9-39, :60-72) as stock library GEMMs with ONE change
in the backward: the weight gradient dW[N,K] = gy[M,N]^T x[M,K] sums over M = batch x frames (16 000 rows at the training
shape) but has only N x K / tile outputs -- 16 to 64 workgroups on a 256-CU chip in the library's choice of kernel.  It is
issued as S batched GEMMs over M / S rows each plus a sum over the S partial results (a split-K by hand): measured on
MI355X (tools/microbench/dw_splitk.py) 512x512 bf16 0.094 -> 0.033 ms, fp32 0.104 -> 0.076 ms, 100x512 0.09 -> 0.024 ms.
Forward and input gradient are the library GEMMs autograd would issue.  Under torch.autocast the three GEMMs run in the
autocast dtype (as F.linear would), the returned parameter gradients are fp32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SPLIT = 8            # partial GEMMs of the weight gradient
MIN_ROWS = 2048      # below this the plain GEMM is at least as fast


def weight_grad(gy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """gy [M,N], x [M,K] -> gy^T x [N,K] in fp32 (inputs may be bf16 / fp16: fp32 accumulation inside the GEMMs)."""
    M = gy.shape[0]
    # (thin outputs -- the 1-wide loudness head -- stay on the plain GEMM: batched bf16 GEMMs with N = 1 take the library's
    #  slow path on this stack, 9 ms of host time per call)
    if gy.is_cuda and M >= MIN_ROWS and M % SPLIT == 0 and gy.shape[1] >= 16 and x.shape[1] >= 16:
        parts = torch.bmm(gy.reshape(SPLIT, M // SPLIT, -1).transpose(1, 2), x.reshape(SPLIT, M // SPLIT, -1))
        return parts.sum(0, dtype=torch.float32)
    return (gy.t() @ x).float()


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        dt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else x.dtype
        if dt not in (torch.float32, torch.bfloat16, torch.float16):
            dt = torch.float32
        xc, wc = x.to(dt), weight.to(dt)
        with torch.autocast("cuda", enabled=False):
            y = F.linear(xc, wc, None if bias is None else bias.to(dt))
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.in_dtype, ctx.param_dtype = x.dtype, weight.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, wc = ctx.saved_tensors
        gy = gy.to(wc.dtype)
        gx = gw = gb = None
        with torch.autocast("cuda", enabled=False):
            g2 = gy.reshape(-1, gy.shape[-1])
            if ctx.needs_input_grad[0]:
                gx = (g2 @ wc).view(xc.shape).to(ctx.in_dtype)
            if ctx.needs_input_grad[1]:
                gw = weight_grad(g2, xc.reshape(-1, xc.shape[-1])).to(ctx.param_dtype)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = g2.sum(0, dtype=torch.float32).to(ctx.param_dtype)
        return gx, gw, gb


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """F.linear with the split weight-gradient GEMM on CUDA tensors that carry gradients; plain F.linear otherwise."""
    if x.is_cuda and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        return _Linear.apply(x, weight, bias)
    return F.linear(x, weight, bias)
