"""Dense layers of the control network (model/autoencoder/decoder.py:9-39, :60-72) as stock library GEMMs with ONE change
in the backward: the weight gradient dW[N,K] = gy[M,N]^T x[M,K] sums over M = batch x frames (16 000 rows at the training
shape) but has only N x K / tile outputs -- 16 to 64 workgroups on a 256-CU chip in the library's choice of kernel.  It is
issued as S batched GEMMs over M / S rows each plus a sum over the S partial results (a split-K by hand): measured on
MI355X (tools/microbench/dw_splitk.py) 512x512 bf16 0.094 -> 0.033 ms, fp32 0.104 -> 0.076 ms, 100x512 0.09 -> 0.024 ms.
Forward and input gradient are the library GEMMs autograd would issue.  Under torch.autocast the three GEMMs run in the
autocast dtype (as F.linear would), the returned parameter gradients are fp32.
"""
from __future__ import annotations

import weakref

import torch
import torch.nn.functional as F

from . import _lib

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


_IO = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def colsum(x: torch.Tensor) -> torch.Tensor:
    """[M, N] -> fp32 [N] = sum over the rows: a dense layer's bias gradient.  One streaming HIP pass on CUDA tensors
    (include/ddsp_hip.h: ddsp_colsum; the stock reduction of a tall thin matrix costs 12-150 us per layer at the training shape)."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype in _IO):
        return x.sum(0, dtype=torch.float32)
    x = x.contiguous()
    M, N = x.shape
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    scratch = torch.empty(L.ddsp_colsum_scratch_bytes(N), device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        _lib.check(L.ddsp_colsum(x.data_ptr(), out.data_ptr(), scratch.data_ptr(), M, N, _IO[x.dtype],
                                 torch.cuda.current_stream().cuda_stream), "ddsp_colsum")
    return out


class LowpWeights:
    """Low-precision copies of the dense layers' parameters for torch.autocast steps, refreshed by ONE multi-tensor copy per
    step instead of a cast launch per parameter and layer (31 launches at the training shape).  `refresh(dtype)` is called once
    before the forward (train_step and GraphedTrainStep do) and renews -- unconditionally: it is one `_foreach_copy_` -- the
    copies of every parameter `_Linear` has asked for so far, then `release()` is called after the forward.  `_Linear` takes
    a copy only inside that refresh..release window AND while the parameter's `_version` and storage address are what they
    were at the refresh.  (`_version` alone is not enough: writes through `p.data` and the optimiser update inside a hipGraph
    replay do not bump it -- which is why the window exists: outside a step every autocast forward casts afresh.)"""

    def __init__(self):
        self._copies = {}      # id(param) -> [weakref(param), copy or None, version at refresh, data_ptr at refresh]
        self._live = None      # the dtype of the open refresh..release window

    def refresh(self, dtype):
        src, dst = [], []
        for key, e in list(self._copies.items()):
            p = e[0]()
            if p is None:
                del self._copies[key]
                continue
            if e[1] is None or e[1].dtype != dtype or e[1].shape != p.shape or e[1].device != p.device:
                e[1] = torch.empty_like(p, dtype=dtype)
            src.append(p.detach())
            dst.append(e[1])
            e[2], e[3] = p._version, p.data_ptr()
        if src:
            torch._foreach_copy_(dst, src)
        self._live = dtype

    def release(self):
        """Close the window: until the next refresh() every request is answered with None (the caller casts afresh)."""
        self._live = None

    def get(self, p, dtype):
        e = self._copies.get(id(p))
        if e is None or e[0]() is not p:
            if p.is_cuda and p.dtype == torch.float32 and isinstance(p, torch.nn.Parameter):
                self._copies[id(p)] = [weakref.ref(p), None, -1, 0]    # wanted: part of the next refresh
            return None
        if self._live == dtype and e[1] is not None and e[1].dtype == dtype and e[2] == p._version and e[3] == p.data_ptr():
            return e[1]
        return None


lowp_weights = LowpWeights()


def _cast(p, dt):
    if p.dtype == dt:
        return p
    c = lowp_weights.get(p, dt)
    return c if c is not None else p.to(dt)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        dt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else x.dtype
        if dt not in (torch.float32, torch.bfloat16, torch.float16):
            dt = torch.float32
        xc, wc = x.to(dt), _cast(weight, dt)
        with torch.autocast("cuda", enabled=False):
            y = F.linear(xc, wc, None if bias is None else _cast(bias, dt))
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
                gb = colsum(g2).to(ctx.param_dtype)
        return gx, gw, gb


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """F.linear with the split weight-gradient GEMM on CUDA tensors that carry gradients; plain F.linear otherwise."""
    if x.is_cuda and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        return _Linear.apply(x, weight, bias)
    return F.linear(x, weight, bias)
