"""`GRU`: `torch.nn.GRU` with the recurrence on one persistent HIP launch (include/ddsp_hip.h: ddsp_gru_*).

The reference's control network runs `nn.GRU(2*width, units, layers, batch_first=True)` over the whole clip
(model/autoencoder/decoder.py:60-65, :91) and over one callback's frames with a carried state in the live path
(:91 via `forward_live` :139-147).  This class IS an `nn.GRU` (same parameters `weight_ih_l0 / weight_hh_l0 /
bias_ih_l0 / bias_hh_l0`, so the reference's checkpoints load unchanged); for CUDA inputs of a unidirectional GRU
with hidden size <= 512 (any number of stacked layers, one recurrence launch per layer) it computes

    gi = x W_ih^T + b_ih                       one library GEMM (differentiated by autograd as usual)
    y, h_T = recurrence(gi, W_hh, b_hh, h_0)   csrc/ddsp_gru.hip, forward and backward

instead of MIOpen's per-time-step launches.  Other configurations, and CPU tensors, use the stock `nn.GRU`
implementation (the controller is a stock-layer caller of the hot path, not part of it).  On a CUDA input a missing
library raises `DdspHipError` -- there is no silent fallback.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import dense


_debug = os.environ.get("DDSP_GRU_DEBUG", "0") not in ("", "0")


def set_debug(flag: bool) -> bool:
    """Debug mode: after every recurrence launch the host reads the launch's status word (one synchronisation per
    launch) and raises `DdspHipError` if a workgroup gave up waiting for its peers.  Without it a failed launch is
    still loud -- every output of the steps it did not finish is NaN -- but asynchronous.  Returns the previous mode."""
    global _debug
    old, _debug = _debug, bool(flag)
    return old


def _ptr(t):
    return None if t is None else t.data_ptr()


def _raise_on_timeout(scratches, what: str) -> None:
    if torch.cuda.is_current_stream_capturing():
        return       # a status read is a synchronising copy: not capturable (a hipGraph's launches stay loud through their NaN outputs)
    for s in scratches:
        if gru_status(s) != 0:
            raise _lib.DdspHipError(
                f"{what}: a workgroup of the persistent GRU launch timed out waiting for its peers (status 1). The launch needs "
                "its whole grid resident: another persistent launch from a different process was probably holding the GPU's "
                "compute units. The outputs of the unfinished steps are NaN.")


def gru_forward(gi, w_hh, b_hh, h0, save: bool, scratch_out: list | None = None, lowp: bool = False):
    """Raw launcher of ddsp_gru_forward. gi [B,T,3Hd] -> (y [B,T,Hd], hT [B,Hd], gates | None, hn | None).
    `scratch_out` (tests): receives the scratch buffer of every launch, for `gru_status`.
    `lowp`: the bf16 matrix-core variant (ddsp_gru_forward_bf16; autocast callers)."""
    B, T, G3 = gi.shape
    Hd = G3 // 3
    L = _lib.lib()
    y = torch.empty((B, T, Hd), device=gi.device, dtype=torch.float32)
    hT = torch.empty((B, Hd), device=gi.device, dtype=torch.float32)
    gates = torch.empty((B, T, G3), device=gi.device, dtype=torch.float32) if save else None
    hn = torch.empty((B, T, Hd), device=gi.device, dtype=torch.float32) if save else None
    if B == 0:
        return y, hT, gates, hn
    with torch.cuda.device(gi.device):
        cap = L.ddsp_gru_max_batch(Hd, 2 if lowp else 0)
        if cap <= 0:
            raise _lib.DdspHipError(f"ddsp_gru_forward: hidden size {Hd} is not supported")
        stream = torch.cuda.current_stream().cuda_stream
        launch = L.ddsp_gru_forward_bf16 if lowp else L.ddsp_gru_forward
        for lo in range(0, B, cap):                      # rows are independent: larger batches go in slices
            hi = min(B, lo + cap)
            scratch = torch.empty(L.ddsp_gru_scratch_bytes(hi - lo, Hd), device=gi.device, dtype=torch.uint8)
            rc = launch(gi[lo:hi].data_ptr(), w_hh.data_ptr(), _ptr(b_hh), _ptr(h0[lo:hi]) if h0 is not None else None,
                                    y[lo:hi].data_ptr(), hT[lo:hi].data_ptr(), _ptr(gates[lo:hi]) if save else None,
                                    _ptr(hn[lo:hi]) if save else None, scratch.data_ptr(), hi - lo, T, Hd, stream)
            _lib.check(rc, "ddsp_gru_forward")
            if scratch_out is not None:
                scratch_out.append(scratch)
    return y, hT, gates, hn


def gru_backward(dy, dhT, w_hh, h0, y, gates, hn, scratch_out: list | None = None, lowp: bool = False, io16: bool = False):
    """Raw launcher of ddsp_gru_backward -> (d_gi [B,T,3Hd], d_gh [B,T,3Hd], dh0 [B,Hd]).
    `io16` (with `lowp`): d_gi / d_gh come back as bf16 tensors, ready for the autocast GEMMs that consume them."""
    B, T, Hd = y.shape
    L = _lib.lib()
    io16 = io16 and lowp
    d_gi = torch.empty_like(gates, dtype=torch.bfloat16 if io16 else torch.float32)
    d_gh = torch.empty_like(d_gi)
    dh0 = torch.empty((B, Hd), device=y.device, dtype=torch.float32)
    with torch.cuda.device(y.device):
        cap = L.ddsp_gru_max_batch(Hd, 3 if lowp else 1)
        stream = torch.cuda.current_stream().cuda_stream
        launch = L.ddsp_gru_backward_bf16 if lowp else L.ddsp_gru_backward
        for lo in range(0, B, cap):
            hi = min(B, lo + cap)
            scratch = torch.empty(L.ddsp_gru_scratch_bytes(hi - lo, Hd), device=y.device, dtype=torch.uint8)
            args = (dy[lo:hi].data_ptr(), _ptr(dhT[lo:hi]) if dhT is not None else None, w_hh.data_ptr(),
                    _ptr(h0[lo:hi]) if h0 is not None else None, y[lo:hi].data_ptr(), gates[lo:hi].data_ptr(),
                    hn[lo:hi].data_ptr(), d_gi[lo:hi].data_ptr(), d_gh[lo:hi].data_ptr(), dh0[lo:hi].data_ptr(),
                    scratch.data_ptr(), hi - lo, T, Hd)
            rc = launch(*args, 1 if io16 else 0, stream) if lowp else launch(*args, stream)
            if rc == -2 and io16:
                # (DDSP_ERANGE: sequences of >= 65536 steps take the fp32 backward kernels, which write fp32 gradients --
                # redo the slice with fp32 outputs and cast)
                f_gi = torch.empty_like(gates[lo:hi], dtype=torch.float32)
                f_gh = torch.empty_like(f_gi)
                args32 = args[:7] + (f_gi.data_ptr(), f_gh.data_ptr()) + args[9:]
                rc = launch(*args32, 0, stream)
                _lib.check(rc, "ddsp_gru_backward")
                d_gi[lo:hi].copy_(f_gi)
                d_gh[lo:hi].copy_(f_gh)
            _lib.check(rc, "ddsp_gru_backward")
            if scratch_out is not None:
                scratch_out.append(scratch)
    return d_gi, d_gh, dh0


def gru_status(scratch) -> int:
    """0, or 1 if a workgroup of the launch that used `scratch` timed out waiting for its peers (synchronises)."""
    out = ctypes.c_int(0)
    _lib.check(_lib.lib().ddsp_gru_status(scratch.data_ptr(), ctypes.byref(out)), "ddsp_gru_status")
    return out.value


class _Recurrence(torch.autograd.Function):
    """gi [B,T,3Hd] -> (y, h_T).  Everything at the boundary is fp32, with one exception: under bf16 autocast the input projection
    wants a bf16 gradient back and the W_hh gradient GEMM consumes bf16 too -- the backward kernel writes d_gi / d_gh as bf16
    (include/ddsp_hip.h: io_type) instead of a cast pass on each.  (`gi` itself is cast to fp32 in front of the forward kernel:
    reading it as bf16 measured slower than the cast costs.)"""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0, gemm_dtype=None):
        lowp = gemm_dtype is not None        # under autocast the recurrence's products run on the matrix cores in bf16 as well
        io16 = lowp and gi.dtype == torch.bfloat16
        gi_dtype = gi.dtype
        gi = gi.contiguous().float()
        w = w_hh.detach().contiguous().float()
        b = None if b_hh is None else b_hh.detach().contiguous().float()
        h = None if h0 is None else h0.detach().contiguous().float()
        need = any(ctx.needs_input_grad)
        launched = [] if _debug else None
        y, hT, gates, hn = gru_forward(gi.detach(), w, b, h, save=need, scratch_out=launched, lowp=lowp)
        if _debug:
            _raise_on_timeout(launched, "ddsp_gru_forward")
        if need:
            ctx.save_for_backward(w, h, y, gates, hn)
            ctx.has_bias = b is not None
            ctx.gemm_dtype = gemm_dtype
            ctx.io16 = io16
            ctx.in_dtypes = (w_hh.dtype, None if b_hh is None else b_hh.dtype, None if h0 is None else h0.dtype, gi_dtype)
        return y, hT

    @staticmethod
    def backward(ctx, dy, dhT):
        w, h0, y, gates, hn = ctx.saved_tensors
        B, T, Hd = y.shape
        dy = torch.zeros_like(y) if dy is None else dy.contiguous().float()
        dhT = None if dhT is None else dhT.contiguous().float()
        launched = [] if _debug else None
        d_gi, d_gh, dh0 = gru_backward(dy, dhT, w, h0, y, gates, hn, scratch_out=launched, lowp=ctx.gemm_dtype is not None, io16=ctx.io16)
        if _debug:
            _raise_on_timeout(launched, "ddsp_gru_backward")
        dw = db = None
        wdt, bdt, hdt, gdt = ctx.in_dtypes
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[1]:
                first = h0 if h0 is not None else torch.zeros((B, Hd), device=y.device, dtype=y.dtype)
                h_prev = torch.cat((first.unsqueeze(1), y[:, :-1]), dim=1)          # h_{t-1} for every step
                a, b_ = d_gh.reshape(B * T, 3 * Hd), h_prev.reshape(B * T, Hd)
                if ctx.gemm_dtype is not None:                                      # the forward ran under autocast: so does this GEMM
                    a, b_ = a.to(ctx.gemm_dtype), b_.to(ctx.gemm_dtype)
                dw = dense.weight_grad(a, b_).to(wdt)                               # library GEMMs [3Hd, BT] x [BT, Hd], split over BT
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = dense.colsum(d_gh.reshape(B * T, 3 * Hd)).to(bdt)
        return d_gi.to(gdt), dw, db, (dh0.to(hdt) if ctx.needs_input_grad[3] else None), None


class GRU(nn.GRU):
    """Drop-in `nn.GRU`; see the module docstring for when the HIP recurrence runs."""

    def _hip_eligible(self, x) -> bool:
        # anything that is not a dense batched CUDA tensor (PackedSequence, unbatched 2-D input, CPU) is nn.GRU's business
        return (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 3 and not self.bidirectional and self.proj_size == 0
                and self.hidden_size <= 512 and x.dtype in (torch.float32, torch.bfloat16, torch.float16))

    def forward(self, input, hx=None):  # noqa: A002 (torch's argument name)
        if not self._hip_eligible(input):
            return super().forward(input, hx)
        x = input if self.batch_first else input.transpose(0, 1)
        if hx is not None and (hx.dim() != 3 or hx.shape[0] != self.num_layers or hx.shape[1] != x.shape[0]
                               or hx.shape[2] != self.hidden_size):
            raise RuntimeError(f"Expected hidden size ({self.num_layers}, {x.shape[0]}, {self.hidden_size}), got {list(hx.shape)}")
        finals = []
        amp = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
        for layer in range(self.num_layers):     # decoder.py:60-65 `num_layers=conf.decoder_gru_layers`: stacked, layer by layer
            b_ih = getattr(self, f"bias_ih_l{layer}", None) if self.bias else None
            b_hh = getattr(self, f"bias_hh_l{layer}", None) if self.bias else None
            gi = dense.linear(x, getattr(self, f"weight_ih_l{layer}"), b_ih)
            x, hT = _Recurrence.apply(gi, getattr(self, f"weight_hh_l{layer}"), b_hh, None if hx is None else hx[layer], amp)
            finals.append(hT)
            if self.dropout > 0 and self.training and layer + 1 < self.num_layers:
                x = F.dropout(x, self.dropout, True)
        if not self.batch_first:
            x = x.transpose(0, 1)
        return x, torch.stack(finals, dim=0)
