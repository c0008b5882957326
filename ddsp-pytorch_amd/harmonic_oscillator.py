"""Drop-in `OscillatorBank` (reference: model/ddsp/harmonic_oscillator.py:7-75) on the HIP kernels.

Same constructor (`conf.n_harmonics / sample_rate / hop_length`, :11-13), same
`forward(x)` / `live(x)` reading `x['f0'] [B,T,1]`, `x['c'] [B,T,H]`, `x['a'] [B,T,1]`
(:57-75), same state-dict keys `harmonics` and `last_phases` (:15-22), same result as the
reference evaluated on the CPU (the oracle BASELINE.json fixes) within 1e-5.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


def _dev_ptr(t):
    return None if t is None else t.data_ptr()


def _check_inputs(f0, c, a):
    if f0.dim() != 3 or c.dim() != 3 or a.dim() != 3 or f0.shape[-1] != 1 or a.shape[-1] != 1:
        raise ValueError("expected f0 [B,T,1], c [B,T,H], a [B,T,1]")
    if f0.shape[:2] != c.shape[:2] or a.shape[:2] != c.shape[:2]:
        raise ValueError(f"batch/frame mismatch: f0 {tuple(f0.shape)}, c {tuple(c.shape)}, a {tuple(a.shape)}")
    if not c.is_cuda:
        raise _lib.DdspHipError("OscillatorBank runs on the GPU only (no CPU fallback): move the controls to cuda")


OSC_KEEP_FRAME_SCRATCH = 1  # include/ddsp_hip.h: DDSP_OSC_KEEP_FRAME_SCRATCH


def osc_forward(f0, c, a, hop: int, sample_rate: int, live_in=None, want_live_out=False, debug_phases=False,
                return_scratch=False, keep_frame_scratch=None):
    """Raw launcher over the C ABI (include/ddsp_hip.h: ddsp_osc_forward_ex). Returns (y, live_out, phi[, scratch]).

    `return_scratch=True` (the autograd path) asks for the frame-form scratch ddsp_osc_backward re-walks, unless
    `keep_frame_scratch=False` (diagnostics: ddsp_osc_clock on the production launch's scratch)."""
    if keep_frame_scratch is None:
        keep_frame_scratch = return_scratch
    _check_inputs(f0, c, a)
    f0 = f0.detach().contiguous().float()
    c = c.detach().contiguous().float()
    a = a.detach().contiguous().float()
    B, T, H = c.shape
    L = _lib.lib()
    y = torch.empty((B, T * hop), device=c.device, dtype=torch.float32)
    if B == 0:
        return (y, None, None, None) if return_scratch else (y, None, None)
    scratch = torch.empty(L.ddsp_osc_scratch_bytes(B, T, H), device=c.device, dtype=torch.uint8)
    live_out = torch.empty(H, device=c.device, dtype=torch.float32) if want_live_out else None
    phi = torch.empty((B, T * hop, H), device=c.device, dtype=torch.float32) if debug_phases else None
    if live_in is not None:
        live_in = live_in.detach().to(device=c.device, dtype=torch.float32).contiguous()
    with torch.cuda.device(c.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = L.ddsp_osc_forward_ex(f0.data_ptr(), c.data_ptr(), a.data_ptr(), y.data_ptr(), scratch.data_ptr(),
                                   _dev_ptr(live_in), _dev_ptr(live_out), _dev_ptr(phi), B, T, H, hop, sample_rate,
                                   OSC_KEEP_FRAME_SCRATCH if keep_frame_scratch else 0, stream)
    _lib.check(rc, "ddsp_osc_forward_ex")
    return (y, live_out, phi, scratch) if return_scratch else (y, live_out, phi)


def osc_backward(grad_y, f0, c, a, scratch, hop: int, sample_rate: int):
    """Raw launcher of ddsp_osc_backward: -> (grad_c [B,T,H], grad_a [B,T,1]); `scratch` comes from the forward."""
    B, T, H = c.shape
    L = _lib.lib()
    grad_y = grad_y.detach().contiguous().float()
    grad_c = torch.empty_like(c)
    grad_a = torch.empty_like(a)
    if B == 0:
        return grad_c, grad_a
    bwd = torch.empty(L.ddsp_osc_backward_scratch_bytes(B, T, H), device=c.device, dtype=torch.uint8)
    with torch.cuda.device(c.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = L.ddsp_osc_backward(grad_y.data_ptr(), f0.data_ptr(), c.data_ptr(), a.data_ptr(), scratch.data_ptr(),
                                 bwd.data_ptr(), grad_c.data_ptr(), grad_a.data_ptr(), B, T, H, hop, sample_rate, stream)
    _lib.check(rc, "ddsp_osc_backward")
    return grad_c, grad_a


class _OscillatorFunction(torch.autograd.Function):
    """Differentiable w.r.t. c and a (train/train.py:33-34); f0 carries no gradient (decoder.py:105)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, f0, c, a, hop, sample_rate):
        f0 = f0.detach().contiguous().float()
        c = c.detach().contiguous().float()
        a = a.detach().contiguous().float()
        y, _, _, scratch = osc_forward(f0, c, a, hop, sample_rate, return_scratch=True)
        ctx.save_for_backward(f0, c, a, scratch)
        ctx.hop, ctx.sample_rate = hop, sample_rate
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_y):
        f0, c, a, scratch = ctx.saved_tensors
        grad_c, grad_a = osc_backward(grad_y, f0, c, a, scratch, ctx.hop, ctx.sample_rate)
        return None, grad_c, grad_a, None, None


class OscillatorBank(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.n_harmonics = conf.n_harmonics
        self.sample_rate = conf.sample_rate
        self.hop_size = conf.hop_length
        # same (non-trainable) parameters, dtypes and names as the reference (:15-22) so that its
        # checkpoints load with strict=True; `last_phases` starts int64 and becomes fp32 after live()
        self.harmonics = nn.Parameter(torch.arange(1, self.n_harmonics + 1, step=1), requires_grad=False)
        self.last_phases = nn.Parameter(torch.zeros_like(self.harmonics), requires_grad=False)

    def forward(self, x):
        f0, c, a = x['f0'], x['c'], x['a']
        if torch.is_grad_enabled() and (c.requires_grad or a.requires_grad):
            _check_inputs(f0, c, a)
            return _OscillatorFunction.apply(f0, c, a, self.hop_size, self.sample_rate)
        y, _, _ = osc_forward(f0, c, a, self.hop_size, self.sample_rate)
        return y

    def live(self, x):
        y, last, _ = osc_forward(x['f0'], x['c'], x['a'], self.hop_size, self.sample_rate,
                                 live_in=self.last_phases.data, want_live_out=True)
        self.last_phases.data = last  # :72 (only batch row 0 carries state, as in the reference)
        return y
