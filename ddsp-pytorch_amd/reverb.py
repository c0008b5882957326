"""Drop-in `Reverb` (reference: model/ddsp/reverb.py:8-49) -- SURVEY §8(f) next row 1.

Learned exponentially-decaying noise impulse (one second long) applied to the whole clip.  Same constructor,
parameter names/shapes (`noise`, `decay`, `wet`, `t`, `buffer`: checkpoint compatible), `forward(x)` and
`live_forward(x)` semantics, including the crop of the impulse for clips shorter than one second (:34) and tap 0
forced to 1 (:28).

On CUDA fp32 tensors the work around the two 2N-point library FFTs (rocFFT via torch.fft) is hand-written HIP
(csrc/ddsp_reverb.hip, include/ddsp_hip.h: ddsp_reverb_* / ddsp_spectral_mul*):

  forward       impulse (cropped to the clip, one launch) -> rfft -> spectral product (one launch) -> irfft, at the smallest
                fast transform length >= N + L - 1 (80 000 points at the training shape, not the reference's 2N = 128 000); the backward is ONE pass over rfft(grad) producing both `grad conj(K)` and the
                batch-reduced correlation spectrum, two irffts, and one deterministic launch for d noise / d decay / d wet.
  live_forward  no FFT at all: only the last n outputs of the one-second window are computed, as a direct causal
                convolution from the device-resident history (two launches per callback instead of three 2L-point
                transforms and ~10 elementwise launches).

CPU tensors run the same arithmetic as stock torch ops (the restatement the G9 fixtures pin on the CPU).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def causal_fft_convolve(signal: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """First len(signal) samples of the linear convolution along the last axis
    (what filtered_noise.py:25-32 `fft_convolve` returns)."""
    n = signal.shape[-1]
    spec = torch.fft.rfft(signal, n=2 * n) * torch.fft.rfft(kernel, n=2 * n)
    return torch.fft.irfft(spec, n=2 * n)[..., :n]


def fft_length(need: int) -> int:
    """Transform length of the HIP path's convolution: the smallest 2^a * {1, 3, 5, 25, 125, 625} >= need.
    The reference pads both operands to 2N (filtered_noise.py:26-27); the first N samples of a linear convolution of N
    samples with L <= N taps only need N + L - 1 points, and the library's transform of 128 000 = 2 * 64 000 points is also a
    slow size: 69.6 + 80.9 us (rfft + irfft, 32 rows) against 42.7 + 43.1 us at 80 000 (tools/microbench/fft_sizes.py)."""
    best = None
    for m in (1, 3, 5, 25, 125, 625):
        n = m
        while n < need or n % 2:
            n *= 2
        if best is None or n < best:
            best = n
    return best


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def reverb_impulse(noise, decay, wet, t, n_out: int) -> torch.Tensor:
    """Raw launcher of ddsp_reverb_impulse: the impulse of reverb.py:24-29, zero-padded or cropped to n_out taps (:34)."""
    length = noise.numel()
    imp = torch.empty(n_out, device=noise.device, dtype=torch.float32)
    with torch.cuda.device(noise.device):
        _lib.check(_lib.lib().ddsp_reverb_impulse(noise.data_ptr(), decay.data_ptr(), wet.data_ptr(), t.data_ptr(), imp.data_ptr(),
                                                  length, n_out, _stream(noise)), "ddsp_reverb_impulse")
    return imp


class _ReverbFunction(torch.autograd.Function):
    """y = first N samples of x * impulse(noise, decay, wet); differentiable w.r.t. x, noise, decay, wet."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, noise, decay, wet, t):
        L = _lib.lib()
        x = x.detach().contiguous().float()
        noise, decay, wet, t = (v.detach().contiguous().float() for v in (noise, decay, wet, t))
        rows, n = x.shape
        used = min(noise.numel(), n)                      # taps that survive the pad / crop of :34
        nfft = fft_length(n + used - 1)                   # no wrap-around reaches the first n outputs (nor the backward's lags)
        bins = nfft // 2 + 1
        imp = reverb_impulse(noise, decay, wet, t, used)
        k_spec = torch.view_as_real(torch.fft.rfft(imp, n=nfft)).contiguous()                 # [bins, 2]
        x_spec = torch.view_as_real(torch.fft.rfft(x, n=nfft)).contiguous()                   # [rows, bins, 2]
        y_spec = torch.empty_like(x_spec)
        with torch.cuda.device(x.device):
            _lib.check(L.ddsp_spectral_mul(x_spec.data_ptr(), k_spec.data_ptr(), y_spec.data_ptr(), rows, bins, _stream(x)),
                       "ddsp_spectral_mul")
        y = torch.fft.irfft(torch.view_as_complex(y_spec), n=nfft)[:, :n]
        need_x, need_p = ctx.needs_input_grad[0], any(ctx.needs_input_grad[1:4])
        if need_x or need_p:
            ctx.save_for_backward(x_spec if need_p else None, k_spec, noise, decay, wet, t)
            ctx.n, ctx.nfft = n, nfft
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_y):
        x_spec, k_spec, noise, decay, wet, t = ctx.saved_tensors
        L = _lib.lib()
        n, nfft = ctx.n, ctx.nfft
        rows, bins = grad_y.shape[0], nfft // 2 + 1
        need_x, need_p = ctx.needs_input_grad[0], any(ctx.needs_input_grad[1:4])
        g_spec = torch.view_as_real(torch.fft.rfft(grad_y.contiguous().float(), n=nfft)).contiguous()
        gk = torch.empty_like(g_spec) if need_x else None
        s = torch.empty((bins, 2), device=g_spec.device, dtype=torch.float32) if need_p else None
        with torch.cuda.device(g_spec.device):
            _lib.check(L.ddsp_spectral_mul_backward(g_spec.data_ptr(), None if x_spec is None else x_spec.data_ptr(), k_spec.data_ptr(),
                                                    None if gk is None else gk.data_ptr(), None if s is None else s.data_ptr(),
                                                    rows, bins, _stream(g_spec)), "ddsp_spectral_mul_backward")
        grad_x = torch.fft.irfft(torch.view_as_complex(gk), n=nfft)[:, :n] if need_x else None
        grad_noise = grad_decay = grad_wet = None
        if need_p:
            length = noise.numel()
            used = min(length, n)
            grad_imp = torch.fft.irfft(torch.view_as_complex(s), n=nfft)[:used].contiguous()
            grad_noise = torch.empty_like(noise)
            grad_decay = torch.empty_like(decay)
            grad_wet = torch.empty_like(wet)
            with torch.cuda.device(noise.device):
                _lib.check(L.ddsp_reverb_impulse_backward(grad_imp.data_ptr(), noise.data_ptr(), decay.data_ptr(), wet.data_ptr(),
                                                          t.data_ptr(), grad_noise.data_ptr(), grad_decay.data_ptr(),
                                                          grad_wet.data_ptr(), length, used, _stream(noise)),
                           "ddsp_reverb_impulse_backward")
        return grad_x, grad_noise, grad_decay, grad_wet, None


class Reverb(nn.Module):
    def __init__(self, conf, initial_wet=0, initial_decay=5):
        super().__init__()
        self.length = conf.sample_rate
        self.sampling_rate = conf.sample_rate
        self.noise = nn.Parameter(torch.rand(self.length) * 2 - 1)
        self.decay = nn.Parameter(torch.tensor(float(initial_decay)))
        self.wet = nn.Parameter(torch.tensor(float(initial_wet)))
        seconds = (torch.arange(self.length) / self.sampling_rate).reshape(1, -1)
        self.t = nn.Parameter(seconds, requires_grad=False)
        self.buffer = nn.Parameter(torch.zeros(1, self.length), requires_grad=False)
        self._spare = None       # where the live path writes the slid history (every element moves: not in place)

    def _hip(self, x: torch.Tensor) -> bool:
        return x.is_cuda and x.dtype == torch.float32 and self.noise.is_cuda and self.noise.dtype == torch.float32

    def build_impulse(self):
        if self.noise.is_cuda and self.noise.dtype == torch.float32 and not (
                torch.is_grad_enabled() and any(p.requires_grad for p in (self.noise, self.decay, self.wet))):
            return reverb_impulse(self.noise.detach(), self.decay.detach(), self.wet.detach(), self.t.detach().reshape(-1),
                                  self.length).reshape(1, -1)
        envelope = torch.exp(-F.softplus(-self.decay) * self.t * 500)      # :25
        taps = self.noise * envelope * torch.sigmoid(self.wet)             # :26-27
        return torch.cat([torch.ones_like(taps[:, :1]), taps[:, 1:]], dim=1)  # :28 (out of place: autograd friendly)

    def forward(self, x):
        n = x.shape[1]
        if self._hip(x):
            if x.dim() != 2:
                raise ValueError("expected audio [B, N]")
            return _ReverbFunction.apply(x, self.noise, self.decay, self.wet, self.t.reshape(-1))
        impulse = self.build_impulse()
        impulse = impulse[:, :n] if n < self.length else F.pad(impulse, (0, n - self.length))  # :34 negative pad crops
        return causal_fft_convolve(x, impulse)

    def live_forward(self, x):
        n = x.shape[1]
        if self._hip(x):
            return self._live_hip(x)
        window = torch.cat([self.buffer[:, n:], x], dim=1)                 # :42-44 slide the one-second history
        self.buffer.data.copy_(window)
        return causal_fft_convolve(window, self.build_impulse())[:, -n:]

    def _live_hip(self, x):
        """reverb.py:40-49 on the device without a transform: the last n samples of (slid history) * impulse, directly."""
        if x.dim() != 2 or x.shape[0] != 1:
            raise ValueError("live_forward expects one row [1, n] (reverb.py:22: the history is [1, sample_rate])")
        L = _lib.lib()
        n = x.shape[1]
        x = x.detach().contiguous()
        hist = self.buffer.data
        if self._spare is None or self._spare.device != hist.device or self._spare.shape != hist.shape:
            self._spare = torch.empty_like(hist)
        new = self._spare
        y = torch.empty_like(x)
        scratch = torch.empty(max(1, L.ddsp_reverb_live_scratch_bytes(self.length, n)), device=x.device, dtype=torch.uint8)
        with torch.cuda.device(x.device):
            _lib.check(L.ddsp_reverb_live(x.data_ptr(), hist.data_ptr(), new.data_ptr(), self.noise.data_ptr(), self.decay.data_ptr(),
                                          self.wet.data_ptr(), self.t.data_ptr(), y.data_ptr(), scratch.data_ptr(), self.length, n,
                                          _stream(x)), "ddsp_reverb_live")
        hist.copy_(new)     # the history keeps its address (a captured callback replays with static pointers): one 4L-byte copy
        return y
