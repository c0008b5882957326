"""TEST INFRASTRUCTURE ONLY -- the reference's CPU path restated as plain torch ops.

This is what `bench.py` times as `cpu_baseline` (kind "port") on the GPU box's host cores: the
same torch-op sequence the reference runs on the CPU (interpolate, cumsum, remainder, sin, sum;
irfft/rfft), written as functions from SURVEY.md §2.1 / Appendix A rather than copied, and pinned
bit-for-bit against tests/golden/ (tests/test_oracle_golden.py).  The reference itself cannot
travel to the GPU box.  Nothing under ddsp-pytorch_amd/ may import this module.

Reference call sites: model/ddsp/harmonic_oscillator.py:24-75, model/ddsp/filtered_noise.py:7-53.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

TWO_PI = 2 * math.pi


def _upsample(x: torch.Tensor, hop: int) -> torch.Tensor:
    # harmonic_oscillator.py:52-55 -- [B,T,C] -> [B,T*hop,C], half-pixel linear, edge clamped
    return F.interpolate(x.transpose(1, 2), scale_factor=hop, mode="linear").transpose(1, 2)


def oscillator_bank(f0, c, a, hop: int, sample_rate: int, live_phase=None, return_phases: bool = False):
    """f0 [B,T,1], c [B,T,H], a [B,T,1] -> y [B,T*hop].  With `live_phase` [H] (updated in place) this is
    `.live` (:64-75): the offsets are added to the first increment row of batch row 0."""
    n_harm = c.shape[-1]
    k = torch.arange(1, n_harm + 1)                       # int64, like the reference's parameter (:15-18)
    hz = k * f0                                           # :26-29
    amp = c.masked_fill(hz > sample_rate // 2, 0.0)       # :31-32
    amp /= amp.sum(-1, keepdim=True)                      # :33
    hz *= TWO_PI                                          # :34
    hz /= sample_rate                                     # :35
    inc = _upsample(hz, hop)                              # :36
    if live_phase is not None:
        inc[0, 0, :] += live_phase                        # :70
    ph = torch.cumsum(inc, dim=1)                         # :41
    ph %= TWO_PI                                          # :42
    if live_phase is not None:
        live_phase.copy_(ph[0, -1, :])                    # :72
    y = (_upsample(a, hop) * _upsample(amp, hop) * torch.sin(ph)).sum(dim=2)  # :46-49
    return (y, ph) if return_phases else y


def impulse_response(mag, target: int):
    # filtered_noise.py:7-22
    ir = torch.fft.irfft(torch.complex(mag, torch.zeros_like(mag)))
    size = ir.shape[-1]
    ir = torch.roll(ir, size // 2, -1) * torch.hann_window(size, dtype=ir.dtype)
    ir = F.pad(ir, (0, int(target) - int(size)))
    return torch.roll(ir, -size // 2, -1)


def causal_block_convolve(signal, kernel):
    # filtered_noise.py:25-32: first len(signal) samples of the linear convolution
    n = signal.shape[-1]
    spec = torch.fft.rfft(F.pad(signal, (0, n))) * torch.fft.rfft(F.pad(kernel, (kernel.shape[-1], 0)))
    out = torch.fft.irfft(spec)
    return out[..., out.shape[-1] // 2:]


def filtered_noise(mag, hop: int, uniform=None):
    """mag [B,T,F] -> [B,T*hop]; `uniform` [B,T,hop] injects the U[0,1) draw, else torch.rand (global CPU RNG, :44-48)."""
    ir = impulse_response(mag, hop)
    if uniform is None:
        uniform = torch.rand(ir.shape[0], ir.shape[1], hop)
    x = uniform.to(ir) * 2 - 1
    y = causal_block_convolve(x, ir).contiguous()
    return y.reshape(y.shape[0], -1)
