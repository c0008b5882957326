"""Drop-in `Reverb` (reference: model/ddsp/reverb.py:8-49) -- SURVEY §8(f) next row 1.

Learned exponentially-decaying noise impulse (one second long) applied to the whole clip.  The
impulse build and the truncated causal convolution run on the device; the 2N-point real FFTs go
through torch.fft (rocFFT) -- a library transform, not a hand-written kernel.
Same constructor, parameter names/shapes (`noise`, `decay`, `wet`, `t`, `buffer`: checkpoint
compatible), `forward(x)` and `live_forward(x)` semantics, including the crop of the impulse for
clips shorter than one second (:34) and tap 0 forced to 1 (:28).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def causal_fft_convolve(signal: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """First len(signal) samples of the linear convolution along the last axis
    (what filtered_noise.py:25-32 `fft_convolve` returns)."""
    n = signal.shape[-1]
    spec = torch.fft.rfft(signal, n=2 * n) * torch.fft.rfft(kernel, n=2 * n)
    return torch.fft.irfft(spec, n=2 * n)[..., :n]


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

    def build_impulse(self):
        envelope = torch.exp(-F.softplus(-self.decay) * self.t * 500)      # :25
        taps = self.noise * envelope * torch.sigmoid(self.wet)             # :26-27
        return torch.cat([torch.ones_like(taps[:, :1]), taps[:, 1:]], dim=1)  # :28 (out of place: autograd friendly)

    def forward(self, x):
        n = x.shape[1]
        impulse = self.build_impulse()
        impulse = impulse[:, :n] if n < self.length else F.pad(impulse, (0, n - self.length))  # :34 negative pad crops
        return causal_fft_convolve(x, impulse)

    def live_forward(self, x):
        n = x.shape[1]
        window = torch.cat([self.buffer[:, n:], x], dim=1)                 # :42-44 slide the one-second history
        self.buffer.data.copy_(window)
        return causal_fft_convolve(window, self.build_impulse())[:, -n:]
