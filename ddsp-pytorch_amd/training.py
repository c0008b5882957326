"""Training-step pieces for BASELINE.json configs[4]: multi-scale spectral loss
and a data-parallel step with ONE flat RCCL all-reduce of the gradients.

* `MSSLoss` restates loss/mss_loss.py:11-68.  The reference builds its spectrograms with
  torchaudio.transforms.Spectrogram (pinned torchaudio==0.8.1, requirements.txt:111), which is not
  installed here: this is the documented equivalent on torch.stft -- **parity unpinned** (no reference
  fixture can be produced for it).
* The reference trains on a single GPU (train/train.py:50); the data-parallel step is new design:
  replicas, per-rank batch shard, gradients flattened into one bucket (19.35 MB for the 16 kHz/100/65
  decoder) and averaged with a single all_reduce -- on 8 MI355X a ring moves 2*7/8 of the bucket per GPU over
  xGMI, latency- not bandwidth-bound, so one bucket beats many.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class SpectralLoss(nn.Module):
    """One scale: L1 of power spectrograms + alpha * L1 of their log2 (loss/mss_loss.py:11-33)."""

    def __init__(self, n_fft: int, alpha: float = 1.0, overlap: float = 0.75, eps: float = 1e-7):
        super().__init__()
        self.n_fft, self.alpha, self.eps = n_fft, alpha, eps
        self.hop = int(n_fft * (1 - overlap))
        self.register_buffer("window", torch.hann_window(n_fft), persistent=False)

    def power(self, x: torch.Tensor) -> torch.Tensor:
        spec = torch.stft(x, self.n_fft, hop_length=self.hop, window=self.window.to(x.device), center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        return spec.real.square() + spec.imag.square()

    def forward(self, x_pred, x_true):
        s_true, s_pred = self.power(x_true), self.power(x_pred)
        linear = F.l1_loss(s_pred, s_true)
        log = F.l1_loss(torch.log2(s_true + self.eps), torch.log2(s_pred + self.eps))
        return linear + self.alpha * log


class MSSLoss(nn.Module):
    """Sum of SpectralLoss over FFT sizes; the trainer uses (2048, 1024, 512, 256, 128, 64) (train/train.py:19)."""

    def __init__(self, n_ffts=(2048, 1024, 512, 256, 128, 64), alpha=1.0, overlap=0.75, eps=1e-7):
        super().__init__()
        self.losses = nn.ModuleList([SpectralLoss(n, alpha, overlap, eps) for n in n_ffts])

    def forward(self, x_pred, x_true):
        if isinstance(x_true, dict):
            x_true = x_true["audio"]
        return sum(loss(x_pred, x_true) for loss in self.losses)


def allreduce_gradients(params, group=None) -> int:
    """Average gradients over the process group with one flat all_reduce. Returns the bucket size in bytes."""
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= dist.get_world_size(group)
    offset = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[offset:offset + n].view_as(g))
        offset += n
    return flat.numel() * flat.element_size()


def train_step(model: nn.Module, loss_fn: nn.Module, optimizer: torch.optim.Optimizer, batch, group=None):
    """One optimisation step of `Zak.training_step` (train/train.py:32-37) + Adam, data-parallel.
    `batch` is this rank's shard: a dict with the controller inputs and the target `audio`."""
    optimizer.zero_grad(set_to_none=True)
    audio = model(batch)
    loss = loss_fn(audio, batch)
    loss.backward()
    nbytes = allreduce_gradients([p for p in model.parameters() if p.requires_grad], group)
    optimizer.step()
    return loss.detach(), nbytes
