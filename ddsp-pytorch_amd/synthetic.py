"""Synthetic control signals for the DDSP synthesis hot path.

These are the inputs SURVEY.md §8(d) / BASELINE.md §4 prescribe for the
benchmark and for the golden fixtures: there is no dataset and no trained
controller in this repository, so `f0`, `c`, `a`, `H` are drawn with the
value ranges the reference's controller produces
(`model/autoencoder/decoder.py:110-116` modified_sigmoid: 2*sigmoid(z)^2.3026 + 1e-7,
 `model/autoencoder/encoder.py:39-48` CREPE pitch grid).

Everything here is NumPy (PCG64 `default_rng`) so that the same seed gives the
same controls in the golden generator (this container), in the CPU tests and on
the GPU box.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class SynthShape:
    """One workload of BASELINE.json `configs` (clip length given in frames)."""
    name: str
    batch: int
    sample_rate: int
    hop: int
    frames: int
    n_harmonics: int
    n_noise_filters: int

    @property
    def samples(self) -> int:
        return self.frames * self.hop


# 4 s clips, hop chosen so that hop >= 2*(F-1)
CFG1 = SynthShape("cfg1_b1_16k_h60", 1, 16000, 128, 500, 60, 65)
CFG2 = SynthShape("cfg2_b64_16k_h100", 64, 16000, 128, 500, 100, 65)
CFG3 = SynthShape("cfg3_b512_48k_h200", 512, 48000, 512, 375, 200, 257)
CFG4_PER_GPU = SynthShape("cfg4_b512pergpu_16k_h100", 512, 16000, 128, 500, 100, 65)


def controller_range(z: np.ndarray) -> np.ndarray:
    """Value range of the reference controller heads (decoder.py:110-116)."""
    s = 1.0 / (1.0 + np.exp(-z.astype(np.float64)))
    return (2.0 * s ** 2.3026 + 1e-7).astype(np.float32)


def all_live_f0(rng: np.random.Generator, batch: int, frames: int,
                sample_rate: int, n_harmonics: int) -> np.ndarray:
    """f0 ~ U[50, 0.98*sr/(2H)] Hz per frame: every harmonic below Nyquist
    (the honest worst case: no partial may be skipped)."""
    hi = 0.98 * sample_rate / (2.0 * n_harmonics)
    lo = min(50.0, 0.5 * hi)
    return rng.uniform(lo, hi, size=(batch, frames, 1)).astype(np.float32)


def musical_f0(rng: np.random.Generator, batch: int, frames: int) -> np.ndarray:
    """Random walk on the CREPE 360-bin grid, 31.7 .. 2005.5 Hz (encoder.py:39-48)."""
    start = rng.integers(60, 300, size=(batch, 1))
    steps = rng.integers(-2, 3, size=(batch, frames))
    bins = np.clip(start + np.cumsum(steps, axis=1), 0, 359)
    cents = 20.0 * bins + 1997.3794084376191
    hz = 10.0 * 2.0 ** (cents / 1200.0)
    return hz.astype(np.float32)[..., None]


def make_controls(shape: SynthShape, seed: int, f0_kind: str = "all_live",
                  batch: int | None = None) -> dict:
    """Control dict with the reference's keys (decoder.py:105): f0 [B,T,1], c [B,T,H], a [B,T,1], H [B,T,F]."""
    rng = np.random.default_rng(seed)
    b = shape.batch if batch is None else batch
    if f0_kind == "all_live":
        f0 = all_live_f0(rng, b, shape.frames, shape.sample_rate, shape.n_harmonics)
    elif f0_kind == "musical":
        f0 = musical_f0(rng, b, shape.frames)
    else:
        raise ValueError(f0_kind)
    c = controller_range(rng.standard_normal((b, shape.frames, shape.n_harmonics), dtype=np.float32))
    a = controller_range(rng.standard_normal((b, shape.frames, 1), dtype=np.float32))
    h = controller_range(rng.standard_normal((b, shape.frames, shape.n_noise_filters), dtype=np.float32))
    return {"f0": f0, "c": c, "a": a, "H": h}
