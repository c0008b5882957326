"""MI355X-native DDSP synthesis hot path (harmonic oscillator bank + filtered noise).

Drop-in for `model/ddsp/harmonic_oscillator.py` and `model/ddsp/filtered_noise.py`
of kureta/ddsp-pytorch: same constructors, `forward()` / `live()` signatures,
control-dict keys and state-dict keys, with the compute in hand-written HIP
kernels (gfx950) behind the C ABI declared in `include/ddsp_hip.h`.
"""
from . import synthetic  # noqa: F401
from . import _lib  # noqa: F401
from . import sharding  # noqa: F401
from . import training  # noqa: F401
from .harmonic_oscillator import OscillatorBank, osc_forward, osc_backward  # noqa: F401
from .filtered_noise import FilteredNoise, noise_forward, noise_backward, calibrate_noise_residency  # noqa: F401
from .reverb import Reverb, causal_fft_convolve  # noqa: F401
from .graphed import GraphedSynth, GraphedLiveDecoder, GraphedTrainStep  # noqa: F401
from .gru import GRU, gru_forward, gru_backward, gru_status  # noqa: F401
from .decoder import Controller, Decoder  # noqa: F401
from .training import MSSLoss, train_step, allreduce_gradients, OverlappedGradientReducer  # noqa: F401

__all__ = ["OscillatorBank", "FilteredNoise", "Reverb", "causal_fft_convolve", "GraphedSynth", "GraphedLiveDecoder", "GraphedTrainStep", "Controller", "Decoder", "GRU", "MSSLoss", "train_step", "allreduce_gradients", "OverlappedGradientReducer", "osc_forward", "osc_backward", "noise_forward", "noise_backward", "calibrate_noise_residency",
           "synthetic"]
