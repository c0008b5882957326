"""hipGraph-captured synthesis for launch-bound shapes (small batches, the real-time path).

One `harmonics + noise` pass is 7 short kernel launches; at batch 1 their launch gaps and the Python between
them cost more than the kernels.  `GraphedSynth` captures the pass once for a fixed control shape into a
`torch.cuda.CUDAGraph` (the launch path allocates nothing and never synchronises) and replays it per call;
the caller writes the controls into the static input tensors (`f0`, `c`, `a`, `H`) and reads `out`.

With `live=True` the oscillator runs as `OscillatorBank.live` (harmonic_oscillator.py:64-75): the phase state
of batch row 0 is carried from replay to replay inside the graph (state_out -> state_in copy node).
"""
from __future__ import annotations

import torch

from . import _lib
from .filtered_noise import noise_forward
from .harmonic_oscillator import osc_forward


class GraphedSynth:
    def __init__(self, conf, batch: int, frames: int, n_noise_filters: int, device="cuda", live: bool = False,
                 noise_seed: int = 0):
        self.hop, self.sample_rate, self.live = conf.hop_length, conf.sample_rate, live
        dev = torch.device(device)
        H = conf.n_harmonics
        self.f0 = torch.full((batch, frames, 1), 100.0, device=dev)
        self.c = torch.ones((batch, frames, H), device=dev)
        self.a = torch.ones((batch, frames, 1), device=dev)
        self.H = torch.ones((batch, frames, n_noise_filters), device=dev)
        self.state = torch.zeros(H, device=dev)           # last_phases (fp32, as after the reference's first live call)
        self.seed = noise_seed
        self._step = torch.zeros(1, dtype=torch.int64, device=dev)
        self.out = None
        self._graph = None
        self._capture()

    def _pass(self):
        if self.live:
            y, new_state, _ = osc_forward(self.f0, self.c, self.a, self.hop, self.sample_rate, live_in=self.state,
                                          want_live_out=True)
            self.state.copy_(new_state)
        else:
            y, _, _ = osc_forward(self.f0, self.c, self.a, self.hop, self.sample_rate)
        # the draw is fixed by (seed, offset) at capture time: replays reuse the same noise realisation unless the caller
        # re-seeds and re-captures; pass noise through `H` scaling if a different realisation per call matters
        return noise_forward(self.H, self.hop, seed=self.seed, out=y, accumulate=True)

    def _capture(self):
        if not self.f0.is_cuda:
            raise _lib.DdspHipError("GraphedSynth needs a GPU")
        saved = self.state.clone()
        side = torch.cuda.Stream(device=self.f0.device)
        side.wait_stream(torch.cuda.current_stream(self.f0.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._pass()                      # warm-up: lazy one-time attribute calls, allocator pools
        torch.cuda.current_stream(self.f0.device).wait_stream(side)
        self.state.copy_(saved)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self.out = self._pass()
        self.state.copy_(saved)

    def run(self):
        """Replay the captured pass on the current controls; returns the static output tensor [B, T*hop]."""
        self._graph.replay()
        return self.out

    def __call__(self, ctrl):
        for name in ("f0", "c", "a", "H"):
            getattr(self, name).copy_(ctrl[name], non_blocking=True)
        return self.run()
