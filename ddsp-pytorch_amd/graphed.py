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
from . import dense
from .filtered_noise import FilteredNoise, noise_forward
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
        self._step = torch.zeros(1, dtype=torch.int64, device=dev)     # Philox offset of the next call's draw
        self._draws_per_call = batch * frames * ((conf.hop_length + 3) // 4)
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
        # the Philox offset of the draw lives on the device and advances inside the graph: every replay is fresh noise,
        # the same sequence as un-captured calls with offset = call index x draws per call
        y = noise_forward(self.H, self.hop, seed=self.seed, out=y, accumulate=True, counter=self._step)
        self._step.add_(self._draws_per_call)
        return y

    def _capture(self):
        if not self.f0.is_cuda:
            raise _lib.DdspHipError("GraphedSynth needs a GPU")
        saved = self.state.clone()
        self._step.zero_()
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
        self._step.zero_()

    def run(self):
        """Replay the captured pass on the current controls; returns the static output tensor [B, T*hop]."""
        self._graph.replay()
        return self.out

    def __call__(self, ctrl):
        for name in ("f0", "c", "a", "H"):
            getattr(self, name).copy_(ctrl[name], non_blocking=True)
        return self.run()


class GraphedLiveDecoder:
    """The whole real-time callback (rt/synth.py:40-55 -> `Decoder.forward_live`, decoder.py:139-147) as ONE hipGraph:
    controller (MLPs, GRU recurrence, heads) -> harmonics.live + noise -> reverb.live_forward.  Per callback the host does
    three small H2D copies into the static inputs, one replay and one D2H of the audio.  Every piece of state stays on
    the device and is advanced by nodes of the graph: the oscillator phases (`state`), the reverb's one-second history
    (`decoder.reverb.buffer`), the Philox offset of the noise draw, and the GRU state (`hidden_out`; the reference hands
    the callback's INPUT state back, SURVEY App. C.7 -- `carry_hidden=True` feeds the new one forward instead).
    """

    def __init__(self, decoder, frames: int, noise_seed: int = 0, carry_hidden: bool = False):
        self.dec = decoder.eval()
        dev = next(decoder.parameters()).device
        if dev.type != "cuda":
            raise _lib.DdspHipError("GraphedLiveDecoder needs the decoder on a GPU")
        osc = decoder.harmonics
        self.hop, self.sample_rate = osc.hop_size, osc.sample_rate
        self.frames, self.seed, self.carry_hidden = frames, noise_seed, carry_hidden
        self.z = {k: torch.zeros(1, frames, 1, device=dev) for k in ("normalized_cents", "loudness", "f0")}
        self.z["f0"].fill_(100.0)
        units = decoder.controller.gru.hidden_size
        self.hidden = torch.zeros(1, 1, units, device=dev)
        self.hidden_out = torch.zeros(1, 1, units, device=dev)
        self.state = osc.last_phases.data.detach().to(device=dev, dtype=torch.float32).clone()
        self._step = torch.zeros(1, dtype=torch.int64, device=dev)
        self._draws_per_call = frames * ((self.hop + 3) // 4)
        self.out = None
        self._capture()

    def _pass(self):
        with torch.no_grad():
            ctrl, _ = self.dec.controller(self.z, self.hidden)
            y, new_state, _ = osc_forward(ctrl["f0"], ctrl["c"], ctrl["a"], self.hop, self.sample_rate, live_in=self.state,
                                          want_live_out=True)
            self.state.copy_(new_state)
            noise_forward(ctrl["H"], self.hop, seed=self.seed, out=y, accumulate=True, counter=self._step)
            self._step.add_(self._draws_per_call)
            audio = self.dec.reverb.live_forward(y)
            self.hidden_out.copy_(ctrl["hidden"])
            if self.carry_hidden:
                self.hidden.copy_(ctrl["hidden"])
            return audio.contiguous()

    def _capture(self):
        dev = self.state.device
        keep = (self.state.clone(), self.dec.reverb.buffer.data.clone(), self.hidden.clone())

        def restore():
            self.state.copy_(keep[0])
            self.dec.reverb.buffer.data.copy_(keep[1])
            self.hidden.copy_(keep[2])
            self._step.zero_()

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._pass()                      # warm-up: FFT plans, library handles, allocator pools
        torch.cuda.current_stream(dev).wait_stream(side)
        restore()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self.out = self._pass()
        restore()

    def run(self, z):
        """z: dict of `normalized_cents`, `loudness`, `f0`, each [1, frames, 1] (numpy or tensor) -> audio [frames*hop] (numpy)."""
        for k, buf in self.z.items():
            v = z[k]
            buf.copy_(v if torch.is_tensor(v) else torch.from_numpy(v), non_blocking=True)
        self._graph.replay()
        return self.out.cpu().squeeze(0).numpy()


class GraphedTrainStep:
    """`training.train_step` for a FIXED batch shape as hipGraph replays: forward, spectral loss, backward and the optimiser
    update of one step are captured once (about 280 launches at the training shape, whose host-side issue time had become as
    long as the GPU's work); a step is then the copies of the batch into the static input tensors and one replay (with more
    than one rank: replay forward + backward, ONE eager flat all-reduce, replay the update).

    * `optimizer` must be capturable (`torch.optim.Adam(..., capturable=True)`; `fused=True` as well if wanted).
    * Constructing the object does not advance training: the warm-up steps capture needs (library handles, FFT plans,
      allocator pools, optimiser state) are undone -- parameters, buffers and optimiser state are put back in place; state the
      optimiser did not have yet is zeroed, which is what Adam-style optimisers start from.
    * A `FilteredNoise(rng='device')` inside the model draws from a device-resident Philox counter from here on (read by the
      forward and the backward kernel, advanced by a node of the graph): every replay gets fresh noise, the same sequence as
      the eager steps; `step()` also advances the modules' host-side offsets by the same amount, so eager calls of the model
      between graphed steps continue the stream.
    * fp32 or `amp_dtype=torch.bfloat16` (no GradScaler: fp16 needs its host-side decisions); every rank must hold rows.
    * The persistent GRU launches inside a graph are ordered by the graph only: do not run another recurrence on a second
      stream of the same device while a replay is in flight (DESIGN.md par. 9a).
    """

    def __init__(self, model, loss_fn, optimizer, example_batch, group=None, amp_dtype=None, warmup: int = 3):
        import torch.distributed as dist
        self.model, self.loss_fn, self.opt, self.group, self.amp_dtype = model, loss_fn, optimizer, group, amp_dtype
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params or not self.params[0].is_cuda:
            raise _lib.DdspHipError("GraphedTrainStep needs the model on a GPU")
        if not all(g.get("capturable", True) for g in optimizer.param_groups):     # (optimisers without the option, e.g. SGD, never synchronise)
            raise ValueError("GraphedTrainStep needs a capturable optimiser, e.g. torch.optim.Adam(params, capturable=True)")
        dev = self.params[0].device
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.batch = {k: (v.detach().to(dev).clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        if next((v.shape[0] for v in self.batch.values() if torch.is_tensor(v)), 0) == 0:
            raise ValueError("GraphedTrainStep cannot capture an empty shard")
        self.noises = [m for m in model.modules() if isinstance(m, FilteredNoise) and m.rng == "device"]
        # (the modules see their counter only inside this object's steps; eager calls keep the host-side offset)
        self.counters = [torch.tensor([m._offset], dtype=torch.int64, device=dev) for m in self.noises]
        self._mirror = [int(m._offset) for m in self.noises]    # what each device counter holds (host-side knowledge)
        self.loss = None
        self.nbytes = sum(p.numel() * p.element_size() for p in self.params)
        self._capture(dev, warmup)

    # one step in two halves (a collective sits between them when there is more than one rank)
    def _forward_backward(self):
        for m, c in zip(self.noises, self.counters):
            m.counter, m._last_draws = c, 0
        try:
            if self.amp_dtype is not None:
                dense.lowp_weights.refresh(self.amp_dtype)
            with torch.autocast("cuda", dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None):
                audio = self.model(self.batch)
            dense.lowp_weights.release()               # (the copies the captured forward uses are refreshed by the graph itself)
            loss = self.loss_fn(audio.float(), self.batch)
            loss.backward()
            self._draws = [m._last_draws for m in self.noises]     # host constants of the captured shape
            for m, c in zip(self.noises, self.counters):
                if m._last_draws:                      # (0: the caller injected its own draw, the counter was not read)
                    c.add_(m._last_draws)
        finally:
            dense.lowp_weights.release()
            for m in self.noises:
                m.counter = None
        return loss.detach()

    def _reduce(self):
        import torch.distributed as dist
        grads = [p.grad for p in self.params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat /= self.world
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])

    def _capture(self, dev, warmup):
        model_state = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        had = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.opt.state.get(p, {}).items()}
               for g in self.opt.param_groups for p in g["params"]}
        counters = [c.clone() for c in self.counters]

        def restore():
            with torch.no_grad():
                live = self.model.state_dict()
                for k, v in model_state.items():
                    live[k].copy_(v)
                for g in self.opt.param_groups:
                    for p in g["params"]:
                        for k, v in self.opt.state.get(p, {}).items():
                            if torch.is_tensor(v):
                                old = had[id(p)].get(k)
                                v.copy_(old) if old is not None else v.zero_()
                for live_c, c in zip(self.counters, counters):
                    live_c.copy_(c)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.opt.zero_grad(set_to_none=True)
                self._forward_backward()
                if self.world > 1:
                    self._reduce()
                self.opt.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        restore()
        self.opt.zero_grad(set_to_none=True)          # the captured backward allocates the gradients inside the graph's pool
        self._graph = torch.cuda.CUDAGraph()
        self._graph_update = None
        with torch.cuda.graph(self._graph):
            self.loss = self._forward_backward()
            if self.world == 1:
                self.opt.step()
        if self.world > 1:
            self._graph_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_update, pool=self._graph.pool()):
                self.opt.step()
        restore()

    def step(self, batch=None):
        """One optimisation step on `batch` (same keys and shapes as the example; None = whatever the static inputs hold).
        Returns (loss tensor -- static, overwritten by the next step --, all-reduce bytes)."""
        if batch is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.batch[k].copy_(v, non_blocking=True)
        # eager calls of the same model between graphed steps (validation, an eager train_step) advance only the modules'
        # host-side offsets: bring the device counters up to them first, so that this replay does not redraw that noise
        for i, (m, c) in enumerate(zip(self.noises, self.counters)):
            if int(m._offset) != self._mirror[i]:
                c.fill_(int(m._offset))                 # an async fill from a host integer, outside the graph
                self._mirror[i] = int(m._offset)
        self._graph.replay()
        if self.world > 1:
            self._reduce()
            self._graph_update.replay()
        # ... and keep the host-side offsets in step with the device counters, so that an eager call after graphed steps
        # continues the stream instead of replaying its start
        for i, (m, d) in enumerate(zip(self.noises, self._draws)):
            self._mirror[i] += d
            m._offset = self._mirror[i]
        return self.loss, self.nbytes

    __call__ = step
