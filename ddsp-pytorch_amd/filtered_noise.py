"""Drop-in `FilteredNoise` (reference: model/ddsp/filtered_noise.py:35-53) on the HIP kernel.

`forward(x)` reads `x['H'] [B,T,F]` and returns `[B, T*hop]`.  The reference draws its noise with
`torch.rand(B,T,hop)` on the CPU global generator and copies it to the device (:44-48); that stays
the default (`rng='host'`, reproducible with torch.manual_seed exactly like the reference).
`rng='device'` draws inside the kernel (Philox4x32-10; a different stream); `noise=` injects a draw.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


def noise_forward(Hmag, hop: int, uniform=None, seed: int = 0, offset: int = 0, out=None, accumulate=False, counter=None):
    """Raw launcher over the C ABI (include/ddsp_hip.h: ddsp_noise_forward).
    `counter`: a 1-element int64 CUDA tensor holding the Philox offset of the in-kernel draw (read on the device at
    launch time, so a captured launch draws from wherever the counter stands at each replay); excludes `uniform`."""
    if Hmag.dim() != 3:
        raise ValueError("expected H [B,T,F]")
    if not Hmag.is_cuda:
        raise _lib.DdspHipError("FilteredNoise runs on the GPU only (no CPU fallback): move the controls to cuda")
    Hmag = Hmag.detach().contiguous().float()
    B, T, F = Hmag.shape
    if out is None:
        out = torch.empty((B, T * hop), device=Hmag.device, dtype=torch.float32)
        accumulate = False
    elif out.shape != (B, T * hop) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous fp32 [B, T*hop] tensor")
    if B == 0:
        return out
    if uniform is not None:
        if tuple(uniform.shape) != (B, T, hop):
            raise ValueError(f"uniform must be [B,T,hop] = {(B, T, hop)}, got {tuple(uniform.shape)}")
        uniform = uniform.detach().to(device=Hmag.device, dtype=torch.float32).contiguous()
    if counter is not None and (uniform is not None or counter.dtype != torch.int64 or counter.numel() != 1 or not counter.is_cuda):
        raise ValueError("counter must be a 1-element int64 CUDA tensor and excludes an injected draw")
    with torch.cuda.device(Hmag.device):
        stream = torch.cuda.current_stream().cuda_stream
        ws_bytes = _lib.lib().ddsp_noise_workspace_bytes(B, T, F, hop)
        if ws_bytes:
            # (the reference's default shape: impulse responses of the whole batch as one matrix-core product, include/ddsp_hip.h)
            ws = torch.empty(ws_bytes, device=Hmag.device, dtype=torch.uint8)
            rc = _lib.lib().ddsp_noise_forward_ws(Hmag.data_ptr(), None if uniform is None else uniform.data_ptr(), out.data_ptr(),
                                                  B, T, F, hop, seed, offset, None if counter is None else counter.data_ptr(),
                                                  1 if accumulate else 0, ws.data_ptr(), ws_bytes, stream)
        elif counter is not None:
            rc = _lib.lib().ddsp_noise_forward_counter(Hmag.data_ptr(), out.data_ptr(), B, T, F, hop, seed, counter.data_ptr(),
                                                       1 if accumulate else 0, stream)
        else:
            rc = _lib.lib().ddsp_noise_forward(Hmag.data_ptr(), None if uniform is None else uniform.data_ptr(),
                                               out.data_ptr(), B, T, F, hop, seed, offset, 1 if accumulate else 0, stream)
    _lib.check(rc, "ddsp_noise_forward")
    return out


def noise_backward(grad_y, hop: int, n_filters: int, uniform=None, seed: int = 0, offset: int = 0, counter=None):
    """Raw launcher of ddsp_noise_backward: grad_y [B,T*hop] -> grad_H [B,T,F] for the same draw as the forward
    (`counter`: the device counter the forward read, still at the same value)."""
    grad_y = grad_y.detach().contiguous().float()
    B = grad_y.shape[0]
    T = grad_y.shape[1] // hop
    grad_h = torch.empty((B, T, n_filters), device=grad_y.device, dtype=torch.float32)
    if B == 0:
        return grad_h
    with torch.cuda.device(grad_y.device):
        stream = torch.cuda.current_stream().cuda_stream
        ws_bytes = _lib.lib().ddsp_noise_workspace_bytes(B, T, n_filters, hop)
        if ws_bytes:
            ws = torch.empty(ws_bytes, device=grad_y.device, dtype=torch.uint8)
            rc = _lib.lib().ddsp_noise_backward_ws(grad_y.data_ptr(), None if uniform is None else uniform.data_ptr(), grad_h.data_ptr(),
                                                   B, T, n_filters, hop, seed, offset, None if counter is None else counter.data_ptr(),
                                                   ws.data_ptr(), ws_bytes, stream)
        elif counter is not None:
            rc = _lib.lib().ddsp_noise_backward_counter(grad_y.data_ptr(), grad_h.data_ptr(), B, T, n_filters, hop, seed,
                                                        counter.data_ptr(), stream)
        else:
            rc = _lib.lib().ddsp_noise_backward(grad_y.data_ptr(), None if uniform is None else uniform.data_ptr(),
                                                grad_h.data_ptr(), B, T, n_filters, hop, seed, offset, stream)
    _lib.check(rc, "ddsp_noise_backward")
    return grad_h


def calibrate_noise_residency(step=None, levels=(3, 4, 5, 6, 7, 8), settle_ms: float = 70.0, measure_ms: float = 40.0):
    """Measures on THIS GPU how many wavefronts per CU the hop-128 noise kernel can take before its power density makes the chip drop
    its shader clock (include/ddsp_hip.h: ddsp_noise_set_residency), and sets the fastest.

    `step`: zero-argument callable that issues the caller's real workload on the current stream (default: the 16 kHz / 100 harmonics /
    65 bands / batch 512 synthesis step on synthetic controls).  For each level, ascending (a level that trips the clock pollutes the
    ~25 ms after it): `settle_ms` of back-to-back steps untimed, then `measure_ms` timed.  -> {"chosen": n, "ms_per_step": {n: ms}}.
    ~0.7 s with the defaults; results of the kernels do not depend on the setting."""
    import time
    L = _lib.lib()
    if step is None:
        from . import synthetic as syn
        from .harmonic_oscillator import osc_forward
        shape = syn.CFG4_PER_GPU
        ctl = {k: torch.from_numpy(v).cuda() for k, v in syn.make_controls(shape, 1, "all_live").items()}
        count = [0]

        def step():
            y = osc_forward(ctl["f0"], ctl["c"], ctl["a"], shape.hop, shape.sample_rate)[0]
            noise_forward(ctl["H"], shape.hop, seed=1, offset=count[0] << 32, out=y, accumulate=True)
            count[0] += 1

    def run_for(ms):
        n, t0 = 0, time.perf_counter()
        while 1e3 * (time.perf_counter() - t0) < ms:
            for _ in range(4):
                step()
                n += 1
            torch.cuda.synchronize()
        return n, time.perf_counter() - t0

    before = L.ddsp_noise_get_residency()
    timings = {}
    try:
        for level in sorted(levels):
            _lib.check(L.ddsp_noise_set_residency(int(level)), "ddsp_noise_set_residency")
            run_for(settle_ms)
            n, el = run_for(measure_ms)
            timings[int(level)] = 1e3 * el / n
    except BaseException:
        L.ddsp_noise_set_residency(before)
        raise
    best = min(timings.values())
    chosen = min(k for k, v in timings.items() if v <= 1.004 * best)       # (the lowest level within 0.4 % of the best)
    _lib.check(L.ddsp_noise_set_residency(chosen), "ddsp_noise_set_residency")
    return {"chosen": chosen, "ms_per_step": timings}


class _NoiseFunction(torch.autograd.Function):
    """Differentiable w.r.t. H; the noise draw is a constant of the graph (filtered_noise.py:44-48)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, Hmag, uniform, hop, seed, offset, counter=None):
        if uniform is not None:
            uniform = uniform.detach().to(device=Hmag.device, dtype=torch.float32).contiguous()
        y = noise_forward(Hmag, hop, uniform=uniform, seed=seed, offset=offset, counter=counter)
        ctx.save_for_backward(uniform)
        ctx.counter = counter        # read again by the backward: the owner advances it only after that (GraphedTrainStep)
        ctx.meta = (hop, Hmag.shape[-1], seed, offset)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_y):
        (uniform,) = ctx.saved_tensors
        hop, nf, seed, offset = ctx.meta
        return noise_backward(grad_y, hop, nf, uniform=uniform, seed=seed, offset=offset, counter=ctx.counter), None, None, None, None, None


class FilteredNoise(nn.Module):
    def __init__(self, conf, rng: str = 'host', seed: int = 0):
        super().__init__()
        self.block_size = conf.hop_length
        if rng not in ('host', 'device'):
            raise ValueError("rng must be 'host' (torch CPU generator, reference-compatible) or 'device' (Philox)")
        self.rng = rng
        self.seed = seed
        # Philox offset of the next in-kernel draw: advanced by what each call consumes (B*T*ceil(hop/4) counters), so
        # calls of different shapes (a last partial batch, train/eval switches) never overlap earlier draws.  Not part of
        # the state_dict (the reference has no such state): a resumed run that must not replay the stream passes a new `seed`.
        self._offset = 0
        # hipGraph-captured training steps (graphed.GraphedTrainStep): the offset lives in this 1-element int64 CUDA tensor, read by
        # the forward AND the backward kernel at run time and advanced by a node of the graph after both
        self.counter = None
        self._last_draws = 0

    def reseed(self, seed: int, offset: int = 0) -> None:
        """Restart the in-kernel (rng='device') stream: a resumed training run passes a fresh seed (or the offset it saved)
        so that it does not replay the draws of its first steps."""
        self.seed, self._offset = int(seed), int(offset)

    def forward(self, x, noise=None, out=None):
        """`out` (inference only): accumulate the noise into this [B, T*hop] buffer instead of returning a new one."""
        param = x['H']
        B, T, _ = param.shape
        if noise is None and self.rng == 'host':
            noise = torch.rand(B, T, self.block_size)  # :44-48: CPU global generator, same shape and order
        offset, counter = 0, None
        if noise is None:
            if self.counter is not None:
                counter = self.counter
                self._last_draws = self.draws(B, T)     # what the owner of the counter adds after the backward
            else:
                offset = self._offset
                self._offset += self.draws(B, T)
        if torch.is_grad_enabled() and param.requires_grad:
            if not param.is_cuda:
                raise _lib.DdspHipError("FilteredNoise runs on the GPU only (no CPU fallback): move the controls to cuda")
            return _NoiseFunction.apply(param, noise, self.block_size, self.seed, offset, counter)
        return noise_forward(param, self.block_size, uniform=noise, seed=self.seed, offset=offset, out=out,
                             accumulate=out is not None, counter=counter)

    def draws(self, batch: int, frames: int) -> int:
        """Philox counters one in-kernel draw of this shape consumes."""
        return batch * frames * ((self.block_size + 3) // 4)
