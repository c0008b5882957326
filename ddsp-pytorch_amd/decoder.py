"""Callers of the hot path: the control network and the `Decoder`
wiring of model/autoencoder/decoder.py:41-147, restated with stock torch.nn layers (rocBLAS does the dense
work) around the HIP synth modules of this package; the GRU's recurrence runs on the persistent HIP kernel of
`gru.py` (MIOpen's per-step launches were 85 % of the training step).

Sub-module and parameter names follow the reference so that its checkpoints load with strict=True
(`rt/utils.py:7-24` strips the `model.` prefix, `rt/synth.py:17` loads into `zak.decoder`):
  controller.mlp_f0 / mlp_loudness / gru / mlp_gru / dense_harmonic / dense_loudness / dense_filter,
  harmonics.{harmonics,last_phases}, reverb.{noise,decay,wet,t,buffer}.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import dense
from .filtered_noise import FilteredNoise
from .gru import GRU
from .harmonic_oscillator import OscillatorBank
from .reverb import Reverb


def _dense_stack(n_in: int, width: int, depth: int) -> nn.Module:
    """`depth` x (Linear -> LayerNorm -> LeakyReLU), registered as mlp_layer1..N like decoder.py:9-39."""
    stack = nn.Module()
    for i in range(depth):
        block = nn.Sequential(nn.Linear(n_in if i == 0 else width, width), nn.LayerNorm(width), nn.LeakyReLU())
        stack.add_module(f"mlp_layer{i + 1}", block)
    stack.depth = depth
    return stack


_LN_IO = {torch.bfloat16: 1, torch.float16: 2}


def _ln_forward(x, g, b, eps, slope):
    """x [.., D] contiguous (fp32 / bf16 / fp16), g / b fp32 -> (y like x, mean, rstd): include/ddsp_hip.h ddsp_ln_lrelu_forward*."""
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        if x.dtype == torch.float32:
            rc = L.ddsp_ln_lrelu_forward(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         rows, D, float(eps), float(slope), stream)
        else:
            rc = L.ddsp_ln_lrelu_forward_16(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            rows, D, float(eps), float(slope), _LN_IO[x.dtype], stream)
    _lib.check(rc, "ddsp_ln_lrelu_forward")
    return y, mean, rstd


def _ln_backward(gy, x, y, g, mean, rstd, slope, want_xsum):
    """-> (grad_x like x, d gamma, d beta, column sums of grad_x or None), parameter gradients fp32."""
    D = x.shape[-1]
    rows = x.numel() // D
    gy = gy.contiguous().to(x.dtype)
    gx = torch.empty_like(x)
    out = torch.empty((3, D), device=x.device, dtype=torch.float32)
    xsum = out[2] if want_xsum else None
    L = _lib.lib()
    scratch = torch.empty(L.ddsp_ln_lrelu_scratch_bytes(D), device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        args = (gy.data_ptr(), x.data_ptr(), y.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gx.data_ptr(), out[0].data_ptr(),
                out[1].data_ptr(), None if xsum is None else xsum.data_ptr(), scratch.data_ptr(), rows, D, float(slope))
        if x.dtype == torch.float32:
            rc = L.ddsp_ln_lrelu_backward(*args, stream)
        else:
            rc = L.ddsp_ln_lrelu_backward_16(*args, _LN_IO[x.dtype], stream)
    _lib.check(rc, "ddsp_ln_lrelu_backward")
    return gx, out[0], out[1], xsum


class _LayerNormLeakyReLU(torch.autograd.Function):
    """LayerNorm -> LeakyReLU of one MLP block as one HIP pass each way (include/ddsp_hip.h: ddsp_ln_lrelu_*).
    fp32 activations take the fp32 entry points; bf16 / fp16 activations (torch.autocast: the Linear in front produced them
    and the Linear behind wants them) are read and written as such by the `_16` entry points -- no cast pass either side;
    statistics, gamma / beta and their gradients are fp32 in both cases."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, gamma, beta, eps, slope):
        x = x.contiguous()
        g, b = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
        y, mean, rstd = _ln_forward(x, g, b, eps, slope)
        ctx.save_for_backward(x, y, g, mean, rstd)
        ctx.slope = float(slope)
        ctx.param_dtypes = (gamma.dtype, beta.dtype)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, y, g, mean, rstd = ctx.saved_tensors
        gx, dg, db, _ = _ln_backward(gy, x, y, g, mean, rstd, ctx.slope, False)
        return gx, dg.to(ctx.param_dtypes[0]), db.to(ctx.param_dtypes[1]), None, None


class _LinearBlock(torch.autograd.Function):
    """Linear -> LayerNorm -> LeakyReLU of one MLP block (decoder.py:9-39) as ONE autograd node: the library GEMMs of dense._Linear
    around the fused LayerNorm pass, whose backward also returns the column sums of its input gradient -- the Linear's bias
    gradient, which therefore costs no pass (dense.colsum: two launches per layer) of its own."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, eps, slope):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if dt not in (torch.float32, torch.bfloat16, torch.float16):
            dt = torch.float32
        xc, wc = x.to(dt), dense._cast(weight, dt)
        with torch.autocast("cuda", enabled=False):
            h = F.linear(xc, wc, dense._cast(bias, dt)).contiguous()
        g, b = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
        y, mean, rstd = _ln_forward(h, g, b, eps, slope)
        ctx.save_for_backward(xc, wc, h, y, g, mean, rstd)
        ctx.slope = float(slope)
        ctx.meta = (x.dtype, weight.dtype, bias.dtype, gamma.dtype, beta.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, wc, h, y, g, mean, rstd = ctx.saved_tensors
        xdt, wdt, cdt, gdt, bdt = ctx.meta
        gh, dg, db, xsum = _ln_backward(gy, h, y, g, mean, rstd, ctx.slope, ctx.needs_input_grad[2])
        gx = gw = None
        with torch.autocast("cuda", enabled=False):
            g2 = gh.reshape(-1, gh.shape[-1])
            if ctx.needs_input_grad[0]:
                gx = (g2 @ wc).view(xc.shape).to(xdt)
            if ctx.needs_input_grad[1]:
                gw = dense.weight_grad(g2, xc.reshape(-1, xc.shape[-1])).to(wdt)
        return gx, gw, (None if xsum is None else xsum.to(cdt)), dg.to(gdt), db.to(bdt), None, None


class _FirstBlock(torch.autograd.Function):
    """Linear(1 -> D) -> LayerNorm -> LeakyReLU, the first block of the f0 / loudness stacks (decoder.py:43-44), as one HIP pass
    each way (include/ddsp_hip.h: ddsp_outer_ln_lrelu_*): the [rows, D] pre-activation is rebuilt from the ONE input feature
    instead of being written and read back, and the backward sums the Linear's weight / bias gradients beside the LayerNorm's
    -- the input carries no gradient (the caller checks).  Output in the autocast dtype when autocast is on, fp32 otherwise."""

    _IO = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, eps, slope):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
        if dt not in _FirstBlock._IO:
            dt = torch.float32
        D = weight.shape[0]
        xs = x.detach().reshape(-1).contiguous().float()
        rows = xs.numel()
        w, c = weight.detach().reshape(-1).contiguous().float(), bias.detach().contiguous().float()
        g, b = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
        y = torch.empty(x.shape[:-1] + (D,), device=x.device, dtype=dt)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ddsp_outer_ln_lrelu_forward(xs.data_ptr(), w.data_ptr(), c.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(),
                                                              mean.data_ptr(), rstd.data_ptr(), rows, D, float(eps), float(slope),
                                                              _FirstBlock._IO[dt], torch.cuda.current_stream().cuda_stream),
                       "ddsp_outer_ln_lrelu_forward")
        ctx.save_for_backward(xs, w, c, y, g, mean, rstd)
        ctx.slope = float(slope)
        ctx.param_meta = (weight.shape, weight.dtype, bias.dtype, gamma.dtype, beta.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        xs, w, c, y, g, mean, rstd = ctx.saved_tensors
        D = w.numel()
        rows = xs.numel()
        gy = gy.contiguous().to(y.dtype)
        out = torch.empty((4, D), device=y.device, dtype=torch.float32)        # d w | d bias | d gamma | d beta
        L = _lib.lib()
        scratch = torch.empty(L.ddsp_outer_ln_lrelu_scratch_bytes(D), device=y.device, dtype=torch.uint8)
        with torch.cuda.device(y.device):
            _lib.check(L.ddsp_outer_ln_lrelu_backward(gy.data_ptr(), xs.data_ptr(), w.data_ptr(), c.data_ptr(), y.data_ptr(), g.data_ptr(),
                                                      mean.data_ptr(), rstd.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                                      out[2].data_ptr(), out[3].data_ptr(), scratch.data_ptr(), rows, D, ctx.slope,
                                                      _FirstBlock._IO[y.dtype], torch.cuda.current_stream().cuda_stream),
                       "ddsp_outer_ln_lrelu_backward")
        wshape, wdt, cdt, gdt, bdt = ctx.param_meta
        return None, out[0].view(wshape).to(wdt), out[1].to(cdt), out[2].to(gdt), out[3].to(bdt), None, None


# Under torch.autocast the Linear layers hand over bf16 / fp16 activations: the fused LayerNorm pass reads and writes
# them as they are; the head non-linearity converts (`custom_fwd(cast_inputs=float32)`) and returns fp32.
_FUSED_DTYPES = (torch.float32, torch.bfloat16, torch.float16)


def _run_stack(stack: nn.Module, x: torch.Tensor) -> torch.Tensor:
    for i in range(stack.depth):
        linear, norm, act = getattr(stack, f"mlp_layer{i + 1}")
        if (linear.in_features == 1 and linear.bias is not None and x.is_cuda and not x.requires_grad and x.shape[-1] == 1
                and linear.out_features in (256, 512) and norm.elementwise_affine and norm.bias is not None and act.negative_slope > 0
                and linear.weight.dtype == torch.float32):
            # decoder.py:43-44: the f0 / loudness stacks start from ONE feature -- the whole block is one pass each way
            x = _FirstBlock.apply(x, linear.weight, linear.bias, norm.weight, norm.bias, norm.eps, act.negative_slope)
            continue
        if linear.in_features == 1 and linear.bias is not None and x.is_cuda:
            # decoder.py:43-44: the f0 / loudness stacks start from ONE feature -- an outer product, not a GEMM (as a
            # library GEMM with K = 1 it costs 0.16 ms in fp32 and 11 ms of host time per call in bf16 on this stack):
            # x * w^T + b as one fp32 elementwise pass, also under autocast
            x = torch.addcmul(linear.bias.float(), x.float(), linear.weight.float().view(-1))
        else:
            D = linear.out_features
            if (x.is_cuda and x.dtype in _FUSED_DTYPES and torch.is_grad_enabled() and linear.bias is not None and linear.in_features >= 16
                    and D % 256 == 0 and D <= 1024 and norm.elementwise_affine and norm.bias is not None and act.negative_slope > 0
                    and (linear.weight.requires_grad or x.requires_grad)):
                x = _LinearBlock.apply(x, linear.weight, linear.bias, norm.weight, norm.bias, norm.eps, act.negative_slope)
                continue
            x = dense.linear(x, linear.weight, linear.bias)
        D = x.shape[-1]
        if (x.is_cuda and x.dtype in _FUSED_DTYPES and D % 256 == 0 and D <= 1024 and norm.elementwise_affine
                and norm.bias is not None and act.negative_slope > 0):
            x = _LayerNormLeakyReLU.apply(x, norm.weight, norm.bias, norm.eps, act.negative_slope)
        else:
            x = act(norm(x))
    return x


class _ScaledSigmoid(torch.autograd.Function):
    """One HIP pass forward, one backward (include/ddsp_hip.h: ddsp_scaled_sigmoid_*)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ddsp_scaled_sigmoid_forward(x.data_ptr(), y.data_ptr(), x.numel(),
                                                              torch.cuda.current_stream().cuda_stream), "ddsp_scaled_sigmoid_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ddsp_scaled_sigmoid_backward(x.data_ptr(), gy.data_ptr(), gx.data_ptr(), x.numel(),
                                                               torch.cuda.current_stream().cuda_stream), "ddsp_scaled_sigmoid_backward")
        return gx


def scaled_sigmoid(x: torch.Tensor) -> torch.Tensor:
    """decoder.py:110-116: 2*sigmoid(x)^ln(10) + 1e-7 (the value range of every synth control)."""
    if x.is_cuda and x.dtype in _FUSED_DTYPES:
        return _ScaledSigmoid.apply(x)
    return 2.0 * torch.sigmoid(x).pow(2.3026) + 1e-7


FUSED_HEADS = True   # (tools/microbench/heads_ab.py flips it for the before / after figure)


class _Heads(torch.autograd.Function):
    """The three control heads (decoder.py:96-100: dense_harmonic, dense_loudness, dense_filter, each followed by modified_sigmoid
    :110-116) as ONE GEMM on the concatenated weights + ONE epilogue pass each way (include/ddsp_hip.h: ddsp_heads_sigmoid_*)
    instead of three GEMMs (one of them 1 wide) and three sigmoid passes forward, and nine GEMMs / reductions backward.  The
    parameters stay three separate tensors (checkpoint compatibility): their gradients are row slices of one weight gradient."""

    _IO = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

    @staticmethod
    def forward(ctx, z, w0, b0, w1, b1, w2, b2):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else z.dtype
        if dt not in _Heads._IO:
            dt = torch.float32
        zc = z.to(dt).contiguous()
        W = torch.cat([dense._cast(w, dt) for w in (w0, w1, w2)], 0)
        b = torch.cat([dense._cast(v, dt) for v in (b0, b1, b2)], 0)
        with torch.autocast("cuda", enabled=False):
            h = F.linear(zc, W, b).contiguous()
        ns = (w0.shape[0], w1.shape[0], w2.shape[0])
        rows = h.numel() // sum(ns)
        outs = [torch.empty(z.shape[:-1] + (n,), device=z.device, dtype=torch.float32) for n in ns]
        with torch.cuda.device(z.device):
            _lib.check(_lib.lib().ddsp_heads_sigmoid_forward(h.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), rows,
                                                             ns[0], ns[1], ns[2], _Heads._IO[dt], torch.cuda.current_stream().cuda_stream),
                       "ddsp_heads_sigmoid_forward")
        ctx.save_for_backward(zc, W, h)
        ctx.ns, ctx.in_dtype, ctx.param_dtypes = ns, z.dtype, (w0.dtype, b0.dtype)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        zc, W, h = ctx.saved_tensors
        ns = ctx.ns
        rows = h.numel() // sum(ns)
        gs = [(torch.zeros(h.shape[:-1] + (n,), device=h.device) if g is None else g.contiguous().float()) for g, n in zip((g0, g1, g2), ns)]
        gh = torch.empty_like(h)
        with torch.cuda.device(h.device):
            _lib.check(_lib.lib().ddsp_heads_sigmoid_backward(h.data_ptr(), gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(), gh.data_ptr(), rows,
                                                              ns[0], ns[1], ns[2], _Heads._IO[h.dtype], torch.cuda.current_stream().cuda_stream),
                       "ddsp_heads_sigmoid_backward")
        wdt, bdt = ctx.param_dtypes
        gz = gw = gb = None
        with torch.autocast("cuda", enabled=False):
            g2d = gh.reshape(-1, gh.shape[-1])
            if ctx.needs_input_grad[0]:
                gz = (g2d @ W).view(zc.shape).to(ctx.in_dtype)
            if any(ctx.needs_input_grad[1::2]):
                gw = dense.weight_grad(g2d, zc.reshape(-1, zc.shape[-1])).to(wdt)
            if any(ctx.needs_input_grad[2::2]):
                gb = dense.colsum(g2d).to(bdt)
        o1, o2 = ns[0], ns[0] + ns[1]
        cut = lambda t: (None, None, None) if t is None else (t[:o1], t[o1:o2], t[o2:])   # noqa: E731
        (gw0, gw1, gw2), (gb0, gb1, gb2) = cut(gw), cut(gb)
        return gz, gw0, gb0, gw1, gb1, gw2, gb2


class Controller(nn.Module):
    """f0 / loudness features -> control dict {f0, c, a, H, hidden} (decoder.py:41-108)."""

    def __init__(self, conf):
        super().__init__()
        width, depth = conf.decoder_mlp_units, conf.decoder_mlp_layers
        self.mlp_f0 = _dense_stack(1, width, depth)
        self.mlp_loudness = _dense_stack(1, width, depth)
        self.gru = GRU(2 * width, conf.decoder_gru_units, conf.decoder_gru_layers, batch_first=True)
        self.mlp_gru = _dense_stack(conf.decoder_gru_units + 2 * width, width, depth)
        self.dense_harmonic = nn.Linear(width, conf.n_harmonics)
        self.dense_loudness = nn.Linear(width, 1)
        self.dense_filter = nn.Linear(width, conf.n_noise_filters)

    def forward(self, batch, hidden=None):
        z_pitch = _run_stack(self.mlp_f0, batch['normalized_cents'])
        z_loud = _run_stack(self.mlp_loudness, batch['loudness'])
        z = torch.cat((z_pitch, z_loud), dim=-1)
        z, state = self.gru(z, hidden) if hidden is not None else self.gru(z)
        if z.dtype != z_pitch.dtype:
            # torch.autocast: the recurrence hands back fp32, the stacks 16-bit activations.  One narrow cast of z here instead of
            # cat widening both stacks to fp32 and the next Linear narrowing all three again (the Linear computes in 16 bit anyway)
            z = z.to(z_pitch.dtype)
        z = _run_stack(self.mlp_gru, torch.cat((z, z_pitch, z_loud), dim=-1))
        if FUSED_HEADS and z.is_cuda and z.dtype in _Heads._IO:
            c, a, H = _Heads.apply(z, self.dense_harmonic.weight, self.dense_harmonic.bias, self.dense_loudness.weight,
                                   self.dense_loudness.bias, self.dense_filter.weight, self.dense_filter.bias)
        else:
            def head(layer):
                return scaled_sigmoid(dense.linear(z, layer.weight, layer.bias))
            c, a, H = head(self.dense_harmonic), head(self.dense_loudness), head(self.dense_filter)
        controls = dict(f0=batch['f0'], c=c, hidden=state, H=H, a=a)
        if hidden is not None:
            return controls, hidden    # the INPUT state, as the reference returns it
        return controls


class Decoder(nn.Module):
    """controller -> harmonics + noise -> reverb (decoder.py:119-147), all on the device."""

    def __init__(self, conf, noise_rng: str = 'host', seed: int = 0):
        super().__init__()
        self.controller = Controller(conf)
        self.harmonics = OscillatorBank(conf)
        self.noise = FilteredNoise(conf, rng=noise_rng, seed=seed)
        self.reverb = Reverb(conf)

    def synthesize(self, ctrl, live: bool = False):
        """`harmonics + noise` for a control dict (decoder.py:129-132 / :141-144)."""
        dry = self.harmonics.live(ctrl) if live else self.harmonics(ctrl)
        if torch.is_grad_enabled() and (dry.requires_grad or ctrl['H'].requires_grad):
            return dry + self.noise(ctrl)
        # inference: the noise kernel accumulates straight into the oscillator's buffer (no extra pass over the audio)
        return self.noise(ctrl, out=dry)

    def forward(self, z):
        return self.reverb(self.synthesize(self.controller(z)))

    def forward_live(self, z, hidden):
        ctrl, hidden = self.controller(z, hidden)
        audio = self.reverb.live_forward(self.synthesize(ctrl, live=True))
        return audio.cpu().squeeze(0).numpy(), hidden   # one D2H per callback (rt/synth.py:50-52)
