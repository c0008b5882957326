#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, read-only); the GPU
box never sees the reference.  The fixtures are data only: inputs + the
outputs the reference's own modules produced on the CPU (torch CPU semantics are
the oracle BASELINE.json fixes).  Plan: SURVEY.md Appendix B (G1..G11).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py

Reference entry points exercised (file:line in /root/reference):
  model/ddsp/harmonic_oscillator.py:24-37,39-43,45-50,57-62,64-75
  model/ddsp/filtered_noise.py:7-22,25-32,40-53
  model/ddsp/reverb.py:24-49
  model/autoencoder/decoder.py:129-133 (wiring, restated as three calls)
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from model.ddsp.harmonic_oscillator import OscillatorBank as RefOsc  # noqa: E402
from model.ddsp.filtered_noise import (FilteredNoise as RefNoise,  # noqa: E402
                                       amp_to_impulse_response as ref_ir,
                                       fft_convolve as ref_fft_convolve)
from model.ddsp.reverb import Reverb as RefReverb  # noqa: E402

from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)


class Conf:
    def __init__(self, n_harmonics, sample_rate, hop_length):
        self.n_harmonics = n_harmonics
        self.sample_rate = sample_rate
        self.hop_length = hop_length


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrays):
    meta = {k: v for k, v in arrays.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **meta)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def osc_with_intermediates(conf, f0, c, a):
    """Run the reference stage by stage (same calls as forward(), lines 57-62)."""
    osc = RefOsc(conf)
    x = {"f0": t(f0), "c": t(c), "a": t(a)}
    inc, amps = osc.prepare_harmonics(x["f0"], x["c"])
    inc = inc.contiguous().clone()
    pre = torch.cumsum(inc, dim=1)                       # :41, before the in-place %=
    phases = osc.generate_phases(inc.clone())
    loud_up = osc.rescale(x["a"])
    amps_up = osc.rescale(amps)
    y = osc.generate_signal(amps, x["a"], phases)
    y_fwd = RefOsc(conf)(x)
    assert torch.equal(torch.nan_to_num(y, nan=1234.5), torch.nan_to_num(y_fwd, nan=1234.5))
    return dict(inc=inc.numpy(), cum=pre.numpy(), phi=phases.numpy(), amp_frame=amps.numpy(),
                amp_up=amps_up.contiguous().numpy(), loud_up=loud_up.contiguous().numpy(), y=y.numpy())


def osc_forward(conf, f0, c, a):
    return RefOsc(conf)({"f0": t(f0), "c": t(c), "a": t(a)}).numpy()


def osc_phases(conf, f0, c):
    osc = RefOsc(conf)
    inc, _ = osc.prepare_harmonics(t(f0), t(c))
    return osc.generate_phases(inc).numpy()


def g1():
    conf = Conf(8, 16000, 64)
    rng = np.random.default_rng(101)
    f0 = rng.uniform(60, 900, (2, 16, 1)).astype(np.float32)
    c = syn.controller_range(rng.standard_normal((2, 16, 8), dtype=np.float32))
    a = syn.controller_range(rng.standard_normal((2, 16, 1), dtype=np.float32))
    r = osc_with_intermediates(conf, f0, c, a)
    save("g1_osc_tiny", sample_rate=16000, hop=64, f0=f0, c=c, a=a, **r)


def g2_g3_g4():
    for name, shape, kind, seed in (("g2_osc_cfg2_live", syn.CFG2, "all_live", 1002),
                                    ("g3_osc_cfg2_musical", syn.CFG2, "musical", 1003)):
        ctl = syn.make_controls(shape, seed, kind, batch=1)
        conf = Conf(shape.n_harmonics, shape.sample_rate, shape.hop)
        y = osc_forward(conf, ctl["f0"], ctl["c"], ctl["a"])
        phi = osc_phases(conf, ctl["f0"], ctl["c"])
        idx = np.arange(0, shape.samples, 997)
        save(name, sample_rate=shape.sample_rate, hop=shape.hop, f0=ctl["f0"], c=ctl["c"], a=ctl["a"],
             y=y, phi_idx=idx, phi_sub=phi[:, idx, :])
    shape = syn.SynthShape("cfg3_1s", 1, 48000, 512, 94, 200, 257)
    ctl = syn.make_controls(shape, 1004, "all_live")
    conf = Conf(200, 48000, 512)
    y = osc_forward(conf, ctl["f0"], ctl["c"], ctl["a"])
    phi = osc_phases(conf, ctl["f0"], ctl["c"])
    idx = np.arange(0, shape.samples, 499)
    save("g4_osc_cfg3_1s", sample_rate=48000, hop=512, f0=ctl["f0"], c=ctl["c"], a=ctl["a"],
         y=y, phi_idx=idx, phi_sub=phi[:, idx, :])
    # cfg1 (the reference's own CPU-runnable case), whole clip
    ctl = syn.make_controls(syn.CFG1, 1001, "all_live")
    conf = Conf(60, 16000, 128)
    y = osc_forward(conf, ctl["f0"], ctl["c"], ctl["a"])
    save("g2b_osc_cfg1", sample_rate=16000, hop=128, f0=ctl["f0"], c=ctl["c"], a=ctl["a"], y=y)


def g5():
    # Nyquist edges: sr//2 = 8000.  k*f0 == 8000 exactly (kept: strict >), just above (masked),
    # all-masked frame (0/0 -> NaN), f0 == 0.
    conf = Conf(4, 16000, 64)
    f0 = np.array([[2000.0], [2000.0001220703125], [9000.0], [0.0], [4000.0], [1999.9998779296875]],
                  dtype=np.float32)[None]
    rng = np.random.default_rng(105)
    c = rng.uniform(0.1, 1.0, (1, 6, 4)).astype(np.float32)
    a = rng.uniform(0.1, 1.0, (1, 6, 1)).astype(np.float32)
    r = osc_with_intermediates(conf, f0, c, a)
    save("g5_osc_nyquist", sample_rate=16000, hop=64, f0=f0, c=c, a=a, **r)
    # same without the all-masked frame, so that finite outputs are compared too
    f0b = f0.copy()
    f0b[0, 2, 0] = 3000.0
    r = osc_with_intermediates(conf, f0b, c, a)
    save("g5b_osc_nyquist_finite", sample_rate=16000, hop=64, f0=f0b, c=c, a=a, **r)


def g6():
    for hop in (100, 441, 3, 7, 160, 480):
        conf = Conf(8, 44100, hop)
        rng = np.random.default_rng(600 + hop)
        f0 = rng.uniform(40, 2500, (2, 12, 1)).astype(np.float32)
        c = syn.controller_range(rng.standard_normal((2, 12, 8), dtype=np.float32))
        a = syn.controller_range(rng.standard_normal((2, 12, 1), dtype=np.float32))
        r = osc_with_intermediates(conf, f0, c, a)
        save(f"g6_osc_hop{hop}", sample_rate=44100, hop=hop, f0=f0, c=c, a=a,
             inc=r["inc"], cum=r["cum"], phi=r["phi"], y=r["y"])
    # single-frame clip (T=1): every sample clamps to frame 0
    conf = Conf(8, 16000, 64)
    rng = np.random.default_rng(699)
    f0 = rng.uniform(40, 900, (3, 1, 1)).astype(np.float32)
    c = syn.controller_range(rng.standard_normal((3, 1, 8), dtype=np.float32))
    a = syn.controller_range(rng.standard_normal((3, 1, 1), dtype=np.float32))
    r = osc_with_intermediates(conf, f0, c, a)
    save("g6_osc_single_frame", sample_rate=16000, hop=64, f0=f0, c=c, a=a,
         inc=r["inc"], cum=r["cum"], phi=r["phi"], y=r["y"])


def g7():
    # live x 3 (harmonic_oscillator.py:64-75), reference default rt shape: sr 44.1k, R 512, H 180, B 1, T 4
    conf = Conf(180, 44100, 512)
    osc = RefOsc(conf)
    rng = np.random.default_rng(107)
    arrays = {}
    for call in range(3):
        f0 = rng.uniform(80, 600, (1, 4, 1)).astype(np.float32)
        c = syn.controller_range(rng.standard_normal((1, 4, 180), dtype=np.float32))
        a = syn.controller_range(rng.standard_normal((1, 4, 1), dtype=np.float32))
        y = osc.live({"f0": t(f0), "c": t(c), "a": t(a)})
        arrays[f"f0_{call}"] = f0
        arrays[f"c_{call}"] = c
        arrays[f"a_{call}"] = a
        arrays[f"y_{call}"] = y.numpy()
        arrays[f"last_phases_{call}"] = osc.last_phases.detach().numpy().copy()
    save("g7_osc_live", sample_rate=44100, hop=512, **arrays)
    # live with batch 2: only row 0 carries state (Appendix C.6)
    conf = Conf(16, 16000, 64)
    osc = RefOsc(conf)
    arrays = {}
    for call in range(2):
        f0 = rng.uniform(80, 400, (2, 3, 1)).astype(np.float32)
        c = syn.controller_range(rng.standard_normal((2, 3, 16), dtype=np.float32))
        a = syn.controller_range(rng.standard_normal((2, 3, 1), dtype=np.float32))
        y = osc.live({"f0": t(f0), "c": t(c), "a": t(a)})
        arrays[f"f0_{call}"] = f0
        arrays[f"c_{call}"] = c
        arrays[f"a_{call}"] = a
        arrays[f"y_{call}"] = y.numpy()
        arrays[f"last_phases_{call}"] = osc.last_phases.detach().numpy().copy()
    save("g7b_osc_live_batch2", sample_rate=16000, hop=64, **arrays)


def g8():
    for hop, nf in ((128, 65), (512, 257), (512, 195), (64, 65), (100, 33)):
        conf = Conf(1, 16000, hop)
        rng = np.random.default_rng(800 + hop + nf)
        H = syn.controller_range(rng.standard_normal((2, 8, nf), dtype=np.float32))
        seed = 4242 + hop
        torch.manual_seed(seed)
        y = RefNoise(conf)({"H": t(H)}).numpy()
        torch.manual_seed(seed)
        u = torch.rand(2, 8, hop).numpy()           # the same draw the module made (filtered_noise.py:44-48)
        ir = ref_ir(t(H), hop).numpy()
        save(f"g8_noise_hop{hop}_f{nf}", hop=hop, H=H, seed=seed, uniform=u, ir=ir, y=y)


def g9():
    rng = np.random.default_rng(109)
    sig = rng.standard_normal((2, 4096)).astype(np.float32)
    ker = rng.standard_normal((2, 4096)).astype(np.float32) * np.exp(-np.arange(4096) / 300.0).astype(np.float32)
    save("g9_fft_convolve", signal=sig, kernel=ker, y=ref_fft_convolve(t(sig), t(ker)).numpy())
    for clip in (4096, 1024):
        conf = Conf(1, 2048, 64)
        torch.manual_seed(9)
        rv = RefReverb(conf, initial_wet=0.5, initial_decay=4.0)
        x = rng.standard_normal((2, clip)).astype(np.float32)
        y = rv(t(x)).numpy()
        save(f"g9_reverb_clip{clip}", sample_rate=2048, x=x, noise=rv.noise.numpy(), decay=rv.decay.numpy(),
             wet=rv.wet.numpy(), impulse=rv.build_impulse().numpy(), y=y)
    conf = Conf(1, 2048, 64)
    torch.manual_seed(10)
    rv = RefReverb(conf, initial_wet=0.3, initial_decay=3.0)
    arrays = dict(noise=rv.noise.numpy(), decay=rv.decay.numpy(), wet=rv.wet.numpy())
    for call in range(3):
        x = rng.standard_normal((1, 256)).astype(np.float32)
        arrays[f"x_{call}"] = x
        arrays[f"y_{call}"] = rv.live_forward(t(x)).numpy()
        arrays[f"buffer_{call}"] = rv.buffer.detach().numpy().copy()
    save("g9_reverb_live", sample_rate=2048, **arrays)


def g10():
    torch.set_grad_enabled(True)
    conf = Conf(8, 16000, 64)
    rng = np.random.default_rng(110)
    f0 = rng.uniform(60, 1500, (2, 16, 1)).astype(np.float32)
    c = syn.controller_range(rng.standard_normal((2, 16, 8), dtype=np.float32))
    a = syn.controller_range(rng.standard_normal((2, 16, 1), dtype=np.float32))
    g = rng.standard_normal((2, 16 * 64)).astype(np.float32)
    ct, at = t(c).requires_grad_(), t(a).requires_grad_()
    y = RefOsc(conf)({"f0": t(f0), "c": ct, "a": at})
    (y * t(g)).sum().backward()
    save("g10_osc_grad", sample_rate=16000, hop=64, f0=f0, c=c, a=a, g=g, y=y.detach().numpy(),
         grad_c=ct.grad.numpy(), grad_a=at.grad.numpy())
    for hop, nf in ((128, 65), (64, 65)):
        H = syn.controller_range(rng.standard_normal((2, 8, nf), dtype=np.float32))
        g = rng.standard_normal((2, 8 * hop)).astype(np.float32)
        Ht = t(H).requires_grad_()
        torch.manual_seed(77)
        y = RefNoise(Conf(1, 16000, hop))({"H": Ht})
        torch.manual_seed(77)
        u = torch.rand(2, 8, hop).numpy()
        (y * t(g)).sum().backward()
        save(f"g10_noise_grad_hop{hop}", hop=hop, H=H, g=g, uniform=u, y=y.detach().numpy(), grad_H=Ht.grad.numpy())
    torch.set_grad_enabled(False)


def g11():
    # decoder.py:129-133 wiring from a fixed ctrl dict: harmonics + noise -> reverb
    shape = syn.SynthShape("g11", 1, 16000, 128, 32, 100, 65)
    ctl = syn.make_controls(shape, 1011, "musical")
    conf = Conf(100, 16000, 128)
    x = {k: t(v) for k, v in ctl.items()}
    torch.manual_seed(11)
    rv = RefReverb(conf, initial_wet=-1.0, initial_decay=4.0)
    torch.manual_seed(12)
    harm = RefOsc(conf)(x)
    noise = RefNoise(conf)(x)
    torch.manual_seed(12)
    u = torch.rand(1, 32, 128).numpy()
    y = rv(harm + noise)
    save("g11_decoder_wiring", sample_rate=16000, hop=128, **ctl, uniform=u, rv_noise=rv.noise.numpy(),
         rv_decay=rv.decay.numpy(), rv_wet=rv.wet.numpy(), harm=harm.numpy(), noise=noise.numpy(), y=y.numpy())


def g12():
    # controller (decoder.py:41-116), tiny widths: weights + inputs + outputs, incl. the hidden-state pass-through (App. C.7)
    from model.autoencoder.decoder import Controller as RefController

    class C:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 8, 9, 16000, 64
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 16, 2, 12, 1

    torch.manual_seed(1212)
    ctl = RefController(C)
    rng = np.random.default_rng(112)
    batch = {"normalized_cents": rng.uniform(0, 1, (2, 6, 1)).astype(np.float32),
             "loudness": rng.uniform(-1, 1, (2, 6, 1)).astype(np.float32),
             "f0": rng.uniform(80, 400, (2, 6, 1)).astype(np.float32)}
    tb = {k: t(v) for k, v in batch.items()}
    out = ctl(tb)
    h0 = torch.from_numpy(rng.standard_normal((1, 2, 12)).astype(np.float32))
    out2, h_ret = ctl(tb, h0)
    arrays = {f"w__{k}": v.numpy() for k, v in ctl.state_dict().items()}
    save("g12_controller", **batch, **arrays, c=out["c"].numpy(), a=out["a"].numpy(), H=out["H"].numpy(),
         hidden=out["hidden"].numpy(), h0=h0.numpy(), c2=out2["c"].numpy(), hidden2=out2["hidden"].numpy(), h_ret=h_ret.numpy())


def g13():
    # end to end: the reference Decoder.forward (decoder.py:127-135) on the CPU with fixed weights -- controller ->
    # harmonics + noise -> reverb.  The noise draw comes from the global CPU generator (seeded right before the call).
    from model.autoencoder.decoder import Decoder as RefDecoder

    class C:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 16, 9, 4000, 64
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 16, 2, 12, 1

    torch.manual_seed(1313)
    dec = RefDecoder(C)
    with torch.no_grad():
        dec.reverb.wet.fill_(0.5)
        dec.reverb.decay.fill_(3.0)
    rng = np.random.default_rng(113)
    batch = {"normalized_cents": rng.uniform(0, 1, (2, 70, 1)).astype(np.float32),
             "loudness": rng.uniform(-1, 1, (2, 70, 1)).astype(np.float32),
             "f0": rng.uniform(60, 300, (2, 70, 1)).astype(np.float32)}
    tb = {k: t(v) for k, v in batch.items()}
    torch.manual_seed(77)
    y = dec(tb)
    torch.manual_seed(78)
    y_short = dec({k: v[:, :20] for k, v in tb.items()})       # clip shorter than one second: the reverb crops (reverb.py:34)
    arrays = {f"w__{k}": v.numpy() for k, v in dec.state_dict().items()}
    save("g13_decoder_end_to_end", **batch, **arrays, y=y.numpy(), y_short=y_short.numpy())


def g14():
    # the real-time callback: the reference Decoder.forward_live (decoder.py:139-147) called three times on the CPU with
    # fixed weights and a carried GRU input state -- oscillator phases (harmonic_oscillator.py:70-72) and reverb history
    # (reverb.py:41-49) persist between the calls; each call's noise draw is seeded right before it.
    from model.autoencoder.decoder import Decoder as RefDecoder

    class C:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 16, 9, 4000, 64
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 16, 2, 12, 1

    torch.manual_seed(1414)
    dec = RefDecoder(C)
    with torch.no_grad():
        dec.reverb.wet.fill_(0.5)
        dec.reverb.decay.fill_(3.0)
    arrays = {f"w__{k}": v.numpy().copy() for k, v in dec.state_dict().items()}   # before the calls mutate the state
    rng = np.random.default_rng(114)
    hidden = torch.from_numpy(rng.standard_normal((1, 1, 12)).astype(np.float32))
    out = {}
    for call in range(3):
        z = {"normalized_cents": rng.uniform(0, 1, (1, 8, 1)).astype(np.float32),
             "loudness": rng.uniform(-1, 1, (1, 8, 1)).astype(np.float32),
             "f0": rng.uniform(60, 300, (1, 8, 1)).astype(np.float32)}
        torch.manual_seed(140 + call)
        audio, h_ret = dec.forward_live({k: t(v) for k, v in z.items()}, hidden)
        assert h_ret is hidden                                       # App. C.7: the INPUT state comes back
        out.update({f"{k}_{call}": v for k, v in z.items()})
        out[f"audio_{call}"] = np.asarray(audio, dtype=np.float32)
    save("g14_decoder_live_callbacks", **arrays, **out, hidden=hidden.numpy(),
         last_phases=dec.harmonics.last_phases.detach().numpy().astype(np.float32))


def g15():
    # gradients end to end: the reference Decoder.forward (decoder.py:127-135) on the CPU, loss = sum(audio * weight), and the
    # reference's own autograd for EVERY trainable parameter (controller MLPs / GRU / heads, reverb) -- what train/train.py:32-37
    # differentiates, without the (torchaudio-based) spectral loss.
    from model.autoencoder.decoder import Decoder as RefDecoder

    class C:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 16, 9, 4000, 64
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 16, 2, 12, 1

    torch.manual_seed(1515)
    dec = RefDecoder(C)
    with torch.no_grad():
        dec.reverb.wet.fill_(0.5)
        dec.reverb.decay.fill_(3.0)
    arrays = {f"w__{k}": v.numpy().copy() for k, v in dec.state_dict().items()}
    rng = np.random.default_rng(115)
    batch = {"normalized_cents": rng.uniform(0, 1, (2, 70, 1)).astype(np.float32),
             "loudness": rng.uniform(-1, 1, (2, 70, 1)).astype(np.float32),
             "f0": rng.uniform(60, 300, (2, 70, 1)).astype(np.float32)}
    weight = rng.standard_normal((2, 70 * 64)).astype(np.float32)
    with torch.enable_grad():
        torch.manual_seed(150)
        y = dec({k: t(v) for k, v in batch.items()})
        (y * t(weight)).sum().backward()
    grads = {f"g__{k}": p.grad.numpy().copy() for k, p in dec.named_parameters() if p.requires_grad and p.grad is not None}
    save("g15_decoder_gradients", **batch, **arrays, **grads, weight=weight, y=y.detach().numpy())



def g16():
    """Round 2: the reverb at realistic lengths and its autograd, straight from the reference (reverb.py:24-49).
    * live: three `live_forward` callbacks of 512 samples against a 16 000-sample history (the HIP path computes the last n
      outputs directly instead of the 2L-point FFT convolution);
    * grad: d(sum(y * w)) / d(x, noise, decay, wet) of `forward` for a padded (clip > sample_rate) and a cropped clip."""
    rng = np.random.default_rng(116)
    conf = Conf(1, 16000, 64)
    torch.manual_seed(16)
    rv = RefReverb(conf, initial_wet=0.4, initial_decay=3.0)
    arrays = dict(noise=rv.noise.numpy(), decay=rv.decay.numpy(), wet=rv.wet.numpy())
    for call in range(3):
        x = rng.standard_normal((1, 512)).astype(np.float32)
        arrays[f"x_{call}"] = x
        arrays[f"y_{call}"] = rv.live_forward(t(x)).numpy()
    arrays["buffer_last"] = rv.buffer.detach().numpy().copy()
    save("g16_reverb_live_16k", sample_rate=16000, **arrays)
    torch.set_grad_enabled(True)
    for clip in (3000, 1200):
        conf = Conf(1, 2048, 64)
        torch.manual_seed(17)
        rv = RefReverb(conf, initial_wet=0.2, initial_decay=2.5)
        x = torch.from_numpy(rng.standard_normal((2, clip)).astype(np.float32)).requires_grad_()
        w = rng.standard_normal((2, clip)).astype(np.float32)
        y = rv(x)
        (y * t(w)).sum().backward()
        save(f"g16_reverb_grad_clip{clip}", sample_rate=2048, x=x.detach().numpy(), w=w, noise=rv.noise.detach().numpy(),
             decay=rv.decay.detach().numpy(), wet=rv.wet.detach().numpy(), y=y.detach().numpy(), grad_x=x.grad.numpy(),
             grad_noise=rv.noise.grad.numpy(), grad_decay=rv.decay.grad.numpy(), grad_wet=rv.wet.grad.numpy())
    torch.set_grad_enabled(False)

def g17():
    """The reference's autograd d/dH of FilteredNoise at hop 512 (filtered_noise.py:25-32 through :7-22): 257 bands (the impulse
    response fills the frame: the packed-FFT route of the HIP backward) and 195 bands (the reference's default, config/default.py:
    S = 388 < hop, cosine-sum route); 3 x 5 frames so that the last frame pair is half empty."""
    torch.set_grad_enabled(True)
    rng = np.random.default_rng(170)
    for nf in (257, 195):
        hop = 512
        H = syn.controller_range(rng.standard_normal((3, 5, nf), dtype=np.float32))
        g = rng.standard_normal((3, 5 * hop)).astype(np.float32)
        Ht = t(H).requires_grad_()
        torch.manual_seed(1700 + nf)
        y = RefNoise(Conf(1, 48000, hop))({"H": Ht})
        torch.manual_seed(1700 + nf)
        u = torch.rand(3, 5, hop).numpy()
        (y * t(g)).sum().backward()
        save(f"g17_noise_grad_hop512_f{nf}", hop=hop, H=H, g=g, uniform=u, y=y.detach().numpy(), grad_H=Ht.grad.numpy())
    torch.set_grad_enabled(False)


if __name__ == "__main__":
    print("torch", torch.__version__, "threads", torch.get_num_threads())
    only = sys.argv[1:]                                       # e.g. `make_goldens.py g16`: just the named generators
    for fn in (g1, g2_g3_g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14, g15, g16, g17):
        if not only or fn.__name__ in only:
            fn()
