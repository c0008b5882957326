#!/usr/bin/env python3
"""Times the hot path on every single-GPU configuration of BASELINE.json (cfg1..cfg3 + the cfg4 shard) and the live
callbacks.  Prints one JSON object per config (not the bench.py contract: bench.py stays on the metric's configuration).
Parity of these shapes is the tests' business (tests/test_gpu_full_size.py: whole rows against the oracle)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


def run(shape, seed, steps=5):
    ctl = syn.make_controls(shape, seed, "all_live")
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}

    def step(i):
        y, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
        ddsp.noise_forward(x["H"], shape.hop, seed=7, offset=i << 32, out=y, accumulate=True)
        return y

    step(0)
    ddsp._lib.profile_enable(8 * steps + 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        y = step(i + 1)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    rec = {}
    for name, ms in ddsp._lib.profile_read():
        rec.setdefault(name, []).append(ms)
    ddsp._lib.profile_enable(0)
    assert bool(torch.isfinite(y).all())
    print(json.dumps({"config": shape.name, "batch": shape.batch, "sample_rate": shape.sample_rate, "hop": shape.hop,
                      "harmonics": shape.n_harmonics, "noise_bands": shape.n_noise_filters, "ms_per_step": 1e3 * el,
                      "samples_per_s": shape.batch * shape.samples / el,
                      "kernel_ms": {k: round(float(np.mean(v)), 4) for k, v in rec.items()}}), flush=True)


def run_live(calls=200):
    """The rt path (rt/synth.py:40-55): OscillatorBank.live + FilteredNoise per callback, B=1, 4 frames of 512 samples
    at 44.1 kHz, 180 harmonics, 195 noise bands (config/default.py:11-19).  Deadline: 2048/44100 s = 46.4 ms."""
    shape = syn.SynthShape("rt_live_default", 1, 44100, 512, 4, 180, 195)

    class Conf:
        n_harmonics, sample_rate, hop_length = 180, 44100, 512

    osc = ddsp.OscillatorBank(Conf).cuda()
    noise = ddsp.FilteredNoise(Conf, rng="device")
    ctl = syn.make_controls(shape, 5, "musical")
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
    with torch.no_grad():
        for _ in range(5):
            y = osc.live(x) + noise(x)
        torch.cuda.synchronize()
        lat = []
        for _ in range(calls):
            t0 = time.perf_counter()
            y = osc.live(x) + noise(x)
            out = y.cpu()                      # the callback needs the samples on the host
            lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    gs = ddsp.GraphedSynth(Conf, 1, 4, 195, live=True)
    for name in ("f0", "c", "a", "H"):
        getattr(gs, name).copy_(x[name])
    glat = []
    for _ in range(calls):
        t0 = time.perf_counter()
        out = gs.run().cpu()
        glat.append(time.perf_counter() - t0)
    glat = np.array(glat) * 1e3
    print(json.dumps({"config": shape.name, "samples_per_call": shape.samples, "latency_ms_median": float(np.median(lat)),
                      "latency_ms_p99": float(np.percentile(lat, 99)), "hipgraph_latency_ms_median": float(np.median(glat)),
                      "hipgraph_latency_ms_p99": float(np.percentile(glat, 99)), "deadline_ms": 1e3 * 2048 / 44100}), flush=True)


def run_live_decoder(calls=200):
    """The whole callback of rt/synth.py:40-55 = Decoder.forward_live (decoder.py:139-147): controller (MLPs + GRU with a
    carried state) -> harmonics.live + noise -> reverb.live_forward -> D2H, at config/default.py:8-24 (44.1 kHz, hop 512,
    180 harmonics, 195 bands, 512-wide MLPs/GRU), 4 frames = 2048 samples per callback.  The same decoder with the stock
    nn.GRU in place of the HIP recurrence is timed beside it."""
    import torch.nn as nn

    class Conf:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 180, 195, 44100, 512
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1

    torch.manual_seed(0)
    out = {"config": "rt_live_decoder_default", "samples_per_call": 2048, "deadline_ms": 1e3 * 2048 / 44100}
    rng = np.random.default_rng(3)
    z = {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (1, 4, 1)).astype(np.float32)).cuda(),
         "loudness": torch.from_numpy(rng.uniform(-1, 1, (1, 4, 1)).astype(np.float32)).cuda(),
         "f0": torch.from_numpy(rng.uniform(200, 400, (1, 4, 1)).astype(np.float32)).cuda()}
    for label in ("hip_gru", "stock_gru"):
        dec = ddsp.Decoder(Conf, noise_rng="device").cuda().eval()
        if label == "stock_gru":
            stock = nn.GRU(1024, 512, 1, batch_first=True).cuda()
            stock.load_state_dict(dec.controller.gru.state_dict())
            dec.controller.gru = stock
        hidden = torch.zeros(1, 1, 512, device="cuda")
        lat = []
        with torch.no_grad():
            for i in range(calls + 10):
                t0 = time.perf_counter()
                audio, _ = dec.forward_live(z, hidden)
                if i >= 10:
                    lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        out[label + "_latency_ms_median"] = float(np.median(lat))
        out[label + "_latency_ms_p99"] = float(np.percentile(lat, 99))
        assert audio.shape == (2048,) and np.isfinite(audio).all()
    dec = ddsp.Decoder(Conf, noise_rng="device").cuda().eval()
    live = ddsp.GraphedLiveDecoder(dec, frames=4)
    zn = {k: v.cpu().numpy() for k, v in z.items()}
    lat = []
    for i in range(calls + 10):
        t0 = time.perf_counter()
        audio = live.run(zn)
        if i >= 10:
            lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    out["hipgraph_latency_ms_median"] = float(np.median(lat))
    out["hipgraph_latency_ms_p99"] = float(np.percentile(lat, 99))
    assert audio.shape == (2048,) and np.isfinite(audio).all()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    run_live()
    run_live_decoder()
    cfg2 = syn.CFG2
    for shape, seed in ((syn.CFG1, 1001), (cfg2, 1002), (syn.CFG3, 1003), (syn.CFG4_PER_GPU, 1004)):
        run(shape, seed)
