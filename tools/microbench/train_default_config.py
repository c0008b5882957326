#!/usr/bin/env python3
"""Training step at the REFERENCE'S DEFAULT configuration (config/default.py:8-24: 44.1 kHz, hop 512, 180 harmonics, 195 noise bands,
batch 16, 2 s clips = 172 frames; decoder + HIP synth + MSS loss + fused Adam on this one GPU), with the 195-band noise path as the
whole-batch matrix product (default) and as round 3 had it (cosine sums forward, direct kernels backward: ddsp_noise_set_generic(16)),
same process, interleaved.  Also the noise module's forward + backward alone at that shape.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("DDSP_TEST_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


class Conf:
    n_harmonics, n_noise_filters, sample_rate, hop_length = 180, 195, 44100, 512
    decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1


B, T = 16, 172
L = ddsp._lib.lib()
torch.manual_seed(0)
model = ddsp.Decoder(Conf, noise_rng="device", seed=0).cuda()
loss_fn = ddsp.MSSLoss().cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
rng = np.random.default_rng(7)
batch = {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (B, T, 1)).astype(np.float32)).cuda(),
         "loudness": torch.from_numpy(rng.uniform(-1, 1, (B, T, 1)).astype(np.float32)).cuda(),
         "f0": torch.from_numpy(syn.musical_f0(rng, B, T)).cuda(),
         "audio": torch.from_numpy((0.1 * rng.standard_normal((B, T * 512))).astype(np.float32)).cuda()}


def train_ms(steps=30, warmup=10):
    for _ in range(warmup):
        ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    return round(1e3 * (time.perf_counter() - t0) / steps, 4)


Hn = torch.from_numpy(syn.controller_range(rng.standard_normal((B, T, 195), dtype=np.float32))).cuda()
gy = torch.randn(B, T * 512, device="cuda")


def noise_ms(reps=50):
    for _ in range(5):
        ddsp.noise_backward(gy, 512, 195, seed=1)
        ddsp.noise_forward(Hn, 512, seed=1)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    e[0].record()
    for i in range(reps):
        ddsp.noise_forward(Hn, 512, seed=1, offset=i << 32)
    e[1].record()
    for i in range(reps):
        ddsp.noise_backward(gy, 512, 195, seed=1, offset=i << 32)
    e[2].record()
    torch.cuda.synchronize()
    return round(e[0].elapsed_time(e[1]) / reps, 4), round(e[1].elapsed_time(e[2]) / reps, 4)


out = {"train_step_ms": {"product": [], "round3_paths": []}, "noise_fwd_bwd_ms": {"product": [], "round3_paths": []}}
train_ms(5, 5)                      # one-time costs
for rnd in range(3):
    for label, mode in (("product", 0), ("round3_paths", 16)):
        assert L.ddsp_noise_set_generic(mode) == 0
        out["train_step_ms"][label].append(train_ms())
        out["noise_fwd_bwd_ms"][label].append(noise_ms())
L.ddsp_noise_set_generic(0)
print(json.dumps(out))
