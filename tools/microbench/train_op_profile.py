#!/usr/bin/env python3
"""Which torch ops launch the small kernels of the bf16 training step (fills, copies, casts, reductions)?  torch.profiler over a few
steps of bench.py's training configuration; prints device time per op, and for the suspicious ops the Python call sites."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


class TrainConf:
    n_harmonics, n_noise_filters, sample_rate, hop_length = 100, 65, 16000, 128
    decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1


b, frames = 32, 500
torch.manual_seed(0)
model = ddsp.Decoder(TrainConf, noise_rng="device", seed=0).cuda()
loss_fn = ddsp.MSSLoss().cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
rng = np.random.default_rng(2000)
batch = {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (b, frames, 1)).astype(np.float32)).cuda(),
         "loudness": torch.from_numpy(rng.uniform(-1, 1, (b, frames, 1)).astype(np.float32)).cuda(),
         "f0": torch.from_numpy(syn.musical_f0(rng, b, frames)).cuda(),
         "audio": torch.from_numpy((0.1 * rng.standard_normal((b, frames * 128))).astype(np.float32)).cuda()}
for _ in range(3):
    ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=torch.bfloat16)
torch.cuda.synchronize()
steps = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(steps):
        ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
print("=" * 40, "call sites of the small ops")
import collections
sites = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::copy_", "aten::sum", "aten::cat", "aten::mul", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::div", "aten::clone"):
        dev = getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0)
        stack = [fr for fr in (e.stack or []) if "ddsp" in fr or "bench" in fr or "train_op_profile" in fr][:3]
        key = (e.name, str(e.input_shapes)[:70], " <- ".join(fr.split("/")[-1][:60] for fr in stack))
        sites[key][0] += 1
        sites[key][1] += dev
for key, (n, t) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t / steps:8.1f} us/step {n / steps:5.1f} calls/step  {key[0]:12s} {key[1]:70s} {key[2]}")
