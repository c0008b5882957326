#!/usr/bin/env python3
"""Training step (BASELINE.json configs[4] per-GPU shape, bf16 and fp16 autocast) with the controller's three heads as separate
Linear + modified_sigmoid passes (rounds 1-3) against the fused form (decoder._Heads: one GEMM on the concatenated weights + one
epilogue each way), interleaved rounds in one process on one box.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import decoder as dec  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402


class TrainConf:
    n_harmonics, n_noise_filters, sample_rate, hop_length = 100, 65, 16000, 128
    decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1


def setup(amp):
    torch.manual_seed(0)
    model = ddsp.Decoder(TrainConf, noise_rng="device", seed=0).cuda()
    loss_fn = ddsp.MSSLoss().cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    rng = np.random.default_rng(2000)
    b, frames = 32, 500
    batch = {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (b, frames, 1)).astype(np.float32)).cuda(),
             "loudness": torch.from_numpy(rng.uniform(-1, 1, (b, frames, 1)).astype(np.float32)).cuda(),
             "f0": torch.from_numpy(syn.musical_f0(rng, b, frames)).cuda(),
             "audio": torch.from_numpy((0.1 * rng.standard_normal((b, frames * 128))).astype(np.float32)).cuda()}
    scaler = torch.amp.GradScaler("cuda") if amp == "fp16" else None
    return model, loss_fn, opt, batch, {"bf16": torch.bfloat16, "fp16": torch.float16}[amp], scaler


def timed(state, steps=30):
    model, loss_fn, opt, batch, dt, scaler = state
    for _ in range(5):
        ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=dt, scaler=scaler)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=dt, scaler=scaler)
    e1.record()
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / steps, 4), round(1e3 * issue / steps, 4)


if __name__ == "__main__":
    out = {}
    for amp in ("bf16", "fp16"):
        state = setup(amp)
        res = {"separate": [], "fused": []}
        for r in range(4):
            for label, flag in (("separate", False), ("fused", True)):
                dec.FUSED_HEADS = flag
                res[label].append(timed(state))
        dec.FUSED_HEADS = True
        out[amp] = {k: {"gpu_ms_per_step": [a for a, _ in v], "host_issue_ms_per_step": [b for _, b in v]} for k, v in res.items()}
    print(json.dumps(out))
