#!/usr/bin/env python3
"""Chunked oscillator form (ddsp_osc_chunk.hip) against the frame kernels (ddsp_osc.hip) on one box:
parity of both against the CPU oracle on a sweep of shapes, then interleaved timing at the BASELINE shapes.

    python tools/microbench/osc_chunk_ab.py [parity] [time] [musical]
"""
import json
import os
import sys

import numpy as np
import torch

os.environ.setdefault("DDSP_TEST_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402

L = ddsp._lib.lib()


def set_path(v):
    rc = L.ddsp_osc_set_path(v)
    assert rc == 0, rc


def run(x, shape):
    y, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
    return y


def parity():
    cases = [
        # name, B, sr, hop, T, H, kind
        ("hop128_h100", 9, 16000, 128, 37, 100, "all_live"),
        ("hop128_h100_musical", 11, 16000, 128, 53, 100, "musical"),
        ("hop64_h60", 3, 16000, 64, 19, 60, "musical"),
        ("hop512_h200", 5, 48000, 512, 23, 200, "all_live"),
        ("hop512_h180_musical", 2, 44100, 512, 11, 180, "musical"),
        ("hop256_h50", 17, 16000, 256, 7, 50, "musical"),
        ("one_frame", 4, 16000, 128, 1, 100, "all_live"),
        ("two_frames", 1, 16000, 128, 2, 100, "musical"),
        ("long_b1", 1, 16000, 128, 500, 100, "all_live"),
        ("cfg2_rows", 64, 16000, 128, 500, 100, "musical"),
    ]
    worst = 0.0
    for name, B, sr, hop, T, H, kind in cases:
        shape = syn.SynthShape(name, B, sr, hop, T, H, 65)
        ctl = syn.make_controls(shape, 4242 + B, kind)
        x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
        set_path(1)
        y_frame_d = run(x, shape)
        poison = torch.full_like(y_frame_d, float("nan"))   # the next torch.empty of this size gets a block full of NaN
        torch.cuda.synchronize()
        del poison
        set_path(0)
        y_chunk = run(x, shape).cpu().numpy()
        y_frame = y_frame_d.cpu().numpy()
        rows = min(B, 3)
        ref = oracle.osc_forward(ctl["f0"][:rows], ctl["c"][:rows], ctl["a"][:rows], hop, sr)
        e_frame = float(np.max(np.abs(y_frame[:rows] - ref)))
        e_chunk = float(np.max(np.abs(y_chunk[:rows] - ref)))
        e_ab = float(np.max(np.abs(y_chunk - y_frame)))
        worst = max(worst, e_chunk, e_ab)
        print(json.dumps({"case": name, "err_frame_vs_oracle": e_frame, "err_chunk_vs_oracle": e_chunk,
                          "chunk_vs_frame": e_ab, "finite": bool(np.isfinite(y_chunk).all())}), flush=True)
    print(json.dumps({"parity_worst": worst, "ok": worst <= 1e-5}), flush=True)
    return worst <= 1e-5


def timing(kind="all_live", rounds=4, steps=20):
    for shape in (syn.CFG4_PER_GPU, syn.CFG2, syn.CFG3):
        ctl = syn.make_controls(shape, 1004, kind)
        x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items() if k != "H"}
        out = {"config": shape.name, "f0": kind}
        for r in range(rounds):
            for path, label in ((1, "frame"), (0, "chunk")):
                set_path(path)
                run(x, shape)
                ddsp._lib.profile_enable(8 * steps + 8)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    run(x, shape)
                e1.record()
                torch.cuda.synchronize()
                rec = {}
                for name, ms in ddsp._lib.profile_read():
                    rec.setdefault(name, []).append(ms)
                ddsp._lib.profile_enable(0)
                out.setdefault(label, []).append({"ms": round(e0.elapsed_time(e1) / steps, 4),
                                                  **{k: round(float(np.mean(v)), 4) for k, v in rec.items()}})
        set_path(0)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:] or ["parity", "time"]
    ok = True
    if "parity" in args:
        ok = parity()
    if "time" in args:
        timing("all_live")
    if "musical" in args:
        timing("musical")
    sys.exit(0 if ok else 1)
