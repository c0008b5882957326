#!/usr/bin/env python3
"""The synthesis step (OscillatorBank.forward + FilteredNoise accumulated) issued launch by launch against ONE hipGraph replay of the
same launches (GraphedSynth.run): ms per step at the headline shape and at cfg2, same box, interleaved, sustained clocks."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

out = {}
for name, shape in (("headline", syn.CFG4_PER_GPU), ("cfg2", syn.CFG2)):
    ctl = syn.make_controls(shape, 1004, "all_live")
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
    osc = ddsp.OscillatorBank(bench.Conf(shape)).cuda()
    g = ddsp.GraphedSynth(bench.Conf(shape), shape.batch, shape.frames, shape.n_noise_filters)
    for k in ("f0", "c", "a", "H"):
        getattr(g, k).copy_(x[k])

    def eager(i):
        y = osc(x)
        ddsp.noise_forward(x["H"], shape.hop, seed=7, offset=i << 32, out=y, accumulate=True)

    def graphed(i):
        g.run()

    res = {"eager": [], "graph": []}
    for rnd in range(3):
        for label, fn in (("eager", eager), ("graph", graphed)):
            bench.settle_clock(fn, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                fn(i)
            torch.cuda.synchronize()
            res[label].append(round(1e3 * (time.perf_counter() - t0) / 200, 4))
    out[name] = res
print(json.dumps(out))
