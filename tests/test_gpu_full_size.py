"""Full-size (BASELINE.json configs[1] / configs[3]-shard) property tests of the HIP path: the oracle cannot run
512 x 64000 x 100 in seconds, so beyond sampled rows these use size-independent properties of the path:
row independence / permutation equivariance (bit-exact), linearity in the loudness and in the filter
magnitudes, determinism, and hipGraph capture/replay equality."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402


def controls(shape, seed, kind="all_live"):
    ctl = syn.make_controls(shape, seed, kind)
    return ctl, {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}


@pytest.mark.parametrize("shape,seed,kind", [(syn.CFG2, 1002, "all_live"), (syn.CFG4_PER_GPU, 1004, "musical")])
def test_oscillator_full_size_properties(shape, seed, kind):
    ctl, x = controls(shape, seed, kind)
    L = ddsp._lib.lib()
    y, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
    assert y.shape == (shape.batch, shape.samples) and bool(torch.isfinite(y).all())
    # determinism
    y2, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
    assert torch.equal(y, y2)
    # rows are independent: a permuted batch gives the permuted result -- bit for bit when every wavefront takes the same walk
    # (all-live f0: nothing to skip).  With silent harmonics the chunked form groups rows by how many harmonic slots they walk,
    # and the wavefront at a group boundary walks its rows with the longer group's variant (quotient reuse on or off: the sine
    # argument then differs by one exact period, v_sin_f32's result by an ulp), so a row's last bits depend on its neighbours.
    perm = torch.randperm(shape.batch, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    yp, _, _ = ddsp.osc_forward(x["f0"][perm], x["c"][perm], x["a"][perm], shape.hop, shape.sample_rate)
    if kind == "all_live":
        assert torch.equal(yp, y[perm])
    else:
        assert float((yp - y[perm]).abs().max()) <= 1e-6
        assert L.ddsp_osc_set_path(1) == 0   # the frame kernels pick one variant per batch: bit-exact there
        try:
            yf, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
            yfp, _, _ = ddsp.osc_forward(x["f0"][perm], x["c"][perm], x["a"][perm], shape.hop, shape.sample_rate)
        finally:
            L.ddsp_osc_set_path(0)
        assert torch.equal(yfp, yf[perm])
        assert float((yf - y).abs().max()) <= 1e-6
    # a sub-batch equals the corresponding rows: bit for bit under the same tiling and the same form (grid-independent arithmetic),
    # and to rounding when the small problem picks fewer harmonics per lane (different summation order over k) or, 7 rows
    # filling one row block of 8 to less than 88 %, the frame kernels
    ys, _, _ = ddsp.osc_forward(x["f0"][3:10], x["c"][3:10], x["a"][3:10], shape.hop, shape.sample_rate)
    assert float((ys - y[3:10]).abs().max()) <= 2e-6
    assert L.ddsp_osc_set_tiling(13) == 0 and L.ddsp_osc_set_path(2) == 0
    try:
        ya, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
        yb, _, _ = ddsp.osc_forward(x["f0"][3:10], x["c"][3:10], x["a"][3:10], shape.hop, shape.sample_rate)
    finally:
        L.ddsp_osc_set_tiling(0)
        L.ddsp_osc_set_path(0)
    if kind == "all_live":
        assert torch.equal(yb, ya[3:10])
    else:
        assert float((yb - ya[3:10]).abs().max()) <= 1e-6
    # linear in the loudness control: y(2a) == 2 y(a) exactly (power-of-two scaling commutes with every rounding)
    yl, _, _ = ddsp.osc_forward(x["f0"], x["c"], 2.0 * x["a"], shape.hop, shape.sample_rate)
    assert torch.equal(yl, 2.0 * y)
    # harmonic amplitudes are normalised: scaling c by a power of two changes nothing
    yc, _, _ = ddsp.osc_forward(x["f0"], 4.0 * x["c"], x["a"], shape.hop, shape.sample_rate)
    assert torch.equal(yc, y)
    # sampled rows against the oracle
    rows = [0, shape.batch // 2, shape.batch - 1]
    ref = oracle.osc_forward(ctl["f0"][rows], ctl["c"][rows], ctl["a"][rows], shape.hop, shape.sample_rate)
    assert np.max(np.abs(y[rows].cpu().numpy() - ref)) <= 1e-5


def test_noise_full_size_properties():
    shape = syn.CFG4_PER_GPU
    ctl, x = controls(shape, 1004)
    y = ddsp.noise_forward(x["H"], shape.hop, seed=5)
    assert y.shape == (shape.batch, shape.samples) and bool(torch.isfinite(y).all())
    assert torch.equal(y, ddsp.noise_forward(x["H"], shape.hop, seed=5))
    # linear in H for a fixed draw: y(H1 + H2) == y(H1) + y(H2) up to fp32 rounding of the sums
    h2 = torch.flip(x["H"], dims=[2])
    lhs = ddsp.noise_forward(x["H"] + h2, shape.hop, seed=5)
    rhs = y + ddsp.noise_forward(h2, shape.hop, seed=5)
    assert float((lhs - rhs).abs().max()) <= 2e-6
    # frames are independent: a permuted batch with the same per-frame draw -> injected draw path
    u = torch.rand(8, shape.frames, shape.hop, device="cuda")
    ya = ddsp.noise_forward(x["H"][:8], shape.hop, uniform=u)
    idx = torch.tensor([7, 3, 0, 1, 6, 2, 5, 4], device="cuda")
    yb = ddsp.noise_forward(x["H"][:8][idx], shape.hop, uniform=u[idx])
    assert torch.equal(yb, ya[idx])
    ref = oracle.noise_forward(ctl["H"][:2], u[:2].cpu().numpy(), shape.hop)
    assert np.max(np.abs(ya[:2].cpu().numpy() - ref)) <= 2e-6
    # the configuration bench.py times -- the in-kernel Philox draw -- three whole rows against the oracle, which regenerates
    # the stream: row b starts at counter b * T * hop/4
    quads = shape.hop // 4
    for b in (0, shape.batch // 2, shape.batch - 1):
        ref = oracle.noise_forward(ctl["H"][b:b + 1], None, shape.hop, seed=5, offset=b * shape.frames * quads)
        assert np.max(np.abs(y[b:b + 1].cpu().numpy() - ref)) <= 2e-6, b


def test_cfg3_shape_slice_against_oracle():
    # configs[2]: 48 kHz, hop 512, 200 harmonics, 257 noise bands -- a 2-row, 1-second slice end to end
    shape = syn.SynthShape("cfg3_slice", 2, 48000, 512, 94, 200, 257)
    ctl, x = controls(shape, 1003)
    u = np.random.default_rng(3).random((2, 94, 512), dtype=np.float32)
    y, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], 512, 48000)
    ddsp.noise_forward(x["H"], 512, uniform=torch.from_numpy(u).cuda(), out=y, accumulate=True)
    ref = oracle.osc_forward(ctl["f0"], ctl["c"], ctl["a"], 512, 48000) + oracle.noise_forward(ctl["H"], u, 512)
    assert np.max(np.abs(y.cpu().numpy() - ref)) <= 1e-5


def test_hipgraph_capture_and_replay():
    # the launch path allocates nothing and never synchronises: it can be captured and replayed
    shape = syn.SynthShape("graph", 4, 16000, 128, 60, 100, 65)
    _, x = controls(shape, 9, "musical")

    def run():
        y, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
        return ddsp.noise_forward(x["H"], shape.hop, seed=11, out=y, accumulate=True)

    eager = run().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()                                   # warm-up on the capture stream (lazy one-time attribute calls)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = run()
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
    x["a"].mul_(2.0)                            # graphs read the live input buffers
    graph.replay()
    torch.cuda.synchronize()
    assert not torch.equal(out, eager)


def test_graphed_synth_offline_and_live():
    class Conf:
        n_harmonics, sample_rate, hop_length = 60, 16000, 128

    shape = syn.SynthShape("g", 1, 16000, 128, 40, 60, 65)
    ctl, x = controls(shape, 21, "musical")
    gs = ddsp.GraphedSynth(Conf, 1, 40, 65, noise_seed=3)
    y = gs(x).clone()
    eager, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], 128, 16000)
    ddsp.noise_forward(x["H"], 128, seed=3, out=eager, accumulate=True)
    assert torch.equal(y, eager)
    # the second replay draws fresh noise: the Philox offset lives on the device and advances inside the graph,
    # exactly like un-captured calls with offset = call index x draws per call
    eager2, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], 128, 16000)
    ddsp.noise_forward(x["H"], 128, seed=3, offset=1 * 40 * 32, out=eager2, accumulate=True)
    y2 = gs(x)
    assert torch.equal(y2, eager2) and not torch.equal(y2, eager)
    # live: the graph carries the oscillator state exactly like OscillatorBank.live
    gl = ddsp.GraphedSynth(Conf, 1, 4, 65, live=True, noise_seed=3)
    osc = ddsp.OscillatorBank(Conf).cuda()
    for call in range(3):
        ctl2, x2 = controls(syn.SynthShape("l", 1, 16000, 128, 4, 60, 65), 30 + call, "musical")
        ref = osc.live(x2)
        ddsp.noise_forward(x2["H"], 128, seed=3, offset=call * 4 * 32, out=ref, accumulate=True)
        assert torch.equal(gl(x2), ref)
        assert torch.equal(gl.state, osc.last_phases.data)


def test_cfg3_whole_workload():
    """BASELINE.json configs[2] as written: batch 512, 48 kHz, hop 512, 375 frames (4 s), 200 harmonics, 257 noise bands --
    the K=13/G=16 oscillator tiling (chunked form, 8 000-sample chunks), 4x longer phase drift than the 1 s fixture G4, 16-frames-per-workgroup noise tiles.
    The oracle does three whole rows in seconds; the other 509 rows are covered by size-independent properties."""
    shape = syn.CFG3
    ctl, x = controls(shape, 1003)
    y, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
    assert y.shape == (512, 192000) and bool(torch.isfinite(y).all())
    y2, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
    assert torch.equal(y, y2)                                                    # determinism
    perm = torch.randperm(shape.batch, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    yp, _, _ = ddsp.osc_forward(x["f0"][perm], x["c"][perm], x["a"][perm], shape.hop, shape.sample_rate)
    assert torch.equal(yp, y[perm])                                              # row permutation equivariance, bit for bit
    del yp, y2
    yl, _, _ = ddsp.osc_forward(x["f0"], x["c"], 2.0 * x["a"], shape.hop, shape.sample_rate)
    assert torch.equal(yl, 2.0 * y)                                              # loudness linearity (power-of-two: exact)
    del yl
    rows = [0, 255, 511]
    ref = oracle.osc_forward(ctl["f0"][rows], ctl["c"][rows], ctl["a"][rows], shape.hop, shape.sample_rate)
    assert np.max(np.abs(y[rows].cpu().numpy() - ref)) <= 1e-5                   # three whole 4 s rows against the oracle
    # noise: the whole batch with the in-kernel draw (determinism, H linearity), three rows with an injected draw vs the oracle
    n1 = ddsp.noise_forward(x["H"], shape.hop, seed=9)
    assert n1.shape == y.shape and bool(torch.isfinite(n1).all())
    assert torch.equal(n1, ddsp.noise_forward(x["H"], shape.hop, seed=9))
    n2 = ddsp.noise_forward(2.0 * x["H"], shape.hop, seed=9)
    assert torch.equal(n2, 2.0 * n1)
    del n2
    # ... and three whole rows of that in-kernel draw against the oracle regenerating the Philox stream (what bench.py's cfg3 line times)
    for b in rows:
        dref = oracle.noise_forward(ctl["H"][b:b + 1], None, shape.hop, seed=9, offset=b * shape.frames * (shape.hop // 4))
        assert np.max(np.abs(n1[b:b + 1].cpu().numpy() - dref)) <= 2e-6 * max(1.0, float(np.max(np.abs(dref)))), b
    u = np.random.default_rng(33).random((3, shape.frames, shape.hop), dtype=np.float32)
    hrows = x["H"][rows].contiguous()
    got = ddsp.noise_forward(hrows, shape.hop, uniform=torch.from_numpy(u).cuda())
    nref = oracle.noise_forward(ctl["H"][rows], u, shape.hop)
    assert np.max(np.abs(got.cpu().numpy() - nref)) <= 2e-6 * max(1.0, float(np.max(np.abs(nref))))
    # the fused epilogue on the full buffer: harmonics + noise accumulated in place == the two parts added
    total = y.clone()
    ddsp.noise_forward(x["H"], shape.hop, seed=9, out=total, accumulate=True)
    assert float((total - (y + n1)).abs().max()) <= 1e-6


def test_cfg5_per_gpu_train_step_full_shape():
    """BASELINE.json configs[4], one GPU's share: batch 32, 500 frames (4 s at 16 kHz, hop 128), 100 harmonics, 65 noise bands,
    controller widths 512 (GRU 8 groups x 32 workgroups).  One whole `train_step` (decoder -> HIP synth -> reverb -> 6-scale
    spectral loss -> backward -> Adam) in GRU debug mode (every recurrence launch's status word is checked: a time-out
    raises), finite loss and gradients, and the batch gradient equals the mean of the two half-batch gradients."""
    from ddsp_pytorch_amd import gru as gru_mod

    class Conf:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 100, 65, 16000, 128
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1

    B, T = 32, 500
    torch.manual_seed(0)
    model = ddsp.Decoder(Conf, noise_rng="device", seed=1).cuda()
    with torch.no_grad():
        model.reverb.wet.fill_(-1.0)
    loss_fn = ddsp.MSSLoss().cuda()
    rng = np.random.default_rng(2000)
    batch = {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (B, T, 1)).astype(np.float32)).cuda(),
             "loudness": torch.from_numpy(rng.uniform(-1, 1, (B, T, 1)).astype(np.float32)).cuda(),
             "f0": torch.from_numpy(syn.musical_f0(rng, B, T)).cuda(),
             "audio": torch.from_numpy((0.1 * rng.standard_normal((B, T * 128))).astype(np.float32)).cuda()}
    draw = torch.from_numpy(rng.random((B, T, 128), dtype=np.float32)).cuda()
    plain = model.noise.forward
    rows = [slice(0, B)]
    model.noise.forward = lambda x, noise=None, out=None: plain(x, noise=draw[rows[0]], out=out)   # the draw travels with the rows
    params = [p for p in model.parameters() if p.requires_grad]

    def grads(sl):
        rows[0] = sl
        for p in params:
            p.grad = None
        loss = loss_fn(model({k: v[sl] for k, v in batch.items()}), batch["audio"][sl])
        loss.backward()
        return float(loss.detach()), [p.grad.detach().clone() for p in params]

    old = gru_mod.set_debug(True)
    try:
        l_all, g_all = grads(slice(0, B))
        l_a, g_a = grads(slice(0, B // 2))
        l_b, g_b = grads(slice(B // 2, B))
        assert np.isfinite(l_all) and all(bool(torch.isfinite(g).all()) for g in g_all)
        assert abs(0.5 * (l_a + l_b) - l_all) <= 1e-5 * abs(l_all)
        worst = 0.0
        for ga, gb, gf in zip(g_a, g_b, g_all):
            worst = max(worst, float((0.5 * (ga + gb) - gf).abs().max()) / (float(gf.abs().max()) + 1e-12))
        assert worst <= 1e-4, worst        # mean-reduced loss: batch gradient = mean of equal halves' gradients (fp32 reductions)
        # and the optimiser step itself on the whole shape
        rows[0] = slice(0, B)
        opt = torch.optim.Adam(params, lr=1e-3)
        l0, nbytes = ddsp.train_step(model, loss_fn, opt, batch)
        l1, _ = ddsp.train_step(model, loss_fn, opt, batch)
        assert nbytes == 4 * sum(p.numel() for p in params) and np.isfinite(float(l0)) and np.isfinite(float(l1))
    finally:
        gru_mod.set_debug(old)


def test_graphed_synth_hop512_uses_the_fft_noise_form():
    """hipGraph capture of harmonics + noise at the 48 kHz shape (hop 512, 257 bands): the noise runs in the in-LDS FFT form
    with the Philox offset read from the device counter at replay time -- replays equal the eager calls, draw after draw."""
    class Conf:
        n_harmonics, sample_rate, hop_length = 200, 48000, 512

    shape = syn.SynthShape("g512", 2, 48000, 512, 9, 200, 257)          # 18 frames: nine frame pairs
    _, x = controls(shape, 23)
    gs = ddsp.GraphedSynth(Conf, 2, 9, 257, noise_seed=4)
    per_call = 2 * 9 * (512 // 4)
    for call in range(3):
        eager, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], 512, 48000)
        ddsp.noise_forward(x["H"], 512, seed=4, offset=call * per_call, out=eager, accumulate=True)
        assert torch.equal(gs(x), eager), call
