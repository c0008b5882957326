"""GPU parity tests of the CHUNKED oscillator form (csrc/ddsp_osc_chunk.hip: the production path for power-of-two hops >= 64
with 4 / 8 / 16 lanes per row) against the CPU oracle, against the frame kernels (ddsp_osc_set_path(1)) and through the
properties the decomposition must not break: chunk boundaries anywhere inside a segment, clip ends, ragged row blocks,
silent harmonics (rows ordered per chunk), the exact-modulo repair of declined chunks, NaN frames.

Tolerance: audio <= 1e-5 absolute (BASELINE.json north_star); measured <= 5e-7.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402

TOL_Y = 1e-5


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.fixture
def lib():
    L = ddsp._lib.lib()
    assert L.ddsp_test_hooks_enabled() == 1, "DDSP_TEST_HOOKS=1 must be set before the library is loaded (tests/conftest.py)"
    yield L
    ddsp._lib.check(L.ddsp_osc_set_tiling(0), "ddsp_osc_set_tiling")
    ddsp._lib.check(L.ddsp_osc_set_path(0), "ddsp_osc_set_path")


def run(f0, c, a, hop, sr):
    """Forward into a buffer that held NaN (a path that writes nothing cannot pass by inheriting an earlier result)."""
    B, T, _ = c.shape
    poison = torch.full((B, T * hop), float("nan"), device="cuda")
    torch.cuda.synchronize()
    del poison
    y, _, _ = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def force_chunked(L, K, B, T, H, hop, sr):
    """Pins the harmonics per lane (small problems would otherwise pick 32+ lanes per row, which the chunked form leaves
    to the frame kernels) and returns the launch plan, which must say `chunked`."""
    ddsp._lib.check(L.ddsp_osc_set_tiling(K), "ddsp_osc_set_tiling")
    ddsp._lib.check(L.ddsp_osc_set_path(2), "ddsp_osc_set_path")      # ... and batches that fill their last row block badly
    plan = ddsp._lib.osc_plan(B, T, H, hop, sr)
    assert plan["chunked"] == 1 and plan["harmonics_per_lane"] == K, plan
    return plan


def controls(B, T, H, sr, kind, seed):
    shape = syn.SynthShape("t", B, sr, 128, T, H, 65)
    ctl = syn.make_controls(shape, seed, kind)
    return ctl["f0"], ctl["c"], ctl["a"]


# (B, T, H, hop, sr, K, f0 kind): lanes per row 4 / 8 / 16, one to many frames, ragged row blocks, every hop class
CASES = [
    (9, 37, 100, 128, 16000, 13, "all_live"),     # 8 lanes, 2 row blocks (one ragged)
    (11, 53, 100, 128, 16000, 13, "musical"),
    (3, 19, 60, 64, 16000, 15, "musical"),        # 4 lanes, smallest hop
    (5, 23, 200, 512, 48000, 13, "all_live"),     # 16 lanes
    (2, 11, 180, 512, 44100, 12, "musical"),      # the reference's default shape (config/default.py:13-19), 16 lanes
    (17, 7, 50, 256, 16000, 13, "musical"),       # 4 lanes
    (4, 1, 100, 128, 16000, 13, "all_live"),      # one frame: both clip ends inside one chunk
    (1, 2, 100, 128, 16000, 13, "musical"),
    (1, 500, 100, 128, 16000, 13, "all_live"),    # one row: seven of the eight row groups of every wavefront idle
    (2, 9, 100, 1024, 16000, 13, "musical"),
    (3, 5, 64, 4096, 16000, 16, "all_live"),
    (64, 125, 100, 128, 16000, 13, "musical"),    # many rows: several row blocks per chunk index, rows reordered per chunk
]


@pytest.mark.parametrize("B,T,H,hop,sr,K,kind", CASES)
def test_chunked_vs_oracle_and_frame_kernels(lib, B, T, H, hop, sr, K, kind):
    f0, c, a = controls(B, T, H, sr, kind, 4242 + B + T)
    force_chunked(lib, K, B, T, H, hop, sr)
    y_chunk = run(f0, c, a, hop, sr)
    ddsp._lib.check(lib.ddsp_osc_set_path(1), "ddsp_osc_set_path")
    assert ddsp._lib.osc_plan(B, T, H, hop, sr)["chunked"] == 0
    y_frame = run(f0, c, a, hop, sr)
    rows = sorted(set([0, B // 2, B - 1]))
    ref = oracle.osc_forward(f0[rows], c[rows], a[rows], hop, sr)
    assert np.isfinite(y_chunk).all()
    assert np.max(np.abs(y_chunk[rows] - ref)) <= TOL_Y
    assert np.max(np.abs(y_chunk - y_frame)) <= 2e-6


@pytest.mark.parametrize("chunk_len", [128, 160, 224, 352, 1344, 4000])
def test_chunk_boundaries_anywhere_in_a_segment(lib, chunk_len, monkeypatch):
    # every offset of a chunk boundary inside a segment (multiples of 32 samples), several rounds of wavefronts
    monkeypatch.setenv("DDSP_OSC_CHUNK_LEN", str(chunk_len - chunk_len % 32))
    B, T, H, hop, sr = 10, 41, 100, 128, 16000
    f0, c, a = controls(B, T, H, sr, "musical", 99)
    plan = force_chunked(lib, 13, B, T, H, hop, sr)
    assert plan["chunk_samples"] == chunk_len - chunk_len % 32
    y = run(f0, c, a, hop, sr)
    ref = oracle.osc_forward(f0, c, a, hop, sr)
    assert np.max(np.abs(y - ref)) <= TOL_Y


def test_chunked_declined_chunks_are_repaired_exactly(lib):
    # phases beyond the fast modulo's exact range (1e7 rad), negative f0 and a NaN f0: the fast kernel declines those wave
    # tasks and walks them a second time with the exact modulo; masked harmonics still accumulate phase
    rng = np.random.default_rng(5)
    B, T, H, hop, sr = 4, 300, 100, 512, 8000
    f0 = rng.uniform(30.0, 39.0, (B, T, 1)).astype(np.float32)
    f0[1] *= 100.0                      # ~3.5 kHz: harmonic 1 audible, harmonic 100 advances 275 rad per sample -> 4e7 rad
    f0[2, 50:60, 0] = -220.0            # phases run backwards
    f0[3, 7, 0] = np.nan
    c = rng.uniform(0.1, 1, (B, T, H)).astype(np.float32)
    a = rng.uniform(0.1, 1, (B, T, 1)).astype(np.float32)
    force_chunked(lib, 13, B, T, H, hop, sr)
    y = run(f0, c, a, hop, sr)
    ref = oracle.osc_forward(f0, c, a, hop, sr)
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(y), finite)
    assert finite[1].all() and np.max(np.abs(y[1] - ref[1])) <= TOL_Y
    assert np.max(np.abs(y[finite] - ref[finite])) <= TOL_Y
    assert finite[0].all() and finite[2].all() and not finite[3].all() and finite[3, :512].all()


def test_chunked_nan_frame_and_silent_rows(lib):
    # an all-masked frame is 0/0 = NaN for the samples that interpolate from it (:33) and only for those; a row whose
    # harmonics are all above Nyquist elsewhere (amplitude 0 everywhere -> walks the fewest slots) stays finite around it
    B, T, H, hop, sr = 9, 12, 100, 128, 16000
    f0, c, a = controls(B, T, H, sr, "musical", 3)
    f0[4, 5, 0] = 9000.0                # every harmonic above Nyquist in frame 5 of row 4
    f0[6, :, 0] = 7000.0                # row 6: only harmonic 1 audible
    force_chunked(lib, 13, B, T, H, hop, sr)
    y = run(f0, c, a, hop, sr)
    ref = oracle.osc_forward(f0, c, a, hop, sr)
    assert np.array_equal(np.isnan(y), np.isnan(ref))
    assert np.isnan(ref[4]).any() and not np.isnan(ref[4]).all()
    ok = ~np.isnan(ref)
    assert np.max(np.abs(y[ok] - ref[ok])) <= TOL_Y


def test_chunked_is_deterministic_and_linear_in_loudness(lib):
    B, T, H, hop, sr = 24, 60, 100, 128, 16000
    f0, c, a = controls(B, T, H, sr, "musical", 8)
    force_chunked(lib, 13, B, T, H, hop, sr)
    y = run(f0, c, a, hop, sr)
    assert np.array_equal(y, run(f0, c, a, hop, sr))
    assert np.array_equal(run(f0, c, 2.0 * a, hop, sr), 2.0 * y)       # power-of-two scaling commutes with every rounding
    assert np.array_equal(run(f0, 4.0 * c, a, hop, sr), y)              # amplitudes are normalised (:33)


def test_plan_reports_the_launch(lib):
    plan = ddsp._lib.osc_plan(512, 500, 100, 128, 16000)
    assert plan["chunked"] == 1 and plan["harmonics_per_lane"] == 13 and plan["lanes_per_row"] == 8
    rows_per_wave = 64 // plan["lanes_per_row"]
    assert plan["row_blocks"] == (512 + rows_per_wave - 1) // rows_per_wave
    assert plan["chunk_samples"] % 32 == 0 and plan["chunk_samples"] >= 128
    assert plan["chunks_per_row"] == -(-500 * 128 // plan["chunk_samples"])
    # every (row block, chunk) task resident at once on this device
    tasks = plan["row_blocks"] * plan["chunks_per_row"]
    assert tasks <= plan["compute_units"] * plan["workgroups_per_unit"] * 4
    # shapes the chunked form leaves to the frame kernels: odd hop, short hop, a batch that fills 9 of 16 row slots
    assert ddsp._lib.osc_plan(512, 500, 100, 100, 16000)["chunked"] == 0
    assert ddsp._lib.osc_plan(512, 500, 100, 32, 16000)["chunked"] == 0
    assert ddsp._lib.osc_plan(9, 4000, 100, 128, 16000)["chunked"] == 0
    assert ddsp._lib.osc_plan(15, 4000, 100, 128, 16000)["chunked"] == 1


def test_backward_refuses_a_chunked_scratch(lib):
    # ddsp_osc_backward re-walks the frame-rate layout; a forward that was not asked to keep it leaves a chunked scratch,
    # and the backward must fail loudly (NaN gradients), not differentiate garbage
    B, T, H, hop, sr = 16, 40, 100, 128, 16000
    f0, c, a = (dev(v) for v in controls(B, T, H, sr, "musical", 21))
    force_chunked(lib, 13, B, T, H, hop, sr)
    g = torch.randn(B, T * hop, device="cuda")
    _, _, _, scratch = ddsp.osc_forward(f0, c, a, hop, sr, return_scratch=True)     # asks for the frame-form scratch
    gc, ga = ddsp.harmonic_oscillator.osc_backward(g, f0, c, a, scratch, hop, sr)
    assert bool(torch.isfinite(gc).all()) and bool(torch.isfinite(ga).all())
    y = torch.empty(B, T * hop, device="cuda")
    scratch2 = torch.empty_like(scratch)
    rc = lib.ddsp_osc_forward_ex(f0.data_ptr(), c.data_ptr(), a.data_ptr(), y.data_ptr(), scratch2.data_ptr(), None, None, None,
                                 B, T, H, hop, sr, ctypes.c_uint(0), None)
    assert rc == 0
    gc2, ga2 = ddsp.harmonic_oscillator.osc_backward(g, f0, c, a, scratch2, hop, sr)
    assert bool(torch.isnan(gc2).all()) and bool(torch.isnan(ga2).all())


def test_chunked_form_is_graph_capturable_and_stream_safe(lib):
    """The chunked launches (occupancy query cached per device, four kernels, no host synchronisation) captured into a hipGraph on
    a side stream and replayed -- on live input buffers -- equal the eager result bit for bit; and a launch from a worker thread
    on its own stream (the JACK callback's situation, rt/synth.py:50-52) does too."""
    import threading
    B, T, H, hop, sr = 16, 40, 100, 128, 16000
    f0, c, a = (dev(v) for v in controls(B, T, H, sr, "musical", 77))
    force_chunked(lib, 13, B, T, H, hop, sr)
    eager, _, _ = ddsp.osc_forward(f0, c, a, hop, sr)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            ddsp.osc_forward(f0, c, a, hop, sr)            # warm-up outside the capture (allocator, occupancy cache)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y_static, _, _ = ddsp.osc_forward(f0, c, a, hop, sr)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_static, eager)
    a.mul_(2.0)                                             # graphs read the live input buffers
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_static, 2.0 * eager)
    a.mul_(0.5)
    out = {}

    def worker():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            out["y"] = ddsp.osc_forward(f0, c, a, hop, sr)[0]
        s.synchronize()

    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert torch.equal(out["y"], eager)


def test_clock_probe_reads_a_plausible_shader_clock(lib):
    B, T, H, hop, sr = 64, 125, 100, 128, 16000
    f0, c, a = (dev(v) for v in controls(B, T, H, sr, "all_live", 5))
    force_chunked(lib, 13, B, T, H, hop, sr)
    _, _, _, scratch = ddsp.osc_forward(f0, c, a, hop, sr, return_scratch=True, keep_frame_scratch=False)
    ghz = ddsp._lib.osc_clock(scratch, B, T, H, hop, sr)
    assert 0.5 <= ghz <= 3.0, ghz           # MI355X: up to 2.4 GHz
    ddsp._lib.check(lib.ddsp_osc_set_path(1), "ddsp_osc_set_path")      # the frame kernels stamp the same words
    _, _, _, scratch = ddsp.osc_forward(f0, c, a, hop, sr, return_scratch=True, keep_frame_scratch=False)
    ghz2 = ddsp._lib.osc_clock(scratch, B, T, H, hop, sr)
    assert 0.5 <= ghz2 <= 3.0, ghz2
