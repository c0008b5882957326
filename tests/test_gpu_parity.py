"""GPU parity tests proper: the HIP path (through the C ABI) against the golden fixtures captured
from the reference and against the CPU oracle on the same seeded inputs.

Tolerances: wrapped phases bit-exact (debug output of ddsp_osc_forward); audio <= 1e-5 absolute
(BASELINE.json north_star); the live state `last_phases` bit-exact; noise <= 2e-6 (+ the oracle's own
5e-7 distance from the reference's fp32 FFT, so 1e-5 is the contract and we assert tighter).
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402

TOL_Y = 1e-5


class Conf:
    def __init__(self, n_harmonics, sample_rate, hop_length):
        self.n_harmonics, self.sample_rate, self.hop_length = n_harmonics, sample_rate, hop_length


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def run_osc(g, debug=False, prefix=""):
    y, _, phi = ddsp.osc_forward(dev(g["f0"]), dev(g["c"]), dev(g["a"]), int(g["hop"]), int(g["sample_rate"]),
                                 debug_phases=debug)
    torch.cuda.synchronize()
    return y.cpu().numpy(), (phi.cpu().numpy() if debug else None)


def test_native_library_is_loaded():
    L = ddsp._lib.lib()
    assert L.ddsp_hip_abi_version() == ddsp._lib.ABI_VERSION == 4
    with open("/proc/self/maps") as f:
        assert "libddsp_hip.so" in f.read()


INTERMEDIATE = ["g1_osc_tiny", "g5b_osc_nyquist_finite", "g6_osc_hop100", "g6_osc_hop441", "g6_osc_hop3",
                "g6_osc_hop7", "g6_osc_hop160", "g6_osc_hop480", "g6_osc_single_frame"]


@pytest.mark.parametrize("name", INTERMEDIATE)
def test_osc_phases_bit_exact_and_audio(name):
    g = load_golden(name)
    y, phi = run_osc(g, debug=True)
    assert np.array_equal(bits(phi), bits(g["phi"])), "wrapped phases differ from the reference"
    assert np.max(np.abs(y - g["y"])) <= TOL_Y
    y_fast, _ = run_osc(g)                       # production path (fast modulo)
    assert np.max(np.abs(y_fast - g["y"])) <= TOL_Y


@pytest.mark.parametrize("name", ["g2_osc_cfg2_live", "g3_osc_cfg2_musical", "g4_osc_cfg3_1s", "g2b_osc_cfg1"])
def test_osc_long_clips(name):
    g = load_golden(name)
    y, _ = run_osc(g)
    err = np.max(np.abs(y - g["y"]))
    assert err <= TOL_Y, err
    if "phi_sub" in g:
        _, phi = run_osc(g, debug=True)
        assert np.array_equal(bits(phi[:, g["phi_idx"], :]), bits(g["phi_sub"]))


@pytest.mark.parametrize("K", [4, 8, 12, 13, 15, 16, 20, 23, 25])
def test_osc_every_tiling(K):
    g = load_golden("g3_osc_cfg2_musical")
    L = ddsp._lib.lib()
    assert L.ddsp_osc_set_tiling(K) == 0
    try:
        y, _ = run_osc(g)
    finally:
        L.ddsp_osc_set_tiling(0)
    assert np.max(np.abs(y - g["y"])) <= TOL_Y


def test_osc_nyquist_nan_frame():
    g = load_golden("g5_osc_nyquist")
    y, phi = run_osc(g, debug=True)
    assert np.array_equal(bits(phi), bits(g["phi"]))
    assert np.array_equal(np.isnan(y), np.isnan(g["y"])) and np.isnan(y).any()
    ok = ~np.isnan(y)
    assert np.max(np.abs(y[ok] - g["y"][ok])) <= TOL_Y


@pytest.mark.parametrize("name,calls,H", [("g7_osc_live", 3, 180), ("g7b_osc_live_batch2", 2, 16)])
def test_osc_live_module_state(name, calls, H):
    g = load_golden(name)
    osc = ddsp.OscillatorBank(Conf(H, int(g["sample_rate"]), int(g["hop"]))).cuda()
    assert osc.last_phases.dtype == torch.int64 and set(osc.state_dict()) == {"harmonics", "last_phases"}
    for k in range(calls):
        y = osc.live({"f0": dev(g[f"f0_{k}"]), "c": dev(g[f"c_{k}"]), "a": dev(g[f"a_{k}"])})
        assert np.array_equal(bits(osc.last_phases.detach().cpu().numpy()), bits(g[f"last_phases_{k}"]))
        assert np.max(np.abs(y.cpu().numpy() - g[f"y_{k}"])) <= TOL_Y


def test_osc_module_forward_and_inputs_untouched():
    g = load_golden("g1_osc_tiny")
    osc = ddsp.OscillatorBank(Conf(8, 16000, 64)).cuda()
    x = {"f0": dev(g["f0"]), "c": dev(g["c"]), "a": dev(g["a"])}
    keep = {k: v.clone() for k, v in x.items()}
    y = osc(x)
    assert y.shape == (2, 16 * 64) and y.dtype == torch.float32 and y.is_cuda
    assert all(torch.equal(x[k], keep[k]) for k in x)
    assert np.max(np.abs(y.cpu().numpy() - g["y"])) <= TOL_Y


def test_osc_vs_oracle_seeded_batch():
    # oracle on the same seeded inputs, a batch that spans several workgroups and a ragged last one
    shape = syn.SynthShape("t", 5, 16000, 128, 77, 100, 65)
    for kind in ("all_live", "musical"):
        ctl = syn.make_controls(shape, 31, kind)
        ref = oracle.osc_forward(ctl["f0"], ctl["c"], ctl["a"], 128, 16000)
        y, _, _ = ddsp.osc_forward(dev(ctl["f0"]), dev(ctl["c"]), dev(ctl["a"]), 128, 16000)
        assert np.max(np.abs(y.cpu().numpy() - ref)) <= TOL_Y


def test_osc_slow_path_large_phase_and_negative_f0():
    # phases beyond the fast-modulo range (masked harmonics still accumulate: App. C.1) and negative f0
    rng = np.random.default_rng(5)
    T, H, hop, sr = 40, 12, 512, 8000
    f0 = rng.uniform(3000, 3900, (2, T, 1)).astype(np.float32) * 40.0   # ~1e5 Hz: everything masked but harmonic 1..
    f0[0, :, 0] = rng.uniform(200, 300, T)                              # row 0 audible
    f0[1, 5:9, 0] = -220.0
    c = rng.uniform(0.1, 1, (2, T, H)).astype(np.float32)
    a = rng.uniform(0.1, 1, (2, T, 1)).astype(np.float32)
    ref, d = oracle.osc_forward(f0, c, a, hop, sr, debug=True)
    y, _, phi = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr, debug_phases=True)
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(y.cpu().numpy()), finite)
    y2, _, _ = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr)
    assert np.max(np.abs(y2.cpu().numpy()[finite] - ref[finite])) <= TOL_Y
    assert np.max(np.abs(y.cpu().numpy()[finite] - ref[finite])) <= TOL_Y


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g8_noise_*.npz"))))
def test_noise_injected_and_seeded(path):
    g = load_golden(os.path.basename(path)[:-4])
    hop = int(g["hop"])
    y = ddsp.noise_forward(dev(g["H"]), hop, uniform=dev(g["uniform"]))
    assert np.max(np.abs(y.cpu().numpy() - g["y"])) <= 2e-6
    mod = ddsp.FilteredNoise(Conf(1, 16000, hop))
    torch.manual_seed(int(g["seed"]))             # reference-compatible host RNG mode (filtered_noise.py:44-48)
    y2 = mod({"H": dev(g["H"])})
    assert np.max(np.abs(y2.cpu().numpy() - g["y"])) <= 2e-6


def test_noise_device_rng_statistics_and_accumulate():
    shape = syn.CFG2
    ctl = syn.make_controls(shape, 3, batch=4)
    H = dev(ctl["H"])
    y1 = ddsp.noise_forward(H, shape.hop, seed=123)
    y2 = ddsp.noise_forward(H, shape.hop, seed=123)
    y3 = ddsp.noise_forward(H, shape.hop, seed=124)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    assert abs(float(y1.mean())) < 0.02 and float(y1.std()) > 0.05
    base = torch.full_like(y1, 0.25)
    out = ddsp.noise_forward(H, shape.hop, seed=123, out=base.clone(), accumulate=True)
    assert torch.allclose(out, base + y1, atol=1e-6)


# ---- the in-kernel draw (rng='device', what bench.py times) against the oracle's Philox4x32-10 restatement -------------
# (the oracle's generator is pinned by the Random123 known-answer vectors: tests/test_oracle_golden.py)
FORMS = {"wave": (0, 128), "batched": (8, 128), "batched_hop64": (0, 64), "generic": (1, 128), "generic_ragged_hop": (1, 6), "batched_hop512": (2, 512), "fft": (0, 512), "fft_hop256": (4, 256)}


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("seed,offset", [(0, 0), (0x9E3779B97F4A7C15, (1 << 40) + 12345), (7, (1 << 32) - 5)])
def test_noise_device_draw_is_the_oracles_philox_stream(form, seed, offset):
    """F = 2, H = 1: z = irfft([1, 1]) = [1, 0] exactly, so the impulse response is a unit tap and y must BE the draw,
    x = 2u - 1 with u = (word >> 8) * 2^-24 -- sample for sample (bit-exact in the time-domain kernels, to FFT rounding in the
    FFT form).  seed 0 / offset 0 starts at the Random123 known-answer block; the large offsets carry into the counter's high word."""
    mode, hop = FORMS[form]
    B, T = 3, 7                                    # 21 frames: a ragged last tile / a half-empty last frame pair
    H = torch.ones(B, T, 2, device="cuda")
    L = ddsp._lib.lib()
    L.ddsp_noise_set_generic(mode)
    try:
        y = ddsp.noise_forward(H, hop, seed=seed, offset=offset).cpu().numpy()
    finally:
        L.ddsp_noise_set_generic(0)
    x = oracle.philox_uniform(seed, offset, B, T, hop).reshape(B, T * hop) * 2.0 - 1.0
    if form.startswith("fft"):
        assert np.max(np.abs(y - x)) <= 2e-6
    else:
        assert np.array_equal(y, x)
    if seed == 0 and offset == 0:                  # the first block is the published vector 6627e8d5 e169c58d bc57ac4c 9b00dbd8
        want = np.array([0x6627e8d5 >> 8, 0xe169c58d >> 8, 0xbc57ac4c >> 8, 0x9b00dbd8 >> 8], np.float64) / 2.0 ** 24 * 2.0 - 1.0
        assert np.max(np.abs(y[0, :4] - want)) <= (2e-6 if form.startswith("fft") else 0.0)


@pytest.mark.parametrize("hop,nf,mode", [(128, 65, 0), (128, 65, 8), (128, 65, 1), (64, 65, 0), (512, 257, 0), (512, 257, 2), (512, 195, 0), (256, 129, 0), (256, 129, 4), (24, 7, 0)])
def test_noise_device_rng_vs_oracle(hop, nf, mode):
    """Real filters with the in-kernel draw, every kernel form: the oracle regenerates the same Philox stream (seed, offset)."""
    rng = np.random.default_rng(hop * 3 + nf)
    B, T = 2, 13
    Hn = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32))
    seed, offset = 1234, (5 << 32) + 77
    L = ddsp._lib.lib()
    L.ddsp_noise_set_generic(mode)
    try:
        y = ddsp.noise_forward(dev(Hn), hop, seed=seed, offset=offset).cpu().numpy()
    finally:
        L.ddsp_noise_set_generic(0)
    ref = oracle.noise_forward(Hn, None, hop, seed=seed, offset=offset)
    assert np.max(np.abs(y - ref)) <= 2e-6 * max(1.0, float(np.max(np.abs(ref))))


def test_filtered_noise_module_device_stream_advances_like_the_oracle():
    """FilteredNoise(rng='device'): call k draws from offset = sum of the earlier calls' counters (B*T*ceil(hop/4))."""
    fn = ddsp.FilteredNoise(Conf(1, 16000, 128), rng="device", seed=99)
    rng = np.random.default_rng(12)
    off = 0
    for B, T in ((2, 5), (1, 3), (3, 4)):
        Hn = syn.controller_range(rng.standard_normal((B, T, 65), dtype=np.float32))
        y = fn({"H": dev(Hn)}).cpu().numpy()
        ref = oracle.noise_forward(Hn, None, 128, seed=99, offset=off)
        assert np.max(np.abs(y - ref)) <= 2e-6
        off += B * T * 32


def test_decoder_wiring_g11():
    # decoder.py:129-133: harmonics + noise (reverb is a "next" row, applied here by the fixture's own impulse)
    g = load_golden("g11_decoder_wiring")
    conf = Conf(100, 16000, 128)
    x = {k: dev(g[k]) for k in ("f0", "c", "a", "H")}
    harm = ddsp.OscillatorBank(conf).cuda()(x)
    assert np.max(np.abs(harm.cpu().numpy() - g["harm"])) <= TOL_Y
    both = ddsp.noise_forward(x["H"], 128, uniform=dev(g["uniform"]), out=harm, accumulate=True)
    assert np.max(np.abs(both.cpu().numpy() - (g["harm"] + g["noise"]))) <= TOL_Y


def test_cpu_tensors_fail_loudly():
    g = load_golden("g1_osc_tiny")
    with pytest.raises(ddsp._lib.DdspHipError):
        ddsp.osc_forward(torch.from_numpy(g["f0"]), torch.from_numpy(g["c"]), torch.from_numpy(g["a"]), 64, 16000)


@pytest.mark.parametrize("mode", [1, 2, 8])  # 1: one frame per workgroup; 2: batched direct form where the FFT form would run; 8: batched where the wavefront-private form would
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g8_noise_*.npz"))))
def test_noise_generic_kernel(path, mode):
    g = load_golden(os.path.basename(path)[:-4])
    L = ddsp._lib.lib()
    L.ddsp_noise_set_generic(mode)
    try:
        y = ddsp.noise_forward(dev(g["H"]), int(g["hop"]), uniform=dev(g["uniform"]))
    finally:
        L.ddsp_noise_set_generic(0)
    assert np.max(np.abs(y.cpu().numpy() - g["y"])) <= 2e-6


@pytest.mark.parametrize("hop,nf,B,T", [(128, 65, 3, 70), (64, 65, 2, 33), (256, 129, 1, 65), (128, 33, 1, 64), (8, 5, 2, 9), (136, 7, 1, 5),
                                         (512, 257, 1, 37), (512, 195, 2, 19), (1024, 65, 1, 9),
                                         # in-LDS FFT form (hop 256 / 512): impulse shorter than the hop (zero gap), tiny filters, odd frame counts
                                         (512, 129, 2, 9), (512, 3, 1, 5), (512, 256, 1, 3), (256, 65, 3, 7), (256, 128, 1, 1), (256, 2, 1, 2)])
def test_noise_vs_oracle_ragged_tiles(hop, nf, B, T):
    rng = np.random.default_rng(hop + nf)
    H = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32))
    u = rng.random((B, T, hop), dtype=np.float32)
    ref = oracle.noise_forward(H, u, hop)
    y = ddsp.noise_forward(dev(H), hop, uniform=dev(u))
    assert np.max(np.abs(y.cpu().numpy() - ref)) <= 2e-6
    if hop == 256:                                   # the FFT form is not the default at hop 256: take it explicitly as well
        L = ddsp._lib.lib()
        L.ddsp_noise_set_generic(4)
        try:
            y = ddsp.noise_forward(dev(H), hop, uniform=dev(u))
        finally:
            L.ddsp_noise_set_generic(0)
        assert np.max(np.abs(y.cpu().numpy() - ref)) <= 2e-6


@pytest.mark.parametrize("B,T", [(1, 1), (1, 15), (1, 16), (1, 17), (3, 11), (2, 500), (37, 3)])
@pytest.mark.parametrize("injected", [True, False])
def test_noise_wave_form_ragged_groups_vs_oracle(B, T, injected):
    """The wavefront-private hop-128 / 65-band form (16 frames per wavefront-iteration, persistent): frame counts around the
    group size, a long row, many short rows; injected draw and in-kernel draw; plain and accumulating; against the oracle and
    (same draw) against the batched kernel it replaces."""
    rng = np.random.default_rng(1000 * B + T)
    Hn = syn.controller_range(rng.standard_normal((B, T, 65), dtype=np.float32))
    u = rng.random((B, T, 128), dtype=np.float32) if injected else None
    seed, offset = 4242, (3 << 32) + 9
    kw = dict(uniform=dev(u)) if injected else dict(seed=seed, offset=offset)
    ref = oracle.noise_forward(Hn, u, 128, seed=seed, offset=offset)
    y = ddsp.noise_forward(dev(Hn), 128, **kw)
    assert np.max(np.abs(y.cpu().numpy() - ref)) <= 2e-6
    base = torch.randn(B, T * 128, device="cuda")
    acc = ddsp.noise_forward(dev(Hn), 128, out=base.clone(), accumulate=True, **kw)
    assert float((acc - (base + y)).abs().max()) <= 1e-6
    L = ddsp._lib.lib()
    L.ddsp_noise_set_generic(8)
    try:
        old = ddsp.noise_forward(dev(Hn), 128, **kw)
    finally:
        L.ddsp_noise_set_generic(0)
    assert float((old - y).abs().max()) <= 2e-6
    assert torch.equal(y, ddsp.noise_forward(dev(Hn), 128, **kw))            # deterministic


@pytest.mark.parametrize("hop,nf", [(512, 257), (512, 195), (256, 129), (256, 100)])
def test_noise_fft_form_equals_direct_form(hop, nf):
    """The in-LDS FFT form against the direct (time-domain) kernels on the same inputs: the in-kernel Philox draw uses the
    same counter layout in both (identical noise), so the outputs agree to FFT rounding; accumulate and odd frame counts too."""
    L = ddsp._lib.lib()
    rng = np.random.default_rng(hop + nf)
    B, T = 3, 21                                                   # 63 frames: the last pair is half empty
    H = dev(syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32)))
    base = torch.randn(B, T * hop, device="cuda")
    L.ddsp_noise_set_generic(4)                                    # FFT form, also at hop 256
    try:
        got = ddsp.noise_forward(H, hop, seed=77, offset=12345)
        acc = ddsp.noise_forward(H, hop, seed=77, offset=12345, out=base.clone(), accumulate=True)
        again = ddsp.noise_forward(H, hop, seed=77, offset=12345)
    finally:
        L.ddsp_noise_set_generic(0)
    L.ddsp_noise_set_generic(2)
    try:
        ref = ddsp.noise_forward(H, hop, seed=77, offset=12345)
    finally:
        L.ddsp_noise_set_generic(0)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= 2e-6 * scale
    assert float((acc - (base + ref)).abs().max()) <= 2e-6 * scale + 1e-6
    assert torch.equal(got, again)                                                    # deterministic


# ---- autograd (boundary contract: differentiable w.r.t. c, a, H; train/train.py:33-34) ------------------
def test_osc_grad_matches_reference_autograd():
    g = load_golden("g10_osc_grad")
    osc = ddsp.OscillatorBank(Conf(8, 16000, 64)).cuda()
    c = dev(g["c"]).requires_grad_()
    a = dev(g["a"]).requires_grad_()
    y = osc({"f0": dev(g["f0"]), "c": c, "a": a})
    assert np.max(np.abs(y.detach().cpu().numpy() - g["y"])) <= TOL_Y
    (y * dev(g["g"])).sum().backward()
    gc, ga = c.grad.cpu().numpy(), a.grad.cpu().numpy()
    assert np.max(np.abs(gc - g["grad_c"])) <= 1e-5 * max(1.0, np.max(np.abs(g["grad_c"])))
    assert np.max(np.abs(ga - g["grad_a"])) <= 1e-5 * max(1.0, np.max(np.abs(g["grad_a"])))


@pytest.mark.parametrize("name", ["g10_noise_grad_hop128", "g10_noise_grad_hop64", "g17_noise_grad_hop512_f257", "g17_noise_grad_hop512_f195"])
@pytest.mark.parametrize("generic", [0, 1, 2])   # 0: default kernels (hop 512: in-LDS FFT form); 1: one frame per workgroup; 2: batched direct form
def test_noise_grad_matches_reference_autograd(name, generic):
    g = load_golden(name)
    hop = int(g["hop"])
    L = ddsp._lib.lib()
    L.ddsp_noise_set_generic(generic)
    try:
        H = dev(g["H"]).requires_grad_()
        y = ddsp.FilteredNoise(Conf(1, 16000, hop))({"H": H}, noise=dev(g["uniform"]))
        (y * dev(g["g"])).sum().backward()
    finally:
        L.ddsp_noise_set_generic(0)
    assert np.max(np.abs(y.detach().cpu().numpy() - g["y"])) <= 2e-6
    ref = g["grad_H"]
    assert np.max(np.abs(H.grad.cpu().numpy() - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("nf", [257, 195])
@pytest.mark.parametrize("injected", [False, True])
def test_noise_fft_backward_equals_direct_backward(nf, injected):
    """hop 512, 257 bands: the in-LDS FFT form of the backward (correlation as conj(X) G; dH from one packed 512-point
    transform) against the direct (time-domain) backward on the same draw -- in-kernel Philox with a 64-bit offset, or
    injected; 21 frames: the last frame pair is half empty.  (195 bands take the direct kernels either way: same results.)"""
    rng = np.random.default_rng(900 + nf)
    B, T, hop = 3, 7, 512
    gy = dev(rng.standard_normal((B, T * hop)).astype(np.float32))
    u = dev(rng.random((B, T, hop), dtype=np.float32)) if injected else None
    kw = dict(uniform=u) if injected else dict(seed=31337, offset=(9 << 32) + 5)
    L = ddsp._lib.lib()
    got = ddsp.noise_backward(gy, hop, nf, **kw)
    again = ddsp.noise_backward(gy, hop, nf, **kw)
    L.ddsp_noise_set_generic(2)
    try:
        ref = ddsp.noise_backward(gy, hop, nf, **kw)
    finally:
        L.ddsp_noise_set_generic(0)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= 1e-5 * scale
    assert torch.equal(got, again)


@pytest.mark.parametrize("hop,nf", [(128, 65), (512, 257), (512, 195), (256, 129)])
def test_noise_error_is_relative_to_each_frames_own_level(hop, nf):
    """The reference transforms every frame alone (filtered_noise.py:7-32), so a quiet frame next to a loud one keeps its own
    relative accuracy.  The hop-512 form transforms two frames as one complex sequence: it equalises them by powers of two first
    (csrc/ddsp_noise_fft.hip: frame_scale).  Frame levels spread over seven decades, one exactly-zero frame, an odd number
    of frames per row; error per frame against the C oracle, relative to the frame's own peak."""
    rng = np.random.default_rng(4100 + hop + nf)
    B, T = 3, 21
    level = (10.0 ** rng.uniform(-4, 3, size=(B, T, 1))).astype(np.float32)
    Hm = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32)) * level
    Hm[1, 4] = 0.0
    u = rng.random((B, T, hop), dtype=np.float32)
    ref = oracle.noise_forward(Hm, u, hop).reshape(B, T, hop)
    got = ddsp.noise_forward(dev(Hm), hop, uniform=dev(u)).cpu().numpy().reshape(B, T, hop)
    peak = np.abs(ref).max(axis=2)
    err = np.abs(got - ref).max(axis=2)
    assert np.all(got[1, 4] == 0.0)
    assert np.all(err <= 2e-6 * np.maximum(peak, level[:, :, 0])), float(np.max(err / np.maximum(peak, level[:, :, 0])))


def test_noise_fft_backward_error_is_relative_to_each_rows_own_level():
    """As above for the hop-512 backward: upstream gradient rows of very different size share a transform pair; each frame's
    dH against the direct (per-frame) backward, relative to that frame's own largest gradient."""
    rng = np.random.default_rng(4200)
    B, T, hop, nf = 3, 7, 512, 257
    level = (10.0 ** rng.uniform(-4, 3, size=(B, T, 1))).astype(np.float32)
    gy = (rng.standard_normal((B, T, hop)).astype(np.float32) * level).reshape(B, T * hop)
    gy[2, 3 * hop:4 * hop] = 0.0
    kw = dict(seed=77, offset=(3 << 32) + 1)
    L = ddsp._lib.lib()
    got = ddsp.noise_backward(dev(gy), hop, nf, **kw).cpu().numpy()
    L.ddsp_noise_set_generic(2)
    try:
        ref = ddsp.noise_backward(dev(gy), hop, nf, **kw).cpu().numpy()
    finally:
        L.ddsp_noise_set_generic(0)
    peak = np.abs(ref).max(axis=2)
    err = np.abs(got - ref).max(axis=2)
    assert np.all(got[2, 3] == 0.0)
    assert np.all(err <= 1e-5 * np.maximum(peak, 1e-30)), float(np.max(err / np.maximum(peak, 1e-30)))


def test_osc_grad_vs_torch_restatement_bigger():
    # autograd of the torch-op restatement (CPU, same ops as the reference) on a cfg2-shaped slice incl. masked harmonics
    from oracle import torch_restatement as tr
    shape = syn.SynthShape("g", 2, 16000, 128, 40, 100, 65)
    ctl = syn.make_controls(shape, 55, "musical")
    rng = np.random.default_rng(56)
    gy = rng.standard_normal((2, 40 * 128)).astype(np.float32)
    c_ref = torch.from_numpy(ctl["c"]).requires_grad_()
    a_ref = torch.from_numpy(ctl["a"]).requires_grad_()
    y_ref = tr.oscillator_bank(torch.from_numpy(ctl["f0"]), c_ref, a_ref, 128, 16000)
    (y_ref * torch.from_numpy(gy)).sum().backward()
    c = dev(ctl["c"]).requires_grad_()
    a = dev(ctl["a"]).requires_grad_()
    y = ddsp.OscillatorBank(Conf(100, 16000, 128)).cuda()({"f0": dev(ctl["f0"]), "c": c, "a": a})
    (y * dev(gy)).sum().backward()
    for got, ref in ((c.grad, c_ref.grad), (a.grad, a_ref.grad)):
        ref = ref.numpy()
        assert np.max(np.abs(got.cpu().numpy() - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))


def test_noise_grad_device_rng_consistent():
    # device RNG: backward regenerates the same Philox draw as the forward (finite-difference check on one bin)
    shape = syn.SynthShape("g", 1, 16000, 128, 4, 10, 65)
    Hn = syn.make_controls(shape, 9)["H"]
    fn = ddsp.FilteredNoise(Conf(1, 16000, 128), rng="device", seed=42)
    H = dev(Hn).requires_grad_()
    y = fn({"H": H})
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    eps = 1e-2
    Hp = Hn.copy(); Hp[0, 2, 7] += eps
    fn2 = ddsp.FilteredNoise(Conf(1, 16000, 128), rng="device", seed=42)
    y0 = fn2({"H": dev(Hn)}); fn2._offset = 0
    y1 = fn2({"H": dev(Hp)})
    fd = float(((y1 - y0) * gy).sum()) / eps
    assert abs(fd - float(H.grad[0, 2, 7])) <= 2e-3 * max(1.0, abs(fd))


# ---- edge shapes ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,H,hop,sr", [(1, 1, 1, 1, 8000), (2, 3, 1, 2, 8000), (1, 2, 5, 1, 16000), (3, 9, 27, 48, 22050),
                                          (1, 4, 180, 512, 44100), (2, 5, 300, 32, 48000), (1, 3, 1600, 8, 48000),
                                          (7, 33, 64, 96, 16000), (1, 700, 3, 16, 16000)])
def test_osc_edge_shapes_vs_oracle(B, T, H, hop, sr):
    rng = np.random.default_rng(B * 1000 + T * 10 + H)
    f0 = rng.uniform(5, max(10.0, 0.6 * sr / max(H, 2)), (B, T, 1)).astype(np.float32)
    c = syn.controller_range(rng.standard_normal((B, T, H), dtype=np.float32))
    a = syn.controller_range(rng.standard_normal((B, T, 1), dtype=np.float32))
    ref, d = oracle.osc_forward(f0, c, a, hop, sr, debug=True)
    y, _, phi = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr, debug_phases=True)
    assert np.array_equal(bits(phi.cpu().numpy()), bits(d["phi"]))
    assert np.max(np.abs(y.cpu().numpy() - ref)) <= TOL_Y
    y2, _, _ = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr)
    assert np.max(np.abs(y2.cpu().numpy() - ref)) <= TOL_Y


def test_shape_limits_are_reported_not_computed():
    L = ddsp._lib.lib()
    x = torch.zeros(4, device="cuda")
    p = x.data_ptr()
    # 1601 harmonics: no tiling; T*hop >= 2^24: sample indices not exact in fp32
    assert L.ddsp_osc_forward(p, p, p, p, p, None, None, None, 1, 1, 1601, 1, 16000, None) == -2
    assert L.ddsp_osc_forward(p, p, p, p, p, None, None, None, 1, 1 << 14, 1, 1 << 10, 16000, None) == -2
    assert L.ddsp_noise_forward(p, None, p, 1, 1, 1, 8, 0, 0, 0, None) == -1          # F < 2
    with pytest.raises(ValueError):
        ddsp.noise_forward(torch.zeros(1, 2, 5, device="cuda"), 8, uniform=torch.zeros(1, 2, 7, device="cuda"))
    empty = ddsp.osc_forward(torch.zeros(0, 3, 1, device="cuda"), torch.zeros(0, 3, 4, device="cuda"),
                             torch.zeros(0, 3, 1, device="cuda"), 8, 16000)[0]
    assert empty.shape == (0, 24)


@pytest.mark.parametrize("nf,hop", [(2, 8), (3, 8), (2, 5), (9, 16), (130, 8), (129, 8192)])   # last: generic kernel, > 64 KiB of LDS
def test_noise_edge_shapes_vs_oracle(nf, hop):
    rng = np.random.default_rng(nf * 100 + hop)
    H = syn.controller_range(rng.standard_normal((2, 5, nf), dtype=np.float32))
    u = rng.random((2, 5, hop), dtype=np.float32)
    ref = oracle.noise_forward(H, u, hop)
    y = ddsp.noise_forward(dev(H), hop, uniform=dev(u))
    assert np.max(np.abs(y.cpu().numpy() - ref)) <= 2e-6


def test_launch_from_worker_thread_on_side_stream():
    # rt/synth.py calls forward_live from the JACK callback thread: launches must work off the main thread
    # and follow the caller's current stream
    import threading
    g = load_golden("g3_osc_cfg2_musical")
    x = {k: dev(g[k]) for k in ("f0", "c", "a")}
    out = {}

    def worker():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            y, _, _ = ddsp.osc_forward(x["f0"], x["c"], x["a"], int(g["hop"]), int(g["sample_rate"]))
            out["y"] = y.cpu()

    th = threading.Thread(target=worker)
    th.start()
    th.join()
    assert np.max(np.abs(out["y"].numpy() - g["y"])) <= TOL_Y


def test_osc_long_clip_extreme_f0_phases_bit_exact():
    # 16 s clip, f0 in [0.01, 4000] Hz (the double sums stay exact, any scan order gives the same bits)
    rng = np.random.default_rng(77)
    B, T, H, hop, sr = 2, 2000, 30, 128, 16000
    f0 = np.exp(rng.uniform(np.log(0.01), np.log(4000.0), (B, T, 1))).astype(np.float32)
    c = syn.controller_range(rng.standard_normal((B, T, H), dtype=np.float32))
    a = syn.controller_range(rng.standard_normal((B, T, 1), dtype=np.float32))
    ref, d = oracle.osc_forward(f0, c, a, hop, sr, debug=True)
    y, _, phi = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr, debug_phases=True)
    assert np.array_equal(bits(phi.cpu().numpy()), bits(d["phi"]))
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(y.cpu().numpy()), ok)
    y2, _, _ = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr)
    assert np.max(np.abs(y2.cpu().numpy()[ok] - ref[ok])) <= TOL_Y


@pytest.mark.parametrize("hop,nf,B,T", [(128, 65, 2, 37), (256, 129, 1, 33), (512, 257, 1, 9), (64, 65, 3, 20), (24, 7, 2, 5)])
def test_noise_grad_batched_equals_generic_and_restatement(hop, nf, B, T):
    # the batched backward (all tile sizes) against the one-frame-per-workgroup kernel and torch autograd of the restatement
    from oracle import torch_restatement as tr
    rng = np.random.default_rng(hop * 7 + nf)
    Hn = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32))
    u = rng.random((B, T, hop), dtype=np.float32)
    gy = rng.standard_normal((B, T * hop)).astype(np.float32)
    Hr = torch.from_numpy(Hn).requires_grad_()
    (tr.filtered_noise(Hr, hop, uniform=torch.from_numpy(u)) * torch.from_numpy(gy)).sum().backward()
    ref = Hr.grad.numpy()
    L = ddsp._lib.lib()
    got = {}
    for generic in (0, 1):
        L.ddsp_noise_set_generic(generic)
        try:
            got[generic] = ddsp.noise_backward(dev(gy), hop, nf, uniform=dev(u)).cpu().numpy()
        finally:
            L.ddsp_noise_set_generic(0)
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert np.max(np.abs(got[0] - ref)) <= 2e-5 * scale
    assert np.max(np.abs(got[1] - ref)) <= 2e-5 * scale


@pytest.mark.parametrize("seed", range(10))
def test_osc_random_shapes_vs_oracle(seed):
    # random shapes / hops / rates, f0 up to 0.7*sr (all-masked NaN frames occur), one zero-f0 frame every other seed
    rng = np.random.default_rng(7000 + seed)
    B, T = int(rng.integers(1, 5)), int(rng.integers(1, 60))
    H = int(rng.integers(1, 130))
    hop = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 100, 128, 256]))
    sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    f0 = np.exp(rng.uniform(np.log(5.0), np.log(0.7 * sr), (B, T, 1))).astype(np.float32)
    if seed % 2 == 0:
        f0[0, rng.integers(0, T), 0] = 0.0
    c = rng.uniform(0.0, 2.0, (B, T, H)).astype(np.float32)
    a = rng.uniform(0.0, 2.0, (B, T, 1)).astype(np.float32)
    ref, d = oracle.osc_forward(f0, c, a, hop, sr, debug=True)
    y, _, phi = ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr, debug_phases=True)
    assert np.array_equal(bits(phi.cpu().numpy()), bits(d["phi"]))
    ok = np.isfinite(ref)
    for out in (y, ddsp.osc_forward(dev(f0), dev(c), dev(a), hop, sr)[0]):
        out = out.cpu().numpy()
        assert np.array_equal(np.isfinite(out), ok)
        if ok.any():
            assert np.max(np.abs(out[ok] - ref[ok])) <= TOL_Y
