"""Pins the CPU oracle (oracle/ddsp_oracle.c) against fixtures captured from the reference.

Tolerances: inc / cum / phi bit-exact; y <= 1e-6; noise <= 2e-6.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import oracle

OSC_INTERMEDIATE = ["g1_osc_tiny", "g5b_osc_nyquist_finite", "g6_osc_hop100", "g6_osc_hop441", "g6_osc_hop3",
                    "g6_osc_hop7", "g6_osc_hop160", "g6_osc_hop480", "g6_osc_single_frame"]


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", OSC_INTERMEDIATE)
def test_osc_phase_path_bit_exact(name):
    g = load_golden(name)
    y, d = oracle.osc_forward(g["f0"], g["c"], g["a"], int(g["hop"]), int(g["sample_rate"]), debug=True)
    assert np.array_equal(bits(d["inc"]), bits(g["inc"])), "upsampled increments differ"
    assert np.array_equal(bits(d["cum"]), bits(g["cum"])), "fl32(double cumsum) differs"
    assert np.array_equal(bits(d["phi"]), bits(g["phi"])), "phases differ"
    assert np.max(np.abs(y - g["y"])) <= 1e-6


@pytest.mark.parametrize("name", ["g2_osc_cfg2_live", "g3_osc_cfg2_musical", "g4_osc_cfg3_1s"])
def test_osc_long_clips(name):
    g = load_golden(name)
    y, d = oracle.osc_forward(g["f0"], g["c"], g["a"], int(g["hop"]), int(g["sample_rate"]), debug=True)
    assert np.array_equal(bits(d["phi"][:, g["phi_idx"], :]), bits(g["phi_sub"]))
    assert np.max(np.abs(y - g["y"])) <= 1e-6


def test_osc_cfg1():
    g = load_golden("g2b_osc_cfg1")
    y = oracle.osc_forward(g["f0"], g["c"], g["a"], int(g["hop"]), int(g["sample_rate"]))
    assert np.max(np.abs(y - g["y"])) <= 1e-6


def test_osc_frames_g1():
    g = load_golden("g1_osc_tiny")
    w, amp = oracle.osc_frames(g["f0"], g["c"], int(g["sample_rate"]))
    np.testing.assert_allclose(amp, g["amp_frame"], rtol=4e-7, atol=0)


def test_osc_nyquist_nan_frame():
    g = load_golden("g5_osc_nyquist")
    y, d = oracle.osc_forward(g["f0"], g["c"], g["a"], int(g["hop"]), int(g["sample_rate"]), debug=True)
    assert np.array_equal(bits(d["phi"]), bits(g["phi"]))
    assert np.array_equal(np.isnan(y), np.isnan(g["y"])) and np.isnan(y).any()
    ok = ~np.isnan(y)
    assert np.max(np.abs(y[ok] - g["y"][ok])) <= 1e-6


@pytest.mark.parametrize("name,calls", [("g7_osc_live", 3), ("g7b_osc_live_batch2", 2)])
def test_osc_live_state(name, calls):
    g = load_golden(name)
    state = np.zeros(g["c_0"].shape[-1], np.float32)     # int64 zeros in the reference before call 1
    for k in range(calls):
        y = oracle.osc_forward(g[f"f0_{k}"], g[f"c_{k}"], g[f"a_{k}"], int(g["hop"]), int(g["sample_rate"]),
                               live_phase=state)
        assert np.array_equal(bits(state), bits(g[f"last_phases_{k}"]))
        assert np.max(np.abs(y - g[f"y_{k}"])) <= 1e-6


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g8_noise_*.npz"))))
def test_noise(path):
    g = load_golden(os.path.basename(path)[:-4])
    y, ir = oracle.noise_forward(g["H"], g["uniform"], int(g["hop"]), debug=True)
    assert np.max(np.abs(ir - g["ir"])) <= 5e-7
    assert np.max(np.abs(y - g["y"])) <= 2e-6


# ---- the torch-op restatement (what bench.py times as cpu_baseline) ---------------------------------
import torch  # noqa: E402

from oracle import torch_restatement as tr  # noqa: E402


@pytest.mark.parametrize("name", OSC_INTERMEDIATE + ["g2b_osc_cfg1", "g3_osc_cfg2_musical"])
def test_torch_restatement_osc_bitwise(name):
    g = load_golden(name)
    y, ph = tr.oscillator_bank(torch.from_numpy(g["f0"]), torch.from_numpy(g["c"]), torch.from_numpy(g["a"]),
                               int(g["hop"]), int(g["sample_rate"]), return_phases=True)
    assert np.array_equal(bits(y.numpy()), bits(g["y"]))
    if "phi" in g:
        assert np.array_equal(bits(ph.numpy()), bits(g["phi"]))


def test_torch_restatement_live_bitwise():
    g = load_golden("g7_osc_live")
    state = torch.zeros(180, dtype=torch.float32)
    for k in range(3):
        y = tr.oscillator_bank(torch.from_numpy(g[f"f0_{k}"]), torch.from_numpy(g[f"c_{k}"]), torch.from_numpy(g[f"a_{k}"]),
                               512, 44100, live_phase=state)
        assert np.array_equal(bits(y.numpy()), bits(g[f"y_{k}"]))
        assert np.array_equal(bits(state.numpy()), bits(g[f"last_phases_{k}"]))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g8_noise_*.npz"))))
def test_torch_restatement_noise_bitwise(path):
    g = load_golden(os.path.basename(path)[:-4])
    y = tr.filtered_noise(torch.from_numpy(g["H"]), int(g["hop"]), uniform=torch.from_numpy(g["uniform"]))
    assert np.array_equal(bits(y.numpy()), bits(g["y"]))
    torch.manual_seed(int(g["seed"]))
    y2 = tr.filtered_noise(torch.from_numpy(g["H"]), int(g["hop"]))
    assert np.array_equal(bits(y2.numpy()), bits(g["y"]))


# ---- C oracle vs torch-op restatement on random shapes (beyond the committed fixtures) ---------------------
@pytest.mark.parametrize("seed", range(12))
def test_c_oracle_equals_torch_restatement_random(seed):
    rng = np.random.default_rng(9000 + seed)
    B, T = int(rng.integers(1, 4)), int(rng.integers(1, 40))
    H = int(rng.integers(1, 40))
    hop = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 100, 128, 257]))
    sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    f0 = np.exp(rng.uniform(np.log(5.0), np.log(0.7 * sr), (B, T, 1))).astype(np.float32)
    if seed % 4 == 0:
        f0[0, rng.integers(0, T), 0] = 0.0
    c = rng.uniform(0.0, 2.0, (B, T, H)).astype(np.float32)
    a = rng.uniform(0.0, 2.0, (B, T, 1)).astype(np.float32)
    y, d = oracle.osc_forward(f0, c, a, hop, sr, debug=True)
    yt, pt = tr.oscillator_bank(torch.from_numpy(f0), torch.from_numpy(c), torch.from_numpy(a), hop, sr, return_phases=True)
    assert np.array_equal(bits(d["phi"]), bits(pt.numpy()))
    ok = np.isfinite(yt.numpy())
    assert np.array_equal(np.isfinite(y), ok)
    if ok.any():
        assert np.max(np.abs(y[ok] - yt.numpy()[ok])) <= 2e-6


@pytest.mark.parametrize("seed", range(8))
def test_c_noise_oracle_equals_torch_restatement_random(seed):
    rng = np.random.default_rng(9100 + seed)
    B, T = int(rng.integers(1, 3)), int(rng.integers(1, 9))
    nf = int(rng.integers(2, 80))
    hop = int(rng.choice([4, 8, 24, 64, 100, 128, 160, 256]))
    Hm = rng.uniform(0.0, 2.0, (B, T, nf)).astype(np.float32)
    u = rng.random((B, T, hop), dtype=np.float32)
    y = oracle.noise_forward(Hm, u, hop)
    yt = tr.filtered_noise(torch.from_numpy(Hm), hop, uniform=torch.from_numpy(u)).numpy()
    assert np.max(np.abs(y - yt)) <= 3e-6 * max(1.0, float(np.max(np.abs(yt))))


# ---- the in-kernel noise stream (FilteredNoise(rng='device')): Philox4x32-10 ---------------------------------------
# Known-answer vectors published with the Random123 library (kat_vectors, "philox4x32 10" lines): counter, key -> output.
PHILOX_KAT = [
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KAT)
def test_philox_known_answers(ctr, key, want):
    assert oracle.philox4x32_10(ctr, key) == want


def test_philox_uniform_layout():
    """counter = offset + frame * ceil(hop/4) + quad, key = seed (64-bit, low word first), u = (word >> 8) * 2^-24."""
    # seed 0 / offset 0: the first four samples are the first KAT vector's words
    u = oracle.philox_uniform(0, 0, 1, 2, 8)
    want = np.array([w >> 8 for w in PHILOX_KAT[0][2]], np.float64) / 2.0 ** 24
    assert np.array_equal(u[0, 0, :4].astype(np.float64), want)
    # a ragged hop (6 -> two counters per frame, the second half used), a 64-bit offset that carries into the high word
    seed, off, hop = 0x0123456789ABCDEF, (1 << 32) - 3, 6
    u = oracle.philox_uniform(seed, off, 2, 3, hop)
    for f in range(6):
        for m in range(hop):
            c = off + f * 2 + (m >> 2)
            w = oracle.philox4x32_10([c & 0xFFFFFFFF, c >> 32, 0, 0], [seed & 0xFFFFFFFF, seed >> 32])[m & 3]
            assert float(u.reshape(6, hop)[f, m]) == (w >> 8) / 2.0 ** 24
    assert u.min() >= 0.0 and u.max() < 1.0


def test_noise_oracle_regenerates_the_device_draw():
    rng = np.random.default_rng(4)
    H = rng.uniform(0.0, 2.0, (2, 3, 9)).astype(np.float32)
    u = oracle.philox_uniform(11, 5, 2, 3, 16)
    assert np.array_equal(oracle.noise_forward(H, None, 16, seed=11, offset=5), oracle.noise_forward(H, u, 16))


@pytest.mark.parametrize("name", ["g10_noise_grad_hop128", "g10_noise_grad_hop64", "g17_noise_grad_hop512_f257", "g17_noise_grad_hop512_f195"])
def test_torch_restatement_noise_gradient_equals_reference_autograd(name):
    """The restatement's autograd d/dH (what several GPU tests use as their reference) against the reference's own, fixtures G10 / G17."""
    import torch
    from oracle import torch_restatement as tr
    g = load_golden(name)
    H = torch.from_numpy(g["H"]).requires_grad_()
    y = tr.filtered_noise(H, int(g["hop"]), uniform=torch.from_numpy(g["uniform"]))
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert np.array_equal(y.detach().numpy(), g["y"])
    assert np.max(np.abs(H.grad.numpy() - g["grad_H"])) <= 1e-6 * max(1.0, float(np.max(np.abs(g["grad_H"]))))
