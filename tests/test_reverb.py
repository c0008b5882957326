"""Reverb against the G9 fixtures captured from the reference.  Device-agnostic module:
the CPU run here pins the restated arithmetic; the GPU run (marker gpu) pins the device FFT path."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import ddsp_pytorch_amd as ddsp


class Conf:
    def __init__(self, sample_rate):
        self.n_harmonics, self.sample_rate, self.hop_length = 1, sample_rate, 64


def make(g, device):
    rv = ddsp.Reverb(Conf(int(g["sample_rate"])))
    with torch.no_grad():
        rv.noise.copy_(torch.from_numpy(g["noise"]))
        rv.decay.copy_(torch.from_numpy(g["decay"]))
        rv.wet.copy_(torch.from_numpy(g["wet"]))
    return rv.to(device)


def check_all(device, tol):
    g = load_golden("g9_fft_convolve")
    y = ddsp.causal_fft_convolve(torch.from_numpy(g["signal"]).to(device), torch.from_numpy(g["kernel"]).to(device))
    scale = np.max(np.abs(g["y"]))
    assert np.max(np.abs(y.cpu().numpy() - g["y"])) <= tol * scale
    for clip in (4096, 1024):
        g = load_golden(f"g9_reverb_clip{clip}")
        rv = make(g, device)
        assert set(rv.state_dict()) == {"noise", "decay", "wet", "t", "buffer"}
        assert np.max(np.abs(rv.build_impulse().detach().cpu().numpy() - g["impulse"])) <= 1e-6
        y = rv(torch.from_numpy(g["x"]).to(device))
        assert np.max(np.abs(y.detach().cpu().numpy() - g["y"])) <= tol * max(1.0, np.max(np.abs(g["y"])))
    g = load_golden("g9_reverb_live")
    rv = make(g, device)
    for k in range(3):
        y = rv.live_forward(torch.from_numpy(g[f"x_{k}"]).to(device))
        assert np.max(np.abs(y.detach().cpu().numpy() - g[f"y_{k}"])) <= tol * max(1.0, np.max(np.abs(g[f"y_{k}"])))
        assert np.array_equal(rv.buffer.detach().cpu().numpy(), g[f"buffer_{k}"])


def test_reverb_cpu():
    check_all("cpu", 2e-6)


@pytest.mark.gpu
def test_reverb_gpu():
    check_all("cuda", 5e-6)


def test_reverb_is_differentiable():
    rv = ddsp.Reverb(Conf(512), initial_wet=0.2)
    x = torch.randn(2, 700, requires_grad=True)
    rv(x).square().sum().backward()
    assert x.grad is not None and rv.noise.grad is not None and rv.decay.grad is not None and rv.wet.grad is not None


def _torch_reverb(rv_cpu, x, live=False):
    """The module's own torch-op formulation on the CPU (pinned against the reference by test_reverb_cpu)."""
    return rv_cpu.live_forward(x) if live else rv_cpu(x)


@pytest.mark.gpu
def test_reverb_hip_impulse_matches_reference_fixture():
    for clip in (4096, 1024):
        g = load_golden(f"g9_reverb_clip{clip}")
        rv = make(g, "cuda")
        with torch.no_grad():
            imp = rv.build_impulse()                               # no grad: ddsp_reverb_impulse
        assert imp.shape == (1, int(g["sample_rate"]))
        assert np.max(np.abs(imp.cpu().numpy() - g["impulse"])) <= 1e-6
        # padded / cropped variants written directly by the kernel (reverb.py:34)
        for n_out in (100, 2048, 5000):
            got = ddsp.reverb.reverb_impulse(rv.noise.detach(), rv.decay.detach(), rv.wet.detach(), rv.t.detach().reshape(-1), n_out)
            ref = np.zeros(n_out, np.float32)
            m = min(n_out, 2048)
            ref[:m] = g["impulse"][0, :m]
            assert np.max(np.abs(got.cpu().numpy() - ref)) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n,rows", [(512, 700, 2), (512, 300, 3), (2048, 2048, 1), (16000, 4096, 2)])
def test_reverb_hip_forward_and_gradients_match_torch_formulation(sr, n, rows):
    """HIP path (impulse kernel, spectral product, fused backward, impulse backward) on the GPU against the torch-op
    formulation differentiated by autograd on the CPU -- padded (n > sr), cropped (n < sr) and equal lengths."""
    torch.manual_seed(sr + n)
    ref = ddsp.Reverb(Conf(sr), initial_wet=0.3, initial_decay=2.0)
    dev = ddsp.Reverb(Conf(sr))
    dev.load_state_dict(ref.state_dict())
    dev = dev.cuda()
    x0 = torch.randn(rows, n)
    w = torch.randn(rows, n)
    xr = x0.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * w).sum().backward()
    xg = x0.clone().cuda().requires_grad_(True)
    yg = dev(xg)
    (yg * w.cuda()).sum().backward()
    scale = max(1.0, float(yr.detach().abs().max()))
    assert float((yg.detach().cpu() - yr.detach()).abs().max()) <= 5e-6 * scale
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
    for name in ("noise", "decay", "wet"):
        gr, gg = getattr(ref, name).grad, getattr(dev, name).grad.cpu()
        assert gg.shape == gr.shape
        assert float((gg - gr).abs().max()) <= 2e-5 * max(1.0, float(gr.abs().max())), name
    with torch.no_grad():                                          # inference: nothing saved
        assert float((dev(x0.cuda()).cpu() - yr.detach()).abs().max()) <= 5e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("sr,n", [(44100, 2048), (16000, 512), (2048, 2048), (1000, 77)])
def test_reverb_hip_live_matches_torch_formulation(sr, n):
    """live_forward as a direct convolution of the last n outputs (no FFT) against the FFT formulation on the CPU:
    four consecutive callbacks, audio and carried history (the history is bit-exact: it is only moved)."""
    torch.manual_seed(sr)
    ref = ddsp.Reverb(Conf(sr), initial_wet=0.5, initial_decay=3.0)
    dev = ddsp.Reverb(Conf(sr))
    dev.load_state_dict(ref.state_dict())
    dev = dev.cuda()
    address = dev.buffer.data_ptr()
    with torch.no_grad():
        for call in range(4):
            x = torch.randn(1, n)
            y_ref = ref.live_forward(x)
            y = dev.live_forward(x.cuda())
            assert y.shape == (1, n)
            assert float((y.cpu() - y_ref).abs().max()) <= 5e-6 * max(1.0, float(y_ref.abs().max())), call
            assert torch.equal(dev.buffer.data.cpu(), ref.buffer.data)
    assert dev.buffer.data_ptr() == address                        # static address: the callback can live in a hipGraph
    with pytest.raises(ddsp._lib.DdspHipError):
        dev.live_forward(torch.zeros(1, sr + 1, device="cuda"))    # longer than the history: outside the reference's semantics


def _check_g16(device, tol_y, tol_g):
    """Fixtures G16 (round 2), straight from the reference: three live callbacks against a 16 000-sample history, and the
    reference's own autograd of `forward` for a padded and a cropped clip."""
    g = load_golden("g16_reverb_live_16k")
    rv = make(g, device)
    with torch.no_grad():
        for k in range(3):
            y = rv.live_forward(torch.from_numpy(g[f"x_{k}"]).to(device))
            assert np.max(np.abs(y.cpu().numpy() - g[f"y_{k}"])) <= tol_y * max(1.0, np.max(np.abs(g[f"y_{k}"]))), k
    assert np.array_equal(rv.buffer.detach().cpu().numpy(), g["buffer_last"])
    for clip in (3000, 1200):
        g = load_golden(f"g16_reverb_grad_clip{clip}")
        rv = make(g, device)
        x = torch.from_numpy(g["x"]).to(device).requires_grad_()
        y = rv(x)
        (y * torch.from_numpy(g["w"]).to(device)).sum().backward()
        assert np.max(np.abs(y.detach().cpu().numpy() - g["y"])) <= tol_y * max(1.0, np.max(np.abs(g["y"])))
        for name, got in (("grad_x", x.grad), ("grad_noise", rv.noise.grad), ("grad_decay", rv.decay.grad), ("grad_wet", rv.wet.grad)):
            ref = g[name]
            err = float(np.max(np.abs(got.detach().cpu().numpy() - ref)))
            assert err <= tol_g * max(1.0, float(np.max(np.abs(ref)))), (clip, name, err)


def test_reverb_reference_fixtures_round2_cpu():
    _check_g16("cpu", 2e-6, 2e-5)


@pytest.mark.gpu
def test_reverb_reference_fixtures_round2_gpu():
    _check_g16("cuda", 5e-6, 5e-5)


def test_transform_length_of_the_hip_path():
    """reverb.fft_length: the smallest 2^a * {1, 3, 5, 25, 125, 625} >= need, always even (rfft / irfft pair), and never
    shorter than the linear convolution needs (N + L - 1: no wrap-around into the first N outputs)."""
    from ddsp_pytorch_amd.reverb import fft_length
    assert fft_length(64000 + 16000 - 1) == 80000          # the training shape: 80 000 points, not the reference's 2N = 128 000
    for need in (1, 2, 3, 5, 7, 1023, 1024, 1025, 5119, 79999, 88199, 239999, 1 << 20):
        n = fft_length(need)
        assert n >= need and n % 2 == 0
        m = n
        while m % 2 == 0:
            m //= 2
        assert m in (1, 3, 5, 25, 125, 625)
        assert n < 2 * max(need, 2) + 2                    # never worse than the next power of two
