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
