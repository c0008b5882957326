"""CPU-side checks: the C-ABI library loads and exports every symbol include/ddsp_hip.h declares
(no compute calls without a GPU), argument validation, host logic, state-dict compatibility."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Conf:
    def __init__(self, n_harmonics, sample_rate, hop_length):
        self.n_harmonics, self.sample_rate, self.hop_length = n_harmonics, sample_rate, hop_length


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ddsp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ddsp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    L = ctypes.CDLL(ddsp._lib.SO_PATH)
    syms = declared_symbols()
    assert set(syms) == set(ddsp._lib.EXPORTS)
    for s in syms:
        assert hasattr(L, s), s


def test_abi_version_and_scratch_size():
    L = ddsp._lib.lib()
    assert L.ddsp_hip_abi_version() == 4 == ddsp._lib.ABI_VERSION
    assert L.ddsp_osc_scratch_bytes(0, 1, 1) == 0
    n = 64 * 500 * 100
    assert L.ddsp_osc_scratch_bytes(64, 500, 100) >= 16 * n


def test_argument_validation_without_gpu():
    L = ddsp._lib.lib()
    assert L.ddsp_osc_forward(None, None, None, None, None, None, None, None, 1, 1, 1, 1, 16000, None) == -1
    assert L.ddsp_osc_forward(None, None, None, None, None, None, None, None, 0, 1, 1, 1, 16000, None) == 0  # empty batch
    assert L.ddsp_noise_forward(None, None, None, 1, 1, 65, 128, 0, 0, 0, None) == -1
    assert L.ddsp_osc_set_tiling(7) == -2 and L.ddsp_osc_set_tiling(0) == 0
    # the callers' entry points validate before touching the device as well
    assert L.ddsp_noise_forward_counter(None, None, 1, 1, 65, 128, 0, None, 0, None) == -1
    assert L.ddsp_gru_forward(None, None, None, None, None, None, None, None, None, 1, 1, 16, None) == -1
    assert L.ddsp_gru_forward(None, None, None, None, None, None, None, None, None, 0, 1, 16, None) == 0   # empty batch
    assert L.ddsp_gru_backward(None, None, None, None, None, None, None, None, None, None, None, 1, 1, 16, None) == -1
    assert L.ddsp_gru_scratch_bytes(4, 1024) == 0 and L.ddsp_gru_scratch_bytes(4, 512) > 0   # hidden sizes up to 512
    assert L.ddsp_gru_set_mode(7) == -2 and L.ddsp_gru_set_mode(0) == 0
    assert L.ddsp_spectral_loss(None, None, None, None, None, 10, 1.0, 1e-7, None) == -1
    assert L.ddsp_scaled_sigmoid_forward(None, None, 10, None) == -1 and L.ddsp_scaled_sigmoid_forward(None, None, 0, None) == 0
    assert L.ddsp_ln_lrelu_forward(None, None, None, None, None, None, 4, 512, 1e-5, 0.01, None) == -1
    assert L.ddsp_ln_lrelu_scratch_bytes(512) > 0
    assert L.ddsp_ln_lrelu_backward(None, None, None, None, None, None, None, None, None, None, None, 0, 512, 0.01, None) == -1   # empty rows still need dgamma/dbeta
    assert L.ddsp_gru_set_fault_step(-1) == -2 and L.ddsp_gru_set_fault_step(0) == 0
    # round-2 entry points: framing, one-kernel loss scale, column sums, reverb, counter-driven noise backward
    assert L.ddsp_stft_frames(None, None, None, 1, 4096, 512, 128, None) == -1 and L.ddsp_stft_frames(None, None, None, 0, 4096, 512, 128, None) == 0
    assert L.ddsp_stft_frames_backward(None, None, None, 1, 4096, 512, 128, 0, None) == -1
    assert L.ddsp_mss_scale_supported(512) == 1 and L.ddsp_mss_scale_supported(96) == 0 and L.ddsp_mss_scale_supported(4096) == 0
    assert L.ddsp_mss_scale_scratch_bytes() > 0
    assert L.ddsp_mss_scale(None, None, None, None, None, None, 1, 4096, 512, 128, 1.0, 1e-7, None) == -1     # no output word
    assert L.ddsp_colsum(None, None, None, 8, 0, 0, None) == 0 and L.ddsp_colsum(None, None, None, 8, 4, 0, None) == -1
    assert L.ddsp_colsum_scratch_bytes(512) > 0 and L.ddsp_colsum_scratch_bytes(0) == 0
    assert L.ddsp_noise_backward_counter(None, None, 1, 1, 65, 128, 0, None, None) == -1
    assert L.ddsp_reverb_impulse(None, None, None, None, None, 16, 16, None) == -1


def test_module_boundary_matches_reference_contract():
    osc = ddsp.OscillatorBank(Conf(60, 16000, 128))
    sd = osc.state_dict()
    assert list(sd) == ["harmonics", "last_phases"]
    assert sd["harmonics"].dtype == torch.int64 and torch.equal(sd["harmonics"], torch.arange(1, 61))
    assert sd["last_phases"].dtype == torch.int64 and not any(p.requires_grad for p in osc.parameters())
    assert (osc.n_harmonics, osc.sample_rate, osc.hop_size) == (60, 16000, 128)
    fn = ddsp.FilteredNoise(Conf(60, 16000, 128))
    assert fn.block_size == 128 and len(fn.state_dict()) == 0
    with pytest.raises(ddsp._lib.DdspHipError):
        osc({"f0": torch.ones(1, 2, 1), "c": torch.ones(1, 2, 60), "a": torch.ones(1, 2, 1)})  # CPU tensors: no fallback
    with pytest.raises(ValueError):
        osc({"f0": torch.ones(1, 2), "c": torch.ones(1, 2, 60), "a": torch.ones(1, 2, 1)})


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ddsp-pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "libddsp_oracle", "ddsp_oracle", "oracle/"):
                    assert needle not in text, (f, needle)


def test_synthetic_controls_ranges():
    ctl = syn.make_controls(syn.CFG2, 1002, "all_live", batch=2)
    assert ctl["f0"].shape == (2, 500, 1) and ctl["c"].shape == (2, 500, 100) and ctl["H"].shape == (2, 500, 65)
    assert ctl["f0"].max() * 100 < 8000 and ctl["f0"].min() >= 39
    assert all(v.dtype == np.float32 for v in ctl.values())
    assert ctl["c"].min() >= 1e-7 and ctl["c"].max() <= 2.0 + 1e-6
    mus = syn.make_controls(syn.CFG2, 1003, "musical", batch=2)["f0"]
    assert 31.0 < mus.min() and mus.max() < 2006.0


def test_hooks_are_refused_without_opt_in():
    """The process-global *_set_* hooks change every later launch: a process that did not set DDSP_TEST_HOOKS=1 before the
    library was loaded gets DDSP_EPERM (-3) and nothing changes; restoring the default (0) is always allowed."""
    import subprocess
    import sys
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); "
            "print(L.ddsp_test_hooks_enabled(), L.ddsp_osc_set_tiling(13), L.ddsp_osc_set_tiling(0), L.ddsp_noise_set_generic(1), "
            "L.ddsp_noise_set_generic(0), L.ddsp_gru_set_mode(2), L.ddsp_gru_set_mode(0), L.ddsp_gru_set_fault_step(3))")
    env = {k: v for k, v in os.environ.items() if k != "DDSP_TEST_HOOKS"}
    out = subprocess.run([sys.executable, "-c", code, ddsp._lib.SO_PATH], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["0", "-3", "0", "-3", "0", "-3", "0", "-3"], out
    env["DDSP_TEST_HOOKS"] = "1"
    out = subprocess.run([sys.executable, "-c", code, ddsp._lib.SO_PATH], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "0", "0", "0", "0", "0", "0", "0"], out
    assert ddsp._lib.lib().ddsp_test_hooks_enabled() == 1          # this process: tests/conftest.py opted in
