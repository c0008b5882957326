"""The reference's default noise shape (195 bands at hop 512, config/default.py:15,19) through ddsp_noise_forward_ws: impulse responses
of the whole batch as one split-bf16 matrix-core product (csrc/ddsp_noise_ir.hip) feeding the in-LDS FFT form, against the CPU oracle
(model/ddsp/filtered_noise.py:7-53) and against the cosine-sum path it replaces.  Tolerance 2e-6 (relative to max(1, |y|)) as for
every other noise form."""
import ctypes

import numpy as np
import pytest
import torch

import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import synthetic as syn
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL = 2e-6


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cosine_sums(fn):
    """Runs fn() with the matrix product switched off (ddsp_noise_set_generic bit 4)."""
    L = ddsp._lib.lib()
    assert L.ddsp_noise_set_generic(16) == 0
    try:
        return fn()
    finally:
        L.ddsp_noise_set_generic(0)


@pytest.mark.parametrize("B,T,nf", [(3, 1399, 195), (1, 4096, 195), (5, 821, 200), (2, 2051, 224), (4, 1027, 193)])
def test_matrix_product_form_vs_oracle_and_cosine_sums(B, T, nf):
    """>= 4 096 frames (where the forward takes the product), ending inside a 64-frame tile, a 16-frame operand and a frame pair; band
    counts across the range the product is built for."""
    rng = np.random.default_rng(B * 1000 + T + nf)
    Hn = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32))
    assert ddsp._lib.lib().ddsp_noise_workspace_bytes(B, T, nf, 512) > 0
    u = rng.random((B, T, 512), dtype=np.float32)
    y = ddsp.noise_forward(dev(Hn), 512, uniform=dev(u)).cpu().numpy()
    ref = oracle.noise_forward(Hn, u, 512)
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert np.max(np.abs(y - ref)) <= TOL * scale
    y_sums = cosine_sums(lambda: ddsp.noise_forward(dev(Hn), 512, uniform=dev(u))).cpu().numpy()
    assert np.max(np.abs(y - y_sums)) <= TOL * scale
    assert np.max(np.abs(y_sums - ref)) <= TOL * scale


def test_matrix_product_form_device_draw_and_accumulate():
    rng = np.random.default_rng(77)
    B, T, nf = 2, 2100, 195
    Hn = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32))
    seed, offset = 4321, (3 << 32) + 5
    ref = oracle.noise_forward(Hn, None, 512, seed=seed, offset=offset)
    y = ddsp.noise_forward(dev(Hn), 512, seed=seed, offset=offset).cpu().numpy()
    assert np.max(np.abs(y - ref)) <= TOL * max(1.0, float(np.max(np.abs(ref))))
    base = rng.standard_normal((B, T * 512)).astype(np.float32)
    both = ddsp.noise_forward(dev(Hn), 512, seed=seed, offset=offset, out=dev(base), accumulate=True).cpu().numpy()
    assert np.max(np.abs(both - (base + ref))) <= TOL * max(1.0, float(np.max(np.abs(ref)))) + 1e-6
    counter = torch.tensor([offset], dtype=torch.int64, device="cuda")
    y_c = ddsp.noise_forward(dev(Hn), 512, seed=seed, counter=counter).cpu().numpy()
    assert np.array_equal(y_c, y)


def test_matrix_product_form_silent_and_unequal_frames():
    """An all-zero frame comes out as exact zeros (its pair partner's rounding does not leak in); a frame 1e6 times quieter than its
    partner keeps its RELATIVE accuracy (power-of-two equalisers from the product kernel's max |H| column)."""
    rng = np.random.default_rng(5)
    B, T, nf = 1, 4100, 195
    Hn = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32))
    Hn[0, 10] = 0.0
    Hn[0, 21] *= 1e-6
    u = rng.random((B, T, 512), dtype=np.float32)
    y = ddsp.noise_forward(dev(Hn), 512, uniform=dev(u)).cpu().numpy().reshape(T, 512)
    ref = oracle.noise_forward(Hn, u, 512).reshape(T, 512)
    assert not y[10].any()
    assert np.max(np.abs(y[21] - ref[21])) <= TOL * max(1e-6, float(np.max(np.abs(ref[21]))))
    assert np.max(np.abs(y - ref)) <= TOL * max(1.0, float(np.max(np.abs(ref))))


def test_workspace_contract():
    """ddsp_noise_workspace_bytes is 0 where no form uses one; ddsp_noise_forward_ws without a workspace is ddsp_noise_forward."""
    L = ddsp._lib.lib()
    for B, T, nf, hop in ((512, 375, 257, 512), (512, 500, 65, 128), (1, 500, 195, 512), (1, 4, 195, 512), (8, 64, 195, 256), (0, 5, 195, 512)):
        assert L.ddsp_noise_workspace_bytes(B, T, nf, hop) == 0
    need = L.ddsp_noise_workspace_bytes(2, 2100, 195, 512)
    assert need >= 4200 * 196 * 4
    assert L.ddsp_noise_workspace_bytes(2, 300, 195, 512) > 0       # (600 frames: the backward's threshold; the forward ignores it there)
    rng = np.random.default_rng(9)
    Hn = dev(syn.controller_range(rng.standard_normal((2, 2100, 195), dtype=np.float32)))
    y0 = torch.empty(2, 2100 * 512, device="cuda")
    y1 = torch.empty_like(y0)
    s = torch.cuda.current_stream().cuda_stream
    assert L.ddsp_noise_forward(Hn.data_ptr(), None, y0.data_ptr(), 2, 2100, 195, 512, 11, 0, 0, s) == 0
    assert L.ddsp_noise_forward_ws(Hn.data_ptr(), None, y1.data_ptr(), 2, 2100, 195, 512, 11, 0, None, 0, None, 0, s) == 0
    assert torch.equal(y0, y1)
    small = torch.empty(need - 16, dtype=torch.uint8, device="cuda")      # too small: the same fallback, no overrun
    assert L.ddsp_noise_forward_ws(Hn.data_ptr(), None, y1.data_ptr(), 2, 2100, 195, 512, 11, 0, None, 0, small.data_ptr(),
                                   ctypes.c_size_t(need - 16), s) == 0
    assert torch.equal(y0, y1)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    assert L.ddsp_noise_forward_ws(Hn.data_ptr(), None, y1.data_ptr(), 2, 2100, 195, 512, 11, 0, None, 0, ws.data_ptr(),
                                   ctypes.c_size_t(need), s) == 0
    assert float((y0 - y1).abs().max()) <= TOL * max(1.0, float(y0.abs().max()))


def test_matrix_product_form_in_a_captured_graph():
    """The three launches (cosine operand, product, FFT form) and torch's workspace allocation replay as one hipGraph."""
    rng = np.random.default_rng(31)
    Hn = dev(syn.controller_range(rng.standard_normal((2, 2060, 195), dtype=np.float32)))
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    eager = ddsp.noise_forward(Hn, 512, seed=3, counter=counter).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ddsp.noise_forward(Hn, 512, seed=3, counter=counter)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = ddsp.noise_forward(Hn, 512, seed=3, counter=counter)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, eager)


@pytest.mark.parametrize("B,T,nf", [(3, 171, 195), (1, 513, 195), (2, 300, 224)])
def test_matrix_product_backward_vs_reference_autograd_and_direct_kernels(B, T, nf):
    """dH through the FFT-form correlation + transposed product against (a) torch autograd of the reference's op sequence on the CPU
    (oracle/torch_restatement.py: filtered_noise.py:7-53) and (b) the direct backward kernels it replaces.  1e-5 relative, as G17."""
    from oracle import torch_restatement as tr
    rng = np.random.default_rng(B + T + nf)
    Hn = syn.controller_range(rng.standard_normal((B, T, nf), dtype=np.float32))
    u = rng.random((B, T, 512), dtype=np.float32)
    gy = rng.standard_normal((B, T * 512)).astype(np.float32)
    gy[:, 512 * 7:512 * 8] *= 1e-5            # a quiet gradient row next to a loud one (the pair's equalisers)
    Ht = torch.from_numpy(Hn).requires_grad_()
    (tr.filtered_noise(Ht, 512, torch.from_numpy(u)) * torch.from_numpy(gy)).sum().backward()
    ref = Ht.grad.numpy()
    got = ddsp.noise_backward(dev(gy), 512, nf, uniform=dev(u)).cpu().numpy()
    direct = cosine_sums(lambda: ddsp.noise_backward(dev(gy), 512, nf, uniform=dev(u))).cpu().numpy()
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert np.max(np.abs(direct - ref)) <= 1e-5 * scale
    assert np.max(np.abs(got - ref)) <= 1e-5 * scale
    quiet = ref[:, 7]
    assert np.max(np.abs(got[:, 7] - quiet)) <= 1e-5 * max(1e-5, float(np.max(np.abs(quiet))))


def test_matrix_product_backward_through_the_module():
    """FilteredNoise autograd at the default shape: in-kernel draw, the backward regenerates it from the same counter."""
    rng = np.random.default_rng(3)
    Hn = syn.controller_range(rng.standard_normal((2, 2100, 195), dtype=np.float32))

    class Conf:
        n_harmonics, sample_rate, hop_length = 1, 44100, 512

    def grad(mode):
        L = ddsp._lib.lib()
        assert L.ddsp_noise_set_generic(mode) == 0
        try:
            H = dev(Hn).requires_grad_()
            fn = ddsp.FilteredNoise(Conf, rng="device", seed=5)
            y = fn({"H": H})
            (y * y).sum().backward()
            return y.detach().cpu().numpy(), H.grad.cpu().numpy()
        finally:
            L.ddsp_noise_set_generic(0)

    y0, g0 = grad(16)
    y1, g1 = grad(0)
    assert np.max(np.abs(y1 - y0)) <= TOL * max(1.0, float(np.max(np.abs(y0))))
    assert np.max(np.abs(g1 - g0)) <= 1e-5 * max(1.0, float(np.max(np.abs(g0))))


def test_noise_residency_knob_changes_nothing_but_the_grid():
    """ddsp_noise_set_residency (wavefronts per CU of the hop-128 kernel's persistent grid): bit-identical audio at every value, range
    checked, default 4; calibrate_noise_residency picks one of the offered levels and leaves it set."""
    L = ddsp._lib.lib()
    assert L.ddsp_noise_set_residency(0) == 0 and L.ddsp_noise_get_residency() == 4
    assert L.ddsp_noise_set_residency(9) != 0 and L.ddsp_noise_set_residency(-1) != 0
    rng = np.random.default_rng(8)
    B, T = 70, 500                                           # 2 187 groups of 16 frames: more than eight per CU, so the knob applies
    Hn = dev(syn.controller_range(rng.standard_normal((B, T, 65), dtype=np.float32)))
    try:
        outs = []
        for level in (0, 1, 3, 8):
            assert L.ddsp_noise_set_residency(level) == 0
            outs.append(ddsp.noise_forward(Hn, 128, seed=2, offset=9))
        for y in outs[1:]:
            assert torch.equal(y, outs[0])
        count = [0]

        def step():
            ddsp.noise_forward(Hn, 128, seed=2, offset=count[0])
            count[0] += 1

        r = ddsp.calibrate_noise_residency(step, levels=(2, 4, 6), settle_ms=3.0, measure_ms=3.0)
        assert r["chosen"] in (2, 4, 6) and set(r["ms_per_step"]) == {2, 4, 6} and L.ddsp_noise_get_residency() == r["chosen"]
    finally:
        L.ddsp_noise_set_residency(0)
