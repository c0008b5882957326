"""GRU recurrence (csrc/ddsp_gru.hip) against torch's own CPU nn.GRU -- what the reference's controller runs
(model/autoencoder/decoder.py:60-65, :91).  Tolerances: fp32, different summation order: 2e-5 abs on h in [-1, 1],
1e-4 of the largest entry on gradients."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import ddsp_pytorch_amd as ddsp
from ddsp_pytorch_amd import gru as gru_mod


def _pair(n_in, hd, seed, batch_first=True):
    torch.manual_seed(seed)
    ref = nn.GRU(n_in, hd, 1, batch_first=batch_first)
    mine = ddsp.GRU(n_in, hd, 1, batch_first=batch_first)
    mine.load_state_dict(ref.state_dict(), strict=True)
    return ref, mine


def test_gru_is_a_stock_gru_on_cpu():
    ref, mine = _pair(5, 12, 0)
    assert set(mine.state_dict()) == {"weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"}
    x = torch.randn(3, 7, 5)
    y0, h0 = ref(x)
    y1, h1 = mine(x)
    assert torch.equal(y0, y1) and torch.equal(h0, h1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,n_in,hd,with_h0", [
    (3, 7, 5, 12, False),        # the fixtures' controller width
    (2, 6, 32, 12, True),
    (5, 20, 16, 100, True),      # hidden size not a multiple of 16
    (70, 9, 8, 64, False),
    (9, 33, 24, 200, True),
    (32, 50, 64, 512, False),    # training shape (BASELINE.json configs[4] widths), 4 rows per group
    (40, 11, 16, 512, True),     # 5 rows per group: two register tiles
    (1, 32, 1024, 512, True),    # live callback: one row, carried state
])
def test_gru_forward_matches_torch_cpu(B, T, n_in, hd, with_h0):
    ref, mine = _pair(n_in, hd, B * 1000 + hd)
    mine = mine.cuda()
    rng = np.random.default_rng(hd + T)
    x = torch.from_numpy(rng.standard_normal((B, T, n_in)).astype(np.float32))
    h0 = torch.from_numpy(rng.standard_normal((1, B, hd)).astype(np.float32)) if with_h0 else None
    with torch.no_grad():
        y_ref, h_ref = ref(x, h0)
        y, h = mine(x.cuda(), None if h0 is None else h0.cuda())
    assert torch.isfinite(y).all()                               # a hand-off timeout poisons the outputs with NaN
    assert y.shape == y_ref.shape and h.shape == h_ref.shape
    assert float((y.cpu() - y_ref).abs().max()) <= 2e-5
    assert float((h.cpu() - h_ref).abs().max()) <= 2e-5
    assert torch.equal(h[0], y[:, -1])


@pytest.mark.gpu
def test_gru_time_major_layout():
    ref, mine = _pair(6, 20, 5, batch_first=False)
    mine = mine.cuda()
    x = torch.randn(9, 4, 6)
    with torch.no_grad():
        y_ref, h_ref = ref(x)
        y, h = mine(x.cuda())
    assert float((y.cpu() - y_ref).abs().max()) <= 2e-5 and float((h.cpu() - h_ref).abs().max()) <= 2e-5


@pytest.mark.gpu
def test_gru_state_carry_equals_one_long_call():
    _, mine = _pair(16, 512, 9)
    mine = mine.cuda()
    x = torch.randn(1, 96, 16, device="cuda")
    with torch.no_grad():
        y_all, h_all = mine(x)
        h = None
        parts = []
        for i in range(3):                                   # three callbacks of 32 frames (rt/synth.py:40-55)
            yi, h = mine(x[:, 32 * i:32 * i + 32], h)
            parts.append(yi)
    assert torch.equal(torch.cat(parts, 1), y_all) and torch.equal(h, h_all)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,n_in,hd,with_h0", [
    (3, 7, 5, 12, True),
    (5, 20, 16, 100, False),
    (32, 40, 64, 512, True),
    (150, 5, 8, 512, False),     # more rows than one backward launch takes: split into slices
    (20, 6, 8, 512, True),       # 3 rows per group (partial register tile)
])
def test_gru_backward_matches_torch_cpu_autograd(B, T, n_in, hd, with_h0):
    ref, mine = _pair(n_in, hd, B + hd)
    mine = mine.cuda()
    rng = np.random.default_rng(B + T)
    x = torch.from_numpy(rng.standard_normal((B, T, n_in)).astype(np.float32))
    h0 = torch.from_numpy(rng.standard_normal((1, B, hd)).astype(np.float32)) if with_h0 else None
    wy = torch.from_numpy(rng.standard_normal((B, T, hd)).astype(np.float32))
    wh = torch.from_numpy(rng.standard_normal((1, B, hd)).astype(np.float32))

    def run(mod, dev):
        xs = x.clone().to(dev).requires_grad_(True)
        hs = None if h0 is None else h0.clone().to(dev).requires_grad_(True)
        y, h = mod(xs, hs)
        loss = (y * wy.to(dev)).sum() + (h * wh.to(dev)).sum()
        loss.backward()
        grads = {"x": xs.grad, **{k: p.grad for k, p in mod.named_parameters()}}
        if hs is not None:
            grads["h0"] = hs.grad
        return {k: v.detach().cpu() for k, v in grads.items()}

    g_ref, g = run(ref, "cpu"), run(mine, "cuda")
    assert set(g) == set(g_ref)
    for k in g_ref:
        scale = float(g_ref[k].abs().max()) + 1e-12
        assert float((g[k] - g_ref[k]).abs().max()) <= 1e-4 * scale, k


@pytest.mark.gpu
def test_gru_inference_needs_no_saved_tensors_and_rejects_bad_state():
    _, mine = _pair(8, 32, 3)
    mine = mine.cuda()
    x = torch.randn(2, 5, 8, device="cuda")
    with torch.no_grad():
        y, _ = mine(x)
    assert not y.requires_grad
    with pytest.raises(RuntimeError):
        mine(x, torch.zeros(1, 3, 32, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1])
def test_gru_hand_off_is_placement_independent(mode):
    """Mode 1 deals every group's workgroups over all XCDs, so every hand-off crosses XCDs: bitwise the same results
    as the default (one group per XCD), forward and backward, with the GPU busy on a second stream (uneven load)."""
    from ddsp_pytorch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(mode)
    B, T, hd = 6, 60, 512
    gi = torch.randn(B, T, 3 * hd, device="cuda")
    w = torch.randn(3 * hd, hd, device="cuda") * 0.05
    b = torch.randn(3 * hd, device="cuda") * 0.1
    h0 = torch.randn(B, hd, device="cuda")
    dy = torch.randn(B, T, hd, device="cuda")

    def run():
        used = []
        y, hT, gates, hn = gru_mod.gru_forward(gi, w, b, h0, save=True, scratch_out=used)
        out = gru_mod.gru_backward(dy, None, w, h0, y, gates, hn, scratch_out=used)
        assert [gru_mod.gru_status(s) for s in used] == [0, 0]    # no workgroup gave up waiting for its peers
        return (y, hT, gates, hn) + tuple(out)

    base = run()
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda")
    try:
        assert L.ddsp_gru_set_mode(mode) == 0
        with torch.cuda.stream(side):
            for _ in range(4):
                big = big @ big * 1e-4                          # competing work while the persistent kernels run
        got = run()
    finally:
        L.ddsp_gru_set_mode(0)
    torch.cuda.synchronize()
    for a, c in zip(base, got):
        assert torch.equal(a, c)


@pytest.mark.gpu
def test_gru_recurrence_is_graph_capturable():
    """memset + one kernel, no allocation or synchronisation inside: the launch can sit in a hipGraph (the live path
    replays one graph per callback); the replay re-zeroes the hand-off granules itself."""
    torch.manual_seed(4)
    B, T, hd = 2, 24, 512
    gi = torch.randn(B, T, 3 * hd, device="cuda")
    w = torch.randn(3 * hd, hd, device="cuda") * 0.05
    b = torch.randn(3 * hd, device="cuda") * 0.1
    h0 = torch.randn(B, hd, device="cuda")
    ref_y, ref_h, _, _ = gru_mod.gru_forward(gi, w, b, h0, save=False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gru_mod.gru_forward(gi, w, b, h0, save=False)          # warm-up on the capture stream
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            y, hT, _, _ = gru_mod.gru_forward(gi, w, b, h0, save=False)
    for _ in range(3):
        y.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ref_y) and torch.equal(hT, ref_h)
    gi.mul_(0.5)                                                # new inputs in the captured buffers
    new_y, _, _, _ = gru_mod.gru_forward(gi, w, b, h0, save=False)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, new_y)
