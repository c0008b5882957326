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


@pytest.mark.gpu
def test_gru_two_streams_concurrently_are_serialised_and_correct():
    """Co-residency guard: two persistent launches want one workgroup per CU each, so launches of one process
    on one device are ordered behind each other on the device (event wait on the launching stream).  Forward + backward
    issued from two streams at the same time at the training shape: correct, finite, status 0, bitwise equal to the
    one-stream results."""
    torch.manual_seed(21)
    B, T, hd = 32, 120, 512
    probs = []
    for i in range(2):
        probs.append(dict(gi=torch.randn(B, T, 3 * hd, device="cuda"), w=torch.randn(3 * hd, hd, device="cuda") * 0.05,
                          b=torch.randn(3 * hd, device="cuda") * 0.1, dy=torch.randn(B, T, hd, device="cuda")))

    def run(p, used):
        y, hT, gates, hn = gru_mod.gru_forward(p["gi"], p["w"], p["b"], None, save=True, scratch_out=used)
        return (y, hT) + tuple(gru_mod.gru_backward(p["dy"], None, p["w"], None, y, gates, hn, scratch_out=used))

    base = [run(p, []) for p in probs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    used, got = [], [None, None]
    for rep in range(3):                                         # interleaved issue: fwd/bwd of both problems in flight together
        for i, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                got[i] = run(probs[i], used)
    torch.cuda.synchronize()
    assert all(gru_mod.gru_status(s) == 0 for s in used)
    for b_, g_ in zip(base, got):
        for a, c in zip(b_, g_):
            assert torch.isfinite(c).all() and torch.equal(a, c)


@pytest.mark.gpu
def test_gru_timeout_is_loud_everywhere():
    """Fault injection (ddsp_gru_set_mode(2)): workgroup 0 stops publishing half way, its peers run into the (shortened)
    spin bound.  The status word is raised, every output of the unfinished steps is NaN -- so the weight gradients the
    caller derives from d_gi / d_gh are non-finite instead of stale memory -- and debug mode turns it into an exception."""
    from ddsp_pytorch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(22)
    B, T, n_in, hd = 8, 40, 16, 128
    _, mine = _pair(n_in, hd, 13)
    mine = mine.cuda()
    x = torch.randn(B, T, n_in, device="cuda")
    gi = torch.randn(B, T, 3 * hd, device="cuda")
    w = mine.weight_hh_l0.detach()
    good = gru_mod.gru_forward(gi, w, None, None, save=True)
    assert L.ddsp_gru_set_fault_step(T // 2) == 0 and L.ddsp_gru_set_mode(2) == 0
    try:
        used = []
        y, hT, gates, hn = gru_mod.gru_forward(gi, w, None, None, save=True, scratch_out=used)
        torch.cuda.synchronize()
        assert gru_mod.gru_status(used[0]) == 1
        # one row per group at this shape: the fault sits in row 0's group; the other groups never notice
        assert torch.equal(y[:, :T // 2], good[0][:, :T // 2]) and torch.equal(y[1:], good[0][1:])
        assert torch.isnan(hT[0]).all() and torch.isnan(y[0, T // 2 + 2:]).all()
        assert torch.isnan(gates[0, T // 2 + 2:]).all() and torch.isnan(hn[0, T // 2 + 2:]).all()
        # backward on good forward results: the reverse sweep faults at reverse step T//2, i.e. t < T - T//2 - 2 unfinished
        used = []
        d_gi, d_gh, dh0 = gru_mod.gru_backward(torch.randn(B, T, hd, device="cuda"), None, w, None, *good[0:1], *good[2:4],
                                               scratch_out=used)
        torch.cuda.synchronize()
        assert gru_mod.gru_status(used[0]) == 1
        assert torch.isnan(dh0[0]).all() and torch.isnan(d_gi[0, :T - T // 2 - 2]).all() and torch.isnan(d_gh[0, :T - T // 2 - 2]).all()
        assert torch.isfinite(d_gi[:, T - T // 2:]).all() and torch.isfinite(d_gi[1:]).all() and torch.isfinite(dh0[1:]).all()
        # module level: the parameter gradients of a step through a failed recurrence are non-finite, never garbage
        mine.zero_grad()
        y_mod, _ = mine(x)
        y_mod.square().mean().backward()
        assert not torch.isfinite(mine.weight_hh_l0.grad).all() and not torch.isfinite(mine.weight_ih_l0.grad).all()
        old = gru_mod.set_debug(True)
        try:
            with pytest.raises(_lib.DdspHipError, match="timed out"):
                mine(x)
        finally:
            gru_mod.set_debug(old)
    finally:
        L.ddsp_gru_set_mode(0)
    y_ok, _ = mine(x)                                              # the next launch is healthy again
    assert torch.isfinite(y_ok).all()


@pytest.mark.gpu
def test_gru_stacked_layers_match_torch_cpu():
    """decoder.py:60-65 passes num_layers=conf.decoder_gru_layers: stacked layers run the HIP recurrence layer by layer."""
    torch.manual_seed(31)
    ref = nn.GRU(10, 48, 3, batch_first=True)
    mine = ddsp.GRU(10, 48, 3, batch_first=True)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    x = torch.randn(4, 25, 10)
    h0 = torch.randn(3, 4, 48)
    xr = x.clone().requires_grad_(True)
    y_ref, h_ref = ref(xr, h0)
    (y_ref.square().sum() + h_ref.sum()).backward()
    xg = x.clone().cuda().requires_grad_(True)
    y, h = mine(xg, h0.cuda())
    (y.square().sum() + h.sum()).backward()
    assert float((y.detach().cpu() - y_ref.detach()).abs().max()) <= 2e-5 and float((h.detach().cpu() - h_ref.detach()).abs().max()) <= 2e-5
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())
    for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        assert float((q.grad.cpu() - p.grad).abs().max()) <= 1e-4 * float(p.grad.abs().max()) + 1e-7, n
    with torch.no_grad():                                          # and without a passed state
        y0, _ = ref(x)
        y1, _ = mine(x.cuda())
    assert float((y1.cpu() - y0).abs().max()) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,hd,with_h0", [(4, 30, 64, True), (32, 60, 512, False), (9, 17, 200, True), (40, 12, 512, True), (1, 32, 512, True),
                                            (200, 6, 128, False), (100, 8, 512, True), (250, 5, 200, False)])   # > 4 rows per group: matrix-core forward
def test_gru_bf16_matrix_core_variant_tracks_fp32(B, T, hd, with_h0):
    """The autocast variants (ddsp_gru_*_bf16: bf16 MFMA products, fp32 everything else) against the fp32 kernels on the same
    inputs: equal up to bf16 rounding of h and W in the products (|dh| ~ 1e-3), gradients with cosine >= 0.999; status 0."""
    torch.manual_seed(B * 7 + hd)
    gi = torch.randn(B, T, 3 * hd, device="cuda")
    w = torch.randn(3 * hd, hd, device="cuda") * (1.0 / hd ** 0.5)
    b = torch.randn(3 * hd, device="cuda") * 0.1
    h0 = torch.randn(B, hd, device="cuda").tanh() if with_h0 else None
    dy = torch.randn(B, T, hd, device="cuda")
    dhT = torch.randn(B, hd, device="cuda")
    ref = gru_mod.gru_forward(gi, w, b, h0, save=True)
    used = []
    got = gru_mod.gru_forward(gi, w, b, h0, save=True, scratch_out=used, lowp=True)
    assert all(gru_mod.gru_status(s) == 0 for s in used)
    for a, c, name in zip(ref, got, ("y", "hT", "gates", "hn")):
        assert torch.isfinite(c).all(), name
        assert float((a - c).abs().max()) <= 3e-2, (name, float((a - c).abs().max()))
    assert float((ref[0] - got[0]).abs().mean()) <= 2e-3
    # backward on the SAME saved forward tensors (isolates the backward products)
    y, _, gates, hn = ref
    rb = gru_mod.gru_backward(dy, dhT, w, h0, y, gates, hn)
    used = []
    gb = gru_mod.gru_backward(dy, dhT, w, h0, y, gates, hn, scratch_out=used, lowp=True)
    assert all(gru_mod.gru_status(s) == 0 for s in used)
    for a, c, name in zip(rb, gb, ("d_gi", "d_gh", "dh0")):
        assert torch.isfinite(c).all(), name
        cos = float((a * c).sum() / (a.norm() * c.norm() + 1e-30))
        assert cos >= 0.999, (name, cos)
        assert float((a - c).abs().max()) <= 5e-2 * float(a.abs().max()), name


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,hd,with_h0", [(4, 30, 64, True), (32, 60, 512, False), (9, 17, 200, True)])
def test_gru_bf16_backward_16bit_outputs_equal_the_cast_of_the_fp32_outputs(B, T, hd, with_h0):
    """io_type 1 of ddsp_gru_backward_bf16 (d_gi / d_gh stored as bf16 for the autocast GEMMs that consume them) against the same
    kernel's fp32 stores: the 16-bit tensors are the round-to-nearest-even cast of the fp32 ones (<= 1 bf16 ulp), dh0 identical."""
    torch.manual_seed(B * 3 + hd)
    gi = torch.randn(B, T, 3 * hd, device="cuda")
    w = torch.randn(3 * hd, hd, device="cuda") * (1.0 / hd ** 0.5)
    b = torch.randn(3 * hd, device="cuda") * 0.1
    h0 = torch.randn(B, hd, device="cuda").tanh() if with_h0 else None
    dy = torch.randn(B, T, hd, device="cuda")
    dhT = torch.randn(B, hd, device="cuda")
    y, _, gates, hn = gru_mod.gru_forward(gi, w, b, h0, save=True, lowp=True)
    f_gi, f_gh, f_dh0 = gru_mod.gru_backward(dy, dhT, w, h0, y, gates, hn, lowp=True)
    used = []
    h_gi, h_gh, h_dh0 = gru_mod.gru_backward(dy, dhT, w, h0, y, gates, hn, scratch_out=used, lowp=True, io16=True)
    assert all(gru_mod.gru_status(s) == 0 for s in used)
    assert h_gi.dtype == torch.bfloat16 and h_gh.dtype == torch.bfloat16 and f_gi.dtype == torch.float32
    assert torch.equal(h_dh0, f_dh0)
    for half, full, name in ((h_gi, f_gi, "d_gi"), (h_gh, f_gh, "d_gh")):
        want = full.to(torch.bfloat16)
        exact = float((half == want).float().mean())
        # one bf16 ulp = 2^-8 relative: the kernel rounds the same fp32 value, so (almost) every element is the exact cast
        assert exact >= 0.999, (name, exact)
        assert float((half.float() - full).abs().max()) <= 2.0 ** -7 * float(full.abs().max()), name


@pytest.mark.gpu
def test_gru_module_under_autocast_uses_the_bf16_variant_and_trains():
    torch.manual_seed(3)
    ref, mine = _pair(24, 128, 77)
    mine = mine.cuda()
    x = torch.randn(6, 40, 24, device="cuda")
    with torch.no_grad():
        y32, _ = mine(x)
    xg = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y16, h16 = mine(xg)
    assert y16.dtype == torch.float32 and float((y16.detach() - y32).abs().max()) <= 5e-2      # bf16 GEMM + bf16 recurrence products
    y16.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mine.parameters()) and torch.isfinite(xg.grad).all()
