#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle (the test_*.py files hold the fixed cases; this is the wide
net, run by tests/test_gpu_fuzz.py with a fixed seed and from the command line for bigger sweeps):
random batch / frames / harmonics / hop / sample rate / noise bands, both f0 kinds, power-of-two and odd hops, hops 256 / 512
(in-LDS FFT noise form, impulse shorter than / equal to / longer than the hop), the benchmarked noise forms with ragged frame
counts, the in-kernel draw, and -- in a third of the cases -- loudness / filter levels spread over seven decades from frame to
frame.  Prints one line per case and a summary; exit code 1 if any case exceeds the tolerances the tests assert, taken LOCALLY:
audio 1e-5 of the loudness around each sample, noise 2e-6 of each frame's own level (or peak), phases bit-exact.
usage: fuzz_parity.py [cases] [seed] [training | chunked]   (`training`: the loss-side kernels against fp64 torch instead;
`chunked`: the chunked oscillator form with forced tilings and chunk lengths)"""
import os
import sys

import numpy as np
import torch

os.environ.setdefault("DDSP_TEST_HOOKS", "1")     # the chunked sweep pins tilings / chunk lengths through the tuning hooks
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def sweep(cases: int, seed: int, verbose: bool = True):
    """-> (failed, worst audio error, worst noise error)"""
    rng = np.random.default_rng(seed)
    worst_osc = worst_noise = 0.0
    bad = 0
    for i in range(cases):
        hop = int(rng.choice([1, 2, 3, 7, 16, 48, 64, 100, 128, 160, 256, 441, 480, 512, 1024]))
        sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
        H = int(rng.integers(1, 241))
        B = int(rng.integers(1, 5))
        T = int(rng.integers(1, max(2, min(60, 40000 // (hop * max(1, H // 8) + 1) + 2))))
        kind = "musical" if rng.random() < 0.5 else "all_live"
        shape = syn.SynthShape("fz", B, sr, hop, T, H, 2)
        ctl = syn.make_controls(shape, int(rng.integers(1 << 30)), kind)
        if rng.random() < 0.2:
            ctl["f0"][0, T // 2:, 0] = 0.0                         # a silent stretch
        if rng.random() < 0.2:
            ctl["c"][:, :, rng.integers(0, H)] = 0.0                # an exactly-zero harmonic
        if rng.random() < 0.3:                                      # loudness spread over seven decades from frame to frame
            ctl["a"] = ctl["a"] * (10.0 ** rng.uniform(-4, 3, size=ctl["a"].shape)).astype(np.float32)
        ref, dbg = oracle.osc_forward(ctl["f0"], ctl["c"], ctl["a"], hop, sr, debug=True)
        dev = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
        small = B * T * hop * H <= 4_000_000
        y, _, phi = ddsp.osc_forward(dev["f0"], dev["c"], dev["a"], hop, sr, debug_phases=small)
        y2, _, _ = ddsp.osc_forward(dev["f0"], dev["c"], dev["a"], hop, sr)      # production (FAST) kernels
        ok_phi = True if not small else np.array_equal(bits(phi.cpu().numpy()), bits(dbg["phi"]))
        fin = np.isfinite(ref)
        # per sample, relative to the loudness around it (the audio is the interpolated loudness x a sum of unit-range terms)
        a2 = ctl["a"][:, :, 0]
        around = np.repeat(np.maximum.reduce([a2, np.roll(a2, 1, axis=1), np.roll(a2, -1, axis=1)]), hop, axis=1)
        e_osc = float(np.max((np.abs(y2.cpu().numpy() - ref) / around)[fin])) if fin.any() else 0.0
        same_nan = np.array_equal(np.isnan(ref), np.isnan(y2.cpu().numpy()))
        # noise
        nhop = int(rng.choice([8, 16, 40, 64, 128, 256, 256, 512, 512, 512]))
        F = int(rng.integers(2, min(300, 2 * nhop) + 1))
        Tn = T
        if rng.random() < 0.3:                                      # the benchmarked kernel forms: wavefront-private / in-LDS FFT
            nhop, F = [(128, 65), (128, 65), (512, 257), (512, 195)][int(rng.integers(4))]
            Tn = int(rng.integers(1, 70))                           # whole groups of 16 frames + a ragged remainder
            if F in (195,) or (nhop == 512 and rng.random() < 0.3):     # >= 4 096 frames: the whole-batch matrix-product form (193..224 bands)
                if rng.random() < 0.15:
                    F = int(rng.integers(193, 225)) if rng.random() < 0.5 else 195
                    Tn = int(rng.integers(4096 // B + 1, 4096 // B + 12))
        Hn = syn.controller_range(rng.standard_normal((B, Tn, F), dtype=np.float32))
        level = np.ones((B, Tn, 1), dtype=np.float32)
        if rng.random() < 0.3:                                      # frames of very different level, exactly-zero bands
            level = (10.0 ** rng.uniform(-4, 3, size=(B, Tn, 1))).astype(np.float32)
            Hn *= level
            Hn[:, :, rng.integers(0, F)] = 0.0
        if rng.random() < 0.3:                                      # the in-kernel draw, regenerated by the oracle
            sd, off = int(rng.integers(1 << 40)), int(rng.integers(1 << 40))
            nref = oracle.noise_forward(Hn, None, nhop, seed=sd, offset=off)
            ny = ddsp.noise_forward(torch.from_numpy(Hn).cuda(), nhop, seed=sd, offset=off)
        else:
            u = rng.random((B, Tn, nhop), dtype=np.float32)
            nref = oracle.noise_forward(Hn, u, nhop)
            ny = ddsp.noise_forward(torch.from_numpy(Hn).cuda(), nhop, uniform=torch.from_numpy(u).cuda())
        # per frame, relative to the larger of the frame's level and its peak (all ones / the clip's peak without the level sweep)
        nerr = np.abs(ny.cpu().numpy() - nref).reshape(B, Tn, nhop).max(axis=2)
        npeak = np.abs(nref).reshape(B, Tn, nhop).max(axis=2)
        scale = np.maximum(level[:, :, 0], npeak)
        e_noise = float(np.max(nerr / scale))
        okay = ok_phi and same_nan and e_osc <= 1e-5 and e_noise <= 2e-6
        bad += not okay
        worst_osc, worst_noise = max(worst_osc, e_osc), max(worst_noise, e_noise)
        if verbose or not okay:
            print(f"{'ok ' if okay else 'BAD'} osc B{B} T{T} H{H} hop{hop} sr{sr} {kind}: phases {'bit-exact' if ok_phi else 'DIFFER'}"
                  f"{'' if small else ' (not dumped)'}, |dy| {e_osc:.1e} | noise T{Tn} hop{nhop} F{F}: {e_noise:.1e}", flush=True)
    return bad, worst_osc, worst_noise


def sweep_chunked(cases: int, seed: int, verbose: bool = True):
    """Random shapes of the CHUNKED oscillator form (csrc/ddsp_osc_chunk.hip) against the C oracle: power-of-two hops 64..2048,
    4 / 8 / 16 lanes per row (harmonics per lane pinned through the tuning hook so that small problems take it too), random
    chunk lengths (every offset of a chunk boundary inside a segment, one to many rounds of wavefronts), ragged row blocks,
    silent stretches, exactly-zero harmonics, loudness over seven decades, negative / huge / NaN f0 (the repair launch).
    -> (failed, worst audio error)"""
    L = ddsp._lib.lib()
    assert L.ddsp_test_hooks_enabled() == 1
    rng = np.random.default_rng(seed)
    worst = 0.0
    bad = 0
    Ks = [4, 8, 12, 13, 15, 16, 20, 23, 25]
    try:
        for i in range(cases):
            hop = int(rng.choice([64, 128, 128, 256, 512, 512, 1024, 2048]))
            sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
            while True:                                          # (K, lanes) with 4, 8 or 16 lanes per row
                K = int(rng.choice(Ks))
                G = int(rng.choice([4, 8, 16]))
                H = int(rng.integers(K * (G // 2) + 1, K * G + 1))
                if H <= 400:
                    break
            B = int(rng.integers(1, 40))
            T = int(rng.integers(1, max(2, min(80, 3_000_000 // (hop * H * B) + 2))))
            kind = "musical" if rng.random() < 0.6 else "all_live"
            shape = syn.SynthShape("fz", B, sr, hop, T, H, 2)
            ctl = syn.make_controls(shape, int(rng.integers(1 << 30)), kind)
            if rng.random() < 0.2:
                ctl["f0"][0, T // 2:, 0] = 0.0
            if rng.random() < 0.2:
                ctl["c"][:, :, rng.integers(0, H)] = 0.0
            if rng.random() < 0.3:
                ctl["a"] = ctl["a"] * (10.0 ** rng.uniform(-4, 3, size=ctl["a"].shape)).astype(np.float32)
            odd = rng.random()
            if odd < 0.08:
                ctl["f0"][B - 1, rng.integers(0, T), 0] = -150.0   # phases run backwards: declined, repaired exactly
            elif odd < 0.16:
                ctl["f0"][B - 1, :, 0] *= 300.0                    # masked harmonics still accumulate phase: beyond 1e7 rad on long clips
            elif odd < 0.2:
                ctl["f0"][B - 1, rng.integers(0, T), 0] = np.nan
            n = T * hop
            lens = [v for v in range(hop, n + 32, 32)]
            os.environ["DDSP_OSC_CHUNK_LEN"] = str(int(rng.choice(lens))) if rng.random() < 0.7 else "0"
            assert L.ddsp_osc_set_tiling(K) == 0 and L.ddsp_osc_set_path(2) == 0
            plan = ddsp._lib.osc_plan(B, T, H, hop, sr)
            if not plan["chunked"] or plan["lanes_per_row"] != G:
                continue                                           # (H / K fell on another lane count: not this sweep's business)
            ref = oracle.osc_forward(ctl["f0"], ctl["c"], ctl["a"], hop, sr)
            dev = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
            y = ddsp.osc_forward(dev["f0"], dev["c"], dev["a"], hop, sr)[0].cpu().numpy()
            fin = np.isfinite(ref)
            a2 = np.abs(ctl["a"][:, :, 0])
            around = np.repeat(np.maximum.reduce([a2, np.roll(a2, 1, axis=1), np.roll(a2, -1, axis=1)]), hop, axis=1)
            e = float(np.max((np.abs(y - ref) / np.maximum(around, 1e-30))[fin])) if fin.any() else 0.0
            same = np.array_equal(np.isfinite(y), fin)
            okay = same and e <= 1e-5
            bad += not okay
            worst = max(worst, e)
            if verbose or not okay:
                print(f"{'ok ' if okay else 'BAD'} chunked B{B} T{T} H{H} hop{hop} sr{sr} K{K} G{G} chunk {plan['chunk_samples']} {kind}"
                      f"{' odd-f0' if odd < 0.2 else ''}: |dy| {e:.1e}{'' if same else ' NON-FINITE PATTERN DIFFERS'}", flush=True)
    finally:
        os.environ.pop("DDSP_OSC_CHUNK_LEN", None)
        L.ddsp_osc_set_tiling(0)
        L.ddsp_osc_set_path(0)
    return bad, worst


def sweep_training_kernels(cases: int, seed: int, verbose: bool = True):
    """Random shapes of the loss-side kernels against torch on the CPU in fp64: ddsp_mss_scale (+ the overlap-add gather) for random
    batch / length / transform size / overlap, the framing pair around a library rfft, and the column sums.  -> failed cases"""
    from ddsp_pytorch_amd import dense
    from ddsp_pytorch_amd.training import SpectralLoss
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    bad = 0
    ratios = []
    for i in range(cases):
        n_fft = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
        overlap = float(rng.choice([0.75, 0.75, 0.5, 0.875, 0.0]))
        B = int(rng.integers(1, 6))
        L = int(rng.integers(n_fft // 2 + 1, n_fft // 2 + 1 + int(rng.choice([3, 200, 5000]))))
        x_true = 0.3 * torch.randn(B, L, generator=g)
        x_pred = 0.3 * torch.randn(B, L, generator=g)
        if rng.random() < 0.3:
            x_true[0, : L // 2] = 0.0
        sl = SpectralLoss(n_fft, alpha=float(rng.choice([1.0, 0.3])), overlap=overlap)
        xp = x_pred.double().requires_grad_(True)
        ref = sl.double()(xp, x_true.double())
        ref.backward()
        sl_gpu = SpectralLoss(n_fft, alpha=sl.alpha, overlap=overlap).cuda()
        xg = x_pred.cuda().requires_grad_(True)
        fused = sl_gpu.fused_scale(xg) is not None
        got = sl_gpu(xg, x_true.cuda())
        got.backward()
        e_loss = abs(got.item() - ref.item()) / abs(ref.item())
        gd = xg.grad.cpu().double() - xp.grad
        e_l2 = float(gd.norm() / xp.grad.norm())
        e_max = float(gd.abs().max() / xp.grad.abs().max())
        # the yardstick: what fp32 arithmetic does to this gradient in the torch formulation itself (the log term goes like
        # 1 / (|S|^2 + eps) per bin: one near-empty bin of the prediction and any fp32 transform is off by 1e-3 of the norm)
        x32 = x_pred.clone().requires_grad_(True)
        SpectralLoss(n_fft, alpha=sl.alpha, overlap=overlap)(x32, x_true).backward()
        d32 = x32.grad.double() - xp.grad
        y_l2, y_max = float(d32.norm() / xp.grad.norm()), float(d32.abs().max() / xp.grad.abs().max())
        # ... and the conditioning of the case itself: the L1 terms' gradient is discontinuous where a bin of the prediction ties
        # with the target's (sign(P - Q), sign(log Q - log P)); an fp32-epsilon perturbation of the INPUT, evaluated in fp64,
        # shows how much of the gradient such near-ties decide (single-frame and half-silent cases reach 1e-2)
        xq = (x_pred.double() + 6e-8 * 0.3 * torch.randn(x_pred.shape, generator=g, dtype=torch.float64)).requires_grad_(True)
        sl.double()(xq, x_true.double()).backward()
        dq = xq.grad - xp.grad
        c_l2, c_max = float(dq.norm() / xp.grad.norm()), float(dq.abs().max() / xp.grad.abs().max())
        y_l2, y_max = max(y_l2, c_l2), max(y_max, c_max)
        # framing pair (+ library rfft) against torch.stft, value and gradient
        xa = x_pred.cuda().requires_grad_(True)
        xb = x_pred.cuda().requires_grad_(True)
        fa = sl_gpu.stft_ri(xa)
        fb = torch.view_as_real(sl_gpu.stft(xb)).transpose(1, 2)
        w = torch.randn(fa.shape, generator=g).cuda()
        (fa * w).sum().backward()
        (fb * w).sum().backward()
        e_fr = float((fa - fb).abs().max() / fb.abs().max())
        e_frg = float((xa.grad - xb.grad).abs().max() / xb.grad.abs().max())
        # column sums
        M, N = int(rng.integers(0, 20000)), int(rng.integers(1, 1600))
        dt = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
        xm = torch.randn(M, N, generator=g).cuda().to(dt)
        e_cs = float((dense.colsum(xm).double() - xm.double().sum(0)).abs().max()) / max(1.0, float(xm.double().abs().sum(0).max())) if M else \
            float(dense.colsum(xm).abs().max())
        # (1e-3: single-frame cases -- a signal barely longer than the padding -- scatter between 2e-6 and 1e-3 for the round-2
        #  kernels, the round-3 kernels and torch's own fp32 formulation alike, dominated by the near-empty bins: measured with
        #  tools/microbench/mss_case.py)
        ratios.append(e_l2 / max(y_l2, 1e-7))
        okay = (fused and e_loss <= 2e-5 and e_l2 <= max(1e-3, 8.0 * y_l2) and e_max <= max(2e-3, 8.0 * y_max)
                and e_fr <= 3e-6 and e_frg <= 3e-6 and e_cs <= 2e-6)
        bad += not okay
        if verbose or not okay:
            print(f"{'ok ' if okay else 'BAD'} case {i}: n_fft {n_fft} overlap {overlap} B {B} L {L} | loss {e_loss:.1e} grad L2 {e_l2:.1e} (fp32 torch {y_l2:.1e}) max {e_max:.1e} ({y_max:.1e}) | "
                  f"frames {e_fr:.1e} grad {e_frg:.1e} | colsum [{M},{N}] {str(dt)[6:]} {e_cs:.1e}")
    r = np.array(ratios)
    print(f"gradient error / yardstick (the larger of torch fp32's own error and the fp32-epsilon conditioning probe): geometric mean "
          f"{float(np.exp(np.log(np.maximum(r, 1e-12)).mean())):.2f}, 90th percentile {float(np.percentile(r, 90)):.2f}, worst {float(r.max()):.2f}")
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    if len(sys.argv) > 3 and sys.argv[3] == "chunked":
        bad, worst = sweep_chunked(cases, seed, verbose=False)
        print(f"chunked oscillator form: cases {cases}, seed {seed}, failed {bad}, worst audio error {worst:.2e}")
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 3 and sys.argv[3] == "training":
        bad = sweep_training_kernels(cases, seed, verbose=False)
        print(f"loss-side kernels (one-kernel spectral scales, framing, column sums): cases {cases}, seed {seed}, failed {bad}")
        sys.exit(1 if bad else 0)
    bad, worst_osc, worst_noise = sweep(cases, seed)
    print(f"cases {cases}, failed {bad}, worst audio error {worst_osc:.2e}, worst noise error {worst_noise:.2e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
