#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle (the test_*.py files hold the fixed cases; this is the wide
net, run by tests/test_gpu_fuzz.py with a fixed seed and from the command line for bigger sweeps):
random batch / frames / harmonics / hop / sample rate / noise bands, both f0 kinds, power-of-two and odd hops, hops 256 / 512
(in-LDS FFT noise form, impulse shorter than / equal to / longer than the hop).  Prints one line per case and a summary;
exit code 1 if any case exceeds the tolerances the tests assert (audio 1e-5, noise 2e-6 of max(1, |y|), phases bit-exact).
usage: fuzz_parity.py [cases] [seed] [training]   (`training`: the loss-side kernels against fp64 torch instead)"""
import os
import sys

import numpy as np
import torch

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
        ref, dbg = oracle.osc_forward(ctl["f0"], ctl["c"], ctl["a"], hop, sr, debug=True)
        dev = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
        small = B * T * hop * H <= 4_000_000
        y, _, phi = ddsp.osc_forward(dev["f0"], dev["c"], dev["a"], hop, sr, debug_phases=small)
        y2, _, _ = ddsp.osc_forward(dev["f0"], dev["c"], dev["a"], hop, sr)      # production (FAST) kernels
        ok_phi = True if not small else np.array_equal(bits(phi.cpu().numpy()), bits(dbg["phi"]))
        fin = np.isfinite(ref)
        e_osc = float(np.max(np.abs(y2.cpu().numpy() - ref)[fin])) if fin.any() else 0.0
        same_nan = np.array_equal(np.isnan(ref), np.isnan(y2.cpu().numpy()))
        # noise
        nhop = int(rng.choice([8, 16, 40, 64, 128, 256, 256, 512, 512, 512]))
        F = int(rng.integers(2, min(300, 2 * nhop) + 1))
        Hn = syn.controller_range(rng.standard_normal((B, T, F), dtype=np.float32))
        u = rng.random((B, T, nhop), dtype=np.float32)
        nref = oracle.noise_forward(Hn, u, nhop)
        ny = ddsp.noise_forward(torch.from_numpy(Hn).cuda(), nhop, uniform=torch.from_numpy(u).cuda())
        e_noise = float(np.max(np.abs(ny.cpu().numpy() - nref))) / max(1.0, float(np.max(np.abs(nref))))
        okay = ok_phi and same_nan and e_osc <= 1e-5 and e_noise <= 2e-6
        bad += not okay
        worst_osc, worst_noise = max(worst_osc, e_osc), max(worst_noise, e_noise)
        if verbose or not okay:
            print(f"{'ok ' if okay else 'BAD'} osc B{B} T{T} H{H} hop{hop} sr{sr} {kind}: phases {'bit-exact' if ok_phi else 'DIFFER'}"
                  f"{'' if small else ' (not dumped)'}, |dy| {e_osc:.1e} | noise hop{nhop} F{F}: {e_noise:.1e}", flush=True)
    return bad, worst_osc, worst_noise


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
    if len(sys.argv) > 3 and sys.argv[3] == "training":
        bad = sweep_training_kernels(cases, seed, verbose=False)
        print(f"loss-side kernels (one-kernel spectral scales, framing, column sums): cases {cases}, seed {seed}, failed {bad}")
        sys.exit(1 if bad else 0)
    bad, worst_osc, worst_noise = sweep(cases, seed)
    print(f"cases {cases}, failed {bad}, worst audio error {worst_osc:.2e}, worst noise error {worst_noise:.2e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
