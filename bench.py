#!/usr/bin/env python3
"""Benchmark of the DDSP synthesis hot path on MI355X (contract: see the task brief / DESIGN.md §7).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic controls resident in HBM:
OscillatorBank.forward (chunk totals, scan, synth + a repair launch that normally returns at once) + FilteredNoise.forward
accumulated into the same buffer
(`harmonics + noise`, decoder.py:132).  Workload = the configuration BASELINE.json's metric is quoted on:
batch 512 per GPU, 16 kHz, 100 harmonics, hop 128, 4 s clips, 65 noise bands (cfg4's per-GPU shard;
weak scaling: every rank synthesises its own 512 rows, no collective on the data path).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (osc_frame_synth) from HIP events
recorded on the launch stream during the timed region; `cpu_baseline` times the torch-op restatement of
the reference's CPU path (oracle/torch_restatement.py) on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh ranks under torch.distributed.run as a CHILD
    process (this parent has not imported torch nor touched the GPU, and never execs), let rank 0's JSON line through
    on the inherited stdout and return the launcher's exit code (non-zero if any rank failed)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    # decide before torch / the HIP library are imported: the parent of a multi-rank run stays GPU-free
    _pre = argparse.ArgumentParser(add_help=False)
    _pre.add_argument("--gpus", type=int, default=1)
    _n = _pre.parse_known_args()[0].gpus
    if _n > 1:
        sys.exit(self_launch(_n))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ddsp_pytorch_amd as ddsp  # noqa: E402
from ddsp_pytorch_amd import synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TLANEOPS = 78.6      # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (= 157.3 TFLOP/s fp32 vector / 2)
NOMINAL_GHZ = 2.4              # the clock the peaks above are quoted at
SYNTH_OPS = 11                 # VALU instructions of the synth kernel per harmonic-sample (12 and 10 on alternate samples)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # fp32 vector peak (FMA = 2 flop)
SURVEY_FLOPS_PER_HS = 27       # algorithmic flops per harmonic-sample of the whole oscillator path
KERNELS = ("osc_frame_totals", "osc_scan", "osc_frame_synth", "noise_frame")


def gather_rows(dist, row, world, rank):
    """Every rank's `row` (a list of floats) on every rank, as a [world, len(row)] array: an all-reduce(SUM) of a matrix in
    which each rank filled its own line -- works on every backend (RCCL and the gloo rehearsal) without object pickling."""
    m = torch.zeros(world, len(row), device="cuda", dtype=torch.float64)
    m[rank] = torch.tensor(row, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
    return m.cpu().numpy()


SETTLE_MS = 60.0               # continuous load before a timed region: the clock governor takes ~25 ms to reach its sustained clock


def settle_clock(step, first, budget_ms=SETTLE_MS):
    """Keeps the GPU under the SAME step, back to back and untimed, for `budget_ms` of wall time: after an idle gap (input
    generation on the host, a barrier) the shader clock starts near 1.9-2.1 GHz and reaches its sustained ~2.4 GHz only after ~25 ms
    of load (profiles/r04_clock_ramp.txt: the same launch takes 1.04 ms cold, 0.89 ms from the 20th on).  A 7 ms warm-up measures
    the ramp, not the kernel.  -> number of steps run."""
    n, t0 = 0, time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < budget_ms:
        for _ in range(8):
            step(first + n)
            n += 1
        torch.cuda.synchronize()
    return n


def kernel_breakdown(step, first, steps, names):
    """Average launch time of each kernel in `names`: one pass of `steps` steps per kernel, HIP events around THAT kernel only.  (Events
    around every launch of a step open enough idle gaps for the chip's power management to drop the clock: the kernels themselves then
    run ~10 % slower for the next 25 ms -- seen in the rocprofv3 trace of this very script -- so neither the timed region nor these
    passes carry more than one event pair per step.)"""
    out = {}
    for j, name in enumerate(names):
        ddsp._lib.profile_enable(2 * steps + 16, only=[name])
        for i in range(steps):
            step(first + j * steps + i)
        torch.cuda.synchronize()
        ms = [m for n, m in ddsp._lib.profile_read() if n == name]
        ddsp._lib.profile_enable(0)
        if ms:
            out[name] = float(np.mean(ms))
    return out


def time_config(shape, seed, steps, warmup, f0_kind="all_live", noise_seed=7):
    """One BASELINE.json configuration on this GPU: `steps` passes of OscillatorBank.forward + FilteredNoise accumulated
    (in-kernel draw), inputs resident.  -> dict (ms_per_step by the host clock around a synchronised region, per-kernel
    averages from HIP events on the launch stream)."""
    ctl = syn.make_controls(shape, seed, f0_kind)
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
    osc = ddsp.OscillatorBank(Conf(shape)).cuda()

    def step(i):
        y = osc(x)
        ddsp.noise_forward(x["H"], shape.hop, seed=noise_seed, offset=i << 32, out=y, accumulate=True)
        return y

    settled = settle_clock(step, 0)
    for i in range(warmup):
        y = step(settled + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        y = step(settled + warmup + i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(y).all()), "non-finite audio"
    clock = measure_clock(x, shape, step)
    # per-kernel averages from more steps, outside the timed region (kernel_breakdown)
    kern = kernel_breakdown(step, settled + warmup + steps, steps, KERNELS)
    plan = ddsp._lib.osc_plan(shape.batch, shape.frames, shape.n_harmonics, shape.hop, shape.sample_rate)
    del y, x, osc
    torch.cuda.empty_cache()
    hs = shape.batch * shape.samples * shape.n_harmonics
    return {"workload": f"batch {shape.batch}, {shape.sample_rate} Hz, {shape.n_harmonics} harmonics, hop {shape.hop}, "
                        f"{shape.frames} frames (4 s), {shape.n_noise_filters} noise bands, {f0_kind} f0, in-kernel noise draw",
            "steps": steps, "warmup": warmup, "settle_steps": settled, "ms_per_step": 1e3 * el,
            "samples_per_s": shape.batch * shape.samples / el, "kernel_ms": kern, "clock_ghz": clock, "osc_plan": plan,
            "synth_harmonic_samples_per_s": hs / (kern["osc_frame_synth"] * 1e-3) if "osc_frame_synth" in kern else None,
            "totals_harmonic_samples_per_s": hs / (kern["osc_frame_totals"] * 1e-3) if "osc_frame_totals" in kern else None,
            "synth_Mcycles": kern.get("osc_frame_synth", 0.0) * 1e-3 * clock * 1e3 if clock else None,
            "totals_Mcycles": kern.get("osc_frame_totals", 0.0) * 1e-3 * clock * 1e3 if clock else None}


class Conf:
    def __init__(self, shape):
        self.n_harmonics, self.sample_rate, self.hop_length = shape.n_harmonics, shape.sample_rate, shape.hop


def kernel_sources_stamp():
    """sha256 over the HIP sources the library is built from: ties a PMC file to the kernels it measured."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ddsp-pytorch_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "ddsp-pytorch_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc():
    """Newest profiles/rNN_pmc.json whose stamp matches the current kernel sources -> (dict, file name) or (None, None)."""
    import glob
    stamp = kernel_sources_stamp()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        if d.get("_meta", {}).get("kernel_sources_sha") == stamp:
            return d, os.path.basename(f)
    return None, None


def measure_clock(x, shape, step=None, reps=3, run=12):
    """Shader clock (GHz) the oscillator's synth kernel runs at on this box under THIS workload: a wavefront of the production launch
    stamps the in-kernel shader-clock counter and the 100 MHz wall clock at its start and end (include/ddsp_hip.h:
    ddsp_osc_clock).  Each reading is the synth launch that FOLLOWS `run - 1` back-to-back steps of the timed region's own `step`
    (oscillator + noise: the clock depends on what ran in the last milliseconds -- the governor ramps over ~25 ms, and a lone 1 ms
    launch after an idle gap reads 10 % low); median of `reps` readings, taken right after the timed region."""
    vals = []
    for _ in range(reps):
        for i in range(run - 1):
            if step is not None:
                step(i)
            else:
                ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate)
        _, _, _, scratch = ddsp.osc_forward(x["f0"], x["c"], x["a"], shape.hop, shape.sample_rate, return_scratch=True,
                                            keep_frame_scratch=False)
        g = ddsp._lib.osc_clock(scratch, shape.batch, shape.frames, shape.n_harmonics, shape.hop, shape.sample_rate,
                                torch.cuda.current_stream().cuda_stream)
        if g > 0:
            vals.append(g)
    return float(np.median(vals)) if vals else None


def live_callback_ms(calls=100):
    """The real-time callback (rt/synth.py:40-55 = Decoder.forward_live, decoder.py:139-147) at the reference's default
    configuration (config/default.py:8-24: 44.1 kHz, hop 512, 180 harmonics, 195 bands, 512-wide MLPs / GRU), 4 frames = 2048
    samples per call, host arrays in -> host audio out, as ONE hipGraph replay (GraphedLiveDecoder).  Deadline: rt/synth.py:54."""
    class LiveConf:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 180, 195, 44100, 512
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1

    torch.manual_seed(0)
    rng = np.random.default_rng(3)
    zn = {"normalized_cents": rng.uniform(0, 1, (1, 4, 1)).astype(np.float32),
          "loudness": rng.uniform(-1, 1, (1, 4, 1)).astype(np.float32),
          "f0": rng.uniform(200, 400, (1, 4, 1)).astype(np.float32)}
    dec = ddsp.Decoder(LiveConf, noise_rng="device").cuda().eval()
    live = ddsp.GraphedLiveDecoder(dec, frames=4)
    lat = []
    for i in range(calls + 10):
        t0 = time.perf_counter()
        audio = live.run(zn)
        if i >= 10:
            lat.append(time.perf_counter() - t0)
    assert audio.shape == (2048,) and np.isfinite(audio).all()
    lat = np.array(lat) * 1e3
    del live, dec
    torch.cuda.empty_cache()
    return {"workload": "Decoder.forward_live, 44.1 kHz, hop 512, 180 harmonics, 195 noise bands, 4 frames = 2048 samples per call "
                        "(config/default.py:8-24), host in -> host out, one hipGraph replay per call",
            "calls": calls, "latency_ms_median": float(np.median(lat)), "latency_ms_p99": float(np.percentile(lat, 99)),
            "deadline_ms": 1e3 * 2048 / 44100, "deadline_source": "rt/synth.py:54 (frames / sample rate)"}


def train_step_ms(amp, steps=20, warmup=8, b=32):
    """BASELINE.json configs[4] per-GPU shape on this ONE GPU (no all-reduce): decoder + HIP synth + reverb + MSS loss + fused
    Adam, batch 32, 16 kHz, 100 harmonics, 65 bands, 4 s.  amp = 'fp16' is the reference's precision=16 (train/train.py:50:
    fp16 autocast + GradScaler); 'bf16' has no scaler."""
    class TrainConf:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 100, 65, 16000, 128
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1

    frames = 500
    torch.manual_seed(0)
    model = ddsp.Decoder(TrainConf, noise_rng="device", seed=0).cuda()
    loss_fn = ddsp.MSSLoss().cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    rng = np.random.default_rng(2000)
    batch = {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (b, frames, 1)).astype(np.float32)).cuda(),
             "loudness": torch.from_numpy(rng.uniform(-1, 1, (b, frames, 1)).astype(np.float32)).cuda(),
             "f0": torch.from_numpy(syn.musical_f0(rng, b, frames)).cuda(),
             "audio": torch.from_numpy((0.1 * rng.standard_normal((b, frames * 128))).astype(np.float32)).cuda()}
    amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[amp]
    scaler = torch.amp.GradScaler("cuda") if amp == "fp16" else None
    for _ in range(warmup):
        loss, _ = ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=amp_dtype, scaler=scaler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=amp_dtype, scaler=scaler)
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert bool(torch.isfinite(loss)), "non-finite loss"
    del model, loss_fn, opt, batch
    torch.cuda.empty_cache()
    return {"workload": f"decoder (4.84 M params) + HIP synth + reverb + MSS loss (6 scales) + fused Adam, batch {b}, 16 kHz, "
                        f"100 harmonics, 65 bands, 4 s; {amp} autocast GEMMs" + (" + GradScaler (train/train.py:50 precision=16)" if scaler else ""),
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps, "host_issue_ms_per_step": 1e3 * issued / steps,
            "samples_per_s": b * frames * 128 * steps / el}


def train_step_figures():
    """Both precisions, after a throw-away pass: the first training run of a process pays one-time costs (GEMM heuristics, allocator
    growth after the synthesis configurations' large buffers were released) that are not the step's."""
    train_step_ms("bf16", steps=5, warmup=5)
    return {"note": "BASELINE.json configs[4] per-GPU shape on this one GPU (no all-reduce); a throw-away pass runs first",
            "fp16_gradscaler": train_step_ms("fp16", steps=30, warmup=10), "bf16": train_step_ms("bf16", steps=30, warmup=10)}


def cfg1_figures():
    """BASELINE.json configs[0]: one 4 s clip, 16 kHz, 60 harmonics, batch 1 -- the reference's CPU-runnable case.  Here: the GPU
    latency of one OscillatorBank.forward + FilteredNoise call (median of 50, synchronised); the CPU figure: cfg1_cpu."""
    shape = syn.CFG1
    ctl = syn.make_controls(shape, 1001, "musical")
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
    osc = ddsp.OscillatorBank(Conf(shape)).cuda()

    def step(i):
        y = osc(x)
        ddsp.noise_forward(x["H"], shape.hop, seed=7, offset=i << 32, out=y, accumulate=True)
        return y

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    lat = []
    for i in range(50):
        t0 = time.perf_counter()
        y = step(5 + i)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    assert bool(torch.isfinite(y).all())
    del x, osc, y
    return {"workload": "batch 1, 16 kHz, 60 harmonics, hop 128, 500 frames (4 s), 65 noise bands, musical f0 (BASELINE.json configs[0])",
            "gpu_latency_ms_median": 1e3 * float(np.median(lat)), "gpu_samples_per_s": shape.samples / float(np.median(lat)),
            "osc_plan": ddsp._lib.osc_plan(shape.batch, shape.frames, shape.n_harmonics, shape.hop, shape.sample_rate)}


def cfg1_cpu():
    """configs[0] as BASELINE.json names it -- the reference path on the CPU, no GPU: the torch-op restatement of the same clip on
    this box's host cores (run last, with the CPU baseline: its thread pool must not compete with the GPU figures' host threads)."""
    from oracle import torch_restatement as tr
    shape = syn.CFG1
    ctl = syn.make_controls(shape, 1001, "musical")
    f0, c, a, H = (torch.from_numpy(ctl[k]) for k in ("f0", "c", "a", "H"))
    with torch.no_grad():
        tr.oscillator_bank(f0, c, a, shape.hop, shape.sample_rate)
        reps, t0 = 0, time.perf_counter()
        while reps < 20 and time.perf_counter() - t0 < 3.0:
            yc = tr.oscillator_bank(f0, c, a, shape.hop, shape.sample_rate)
            yc += tr.filtered_noise(H, shape.hop)
            reps += 1
        cpu_el = (time.perf_counter() - t0) / reps
    return {"cpu_ms_per_clip": 1e3 * cpu_el, "cpu_samples_per_s": shape.samples / cpu_el, "cpu_threads": torch.get_num_threads(),
            "cpu_kind": "port (oracle/torch_restatement.py: the reference's torch-op sequence)"}


def live_traffic(f0_kind):
    """HBM bytes per launch of the oscillator's synth kernel measured on THIS box, now: two rocprofv3 counter passes (FETCH_SIZE,
    WRITE_SIZE: they do not fit one pass) of a child process that runs the same oscillator workload three times
    (tools/microbench/osc_only.py), corrected as MI355X_MICROARCH.md prescribes (KiB units; gfx950's FETCH_SIZE counts half of the
    bytes of coalesced reads).  -> (bytes, description) or (None, reason); never raises.  Skipped when this process itself runs under
    a profiler."""
    import csv
    import glob
    import shutil
    import tempfile
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, "bench.py runs under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="ddsp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", DDSP_TEST_HOOKS="1")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(out, counter), "--",
                                sys.executable, os.path.join(ROOT, "tools", "microbench", "osc_only.py"), "chunk", "3", "cfg4", f0_kind],
                               env=env, cwd="/tmp", timeout=180, capture_output=True)
            acc = []
            for f in glob.glob(os.path.join(out, counter, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"]
                    if row["Counter_Name"] == counter and ("osc_chunk_synth_kernel" in k or "osc_synth_kernel" in k):
                        acc.append(float(row["Counter_Value"]))
            acc = [v for v in acc if v > 0.25 * max(acc)] if acc else acc      # (the repair / twin launches that return at once)
            if not acc:
                return None, f"no {counter} rows (rocprofv3 exit {r.returncode})"
            vals[counter] = sum(acc) / len(acc)
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0, \
        "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) of tools/microbench/osc_only.py on this box, after the timed region; 2 x FETCH_SIZE + WRITE_SIZE, KiB"


def cpu_baseline(shape, seconds_target=12.0):
    """Reference CPU path (torch-op restatement) on a bounded sample: B=8 rows of the same workload."""
    from oracle import torch_restatement as tr
    # the GPU box gives one GPU a share of 16 host cores (os.cpu_count() reports the whole machine)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("DDSP_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    b = 8
    ctl = syn.make_controls(shape, 1004, "all_live", batch=b)
    f0, c, a, H = (torch.from_numpy(ctl[k]) for k in ("f0", "c", "a", "H"))
    with torch.no_grad():
        tr.oscillator_bank(f0, c, a, shape.hop, shape.sample_rate)  # warm-up
        reps, t0 = 0, time.perf_counter()
        while True:
            y = tr.oscillator_bank(f0, c, a, shape.hop, shape.sample_rate)
            y += tr.filtered_noise(H, shape.hop)
            reps += 1
            el = time.perf_counter() - t0
            if el >= seconds_target or reps >= 200:
                break
    return {"value": b * shape.samples * reps / el, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} passes of batch {b} (of {shape.batch}) x {shape.samples} samples, same shape; torch-op "
                      f"restatement of harmonic_oscillator.py+filtered_noise.py, {el:.1f} s"}


def train_mode(args, rank, world, dist):
    """BASELINE.json configs[4]: synth + MSS loss + Adam, batch 32 per GPU, one flat gradient all-reduce per step."""
    class TrainConf:
        n_harmonics, n_noise_filters, sample_rate, hop_length = 100, 65, 16000, 128
        decoder_mlp_units, decoder_mlp_layers, decoder_gru_units, decoder_gru_layers = 512, 3, 512, 1

    b, frames = args.batch or 32, 500
    torch.manual_seed(0)                                     # identical replicas on every rank
    model = ddsp.Decoder(TrainConf, noise_rng="device", seed=rank).cuda()
    loss_fn = ddsp.MSSLoss().cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=args.graph)      # one multi-tensor kernel for the whole update
    rng = np.random.default_rng(2000 + rank)
    batch = {"normalized_cents": torch.from_numpy(rng.uniform(0, 1, (b, frames, 1)).astype(np.float32)).cuda(),
             "loudness": torch.from_numpy(rng.uniform(-1, 1, (b, frames, 1)).astype(np.float32)).cuda(),
             "f0": torch.from_numpy(syn.musical_f0(rng, b, frames)).cuda(),
             "audio": torch.from_numpy((0.1 * rng.standard_normal((b, frames * 128))).astype(np.float32)).cuda()}

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    amp_dtype = {"none": None, "bf16": torch.bfloat16, "fp16": torch.float16}[args.amp]
    scaler = torch.amp.GradScaler("cuda") if args.amp == "fp16" else None
    nbytes = 0
    if args.graph:
        if scaler is not None:
            raise SystemExit("--graph: bf16 or fp32 only")
        graphed = ddsp.GraphedTrainStep(model, loss_fn, opt, batch, amp_dtype=amp_dtype)     # the whole step as hipGraph replays

        def one_step():
            return graphed.step()
    else:
        reducer = ddsp.OverlappedGradientReducer(model.parameters()) if (args.overlap_allreduce and world > 1) else None

        def one_step():
            return ddsp.train_step(model, loss_fn, opt, batch, amp_dtype=amp_dtype, scaler=scaler, reducer=reducer)
    for _ in range(args.warmup):
        _, nbytes = one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, nbytes = one_step()
    submitted = time.perf_counter() - t0                     # host time to ISSUE the steps (nothing in a step synchronises)
    fence()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # the gradient all-reduce on its own (after the timed region): the flat fp32 bucket of this model, 10 rounds between events --
    # what one step pays when nothing overlaps it; decides "one flat bucket" against --overlap-allreduce on the first real run
    allreduce_ms = None
    if dist is not None:
        flat = torch.zeros(max(1, nbytes // 4), device="cuda", dtype=torch.float32)
        for _ in range(2):
            dist.all_reduce(flat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        e0.record()
        for _ in range(10):
            dist.all_reduce(flat)
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = e0.elapsed_time(e1) / 10.0
    per_rank = gather_rows(dist, [1e3 * own_elapsed / args.steps, allreduce_ms or 0.0], world, rank)
    if rank == 0:
        print(json.dumps({
            "metric": "train-step audio samples/sec (BASELINE.json configs[4]; secondary figure)",
            "value": world * b * frames * 128 * args.steps / elapsed, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "host_issue_ms_per_step": 1e3 * submitted / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.amp == "none" else f"{args.amp} GEMMs (autocast), f32 synthesis / loss / optimiser", "data": "synthetic",
            "config": {"workload": f"decoder (4.84 M params) + HIP synth + reverb + MSS loss (6 scales) + Adam, batch {b}/GPU, "
                                   f"16 kHz, 100 harmonics, 65 noise bands, 4 s", "parallelism": f"dp{world}, " + ("bucketed all-reduces overlapping the backward" if (args.overlap_allreduce and world > 1 and not args.graph)
                                                          else "one flat all-reduce") + (", step captured as a hipGraph" if args.graph else ""),
                       "allreduce_bytes": nbytes},
            "allreduce_ms": allreduce_ms, "allreduce_GBps_per_rank": (nbytes / (allreduce_ms * 1e-3) / 1e9) if allreduce_ms else None,
            "per_rank_ms": [float(v) for v in per_rank[:, 0]], "per_rank_allreduce_ms": [float(v) for v in per_rank[:, 1]],
            "slowest_rank": int(np.argmax(per_rank[:, 0])),
            "collective_backend": dist.get_backend() if dist is not None else None,
            "rccl_ranks": dist.get_world_size() if dist is not None else 1,
            "final_loss": float(loss)}), file=RESULT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


RESULT = sys.stdout     # where the ONE JSON line goes (main() moves everything else that writes to fd 1 over to stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--f0", default="all_live", choices=["all_live", "musical"])
    ap.add_argument("--batch", type=int, default=0, help="rows per GPU (default: the metric's 512)")
    ap.add_argument("--tiling", type=int, default=0, help="force harmonics per lane (tuning)")
    ap.add_argument("--harmonics", type=int, default=0, help="override the number of harmonics (tuning experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic with two rocprofv3 child passes after the timed region (N = 1); use the committed profile")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="keep the noise kernel's default residency (4 wavefronts per CU) instead of measuring this box first")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (cfg1, cfg2, cfg3, musical f0, live callback, training step) timed after the headline at N = 1")
    ap.add_argument("--mode", default="synth", choices=["synth", "train"],
                    help="synth: the headline hot path; train: BASELINE.json configs[4] (decoder + MSS loss + Adam, "
                         "batch 32/GPU, flat RCCL gradient all-reduce) -- a secondary figure, not the metric")
    ap.add_argument("--amp", default="none", choices=["none", "bf16", "fp16"],
                    help="train mode: autocast dtype of the dense layers' GEMMs (reference: precision=16, train/train.py:50); "
                         "off by default, the synthesis kernels stay fp32 either way")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="train mode, N > 1: bucketed gradient all-reduces that start during the backward (OverlappedGradientReducer) "
                         "instead of one flat all-reduce after it")
    ap.add_argument("--graph", action="store_true",
                    help="train mode: capture the whole step (forward, loss, backward, Adam) as hipGraph replays (GraphedTrainStep)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (with --backend gloo), instead of cuda:LOCAL_RANK")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise the process group (and run every collective of the N > 1 path) with ONE rank too")
    ap.add_argument("--noise", default="device", choices=["device", "resident"],
                    help="uniform draw: in-kernel Philox, or a [B,T,hop] tensor already resident in HBM")
    args = ap.parse_args()
    # libraries print to stdout on their own (gloo's "[Gloo] Rank 0 is connected ..." at the first collective): from here on
    # fd 1 is stderr for everybody, and only the result line goes to the real stdout
    global RESULT
    try:
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        RESULT = os.fdopen(saved, "w")
    except OSError:          # no usable stderr / stdout descriptors: leave stdout alone
        RESULT = sys.stdout

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                    # a lone rank started without a launcher
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]), RANK="0", WORLD_SIZE="1")
            sock.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    if args.mode == "train":
        return train_mode(args, rank, world, dist)
    shape = syn.CFG4_PER_GPU
    if args.batch or args.harmonics:
        shape = syn.SynthShape(shape.name, args.batch or shape.batch, shape.sample_rate, shape.hop, shape.frames,
                               args.harmonics or shape.n_harmonics, shape.n_noise_filters)
    ctl = syn.make_controls(shape, 1004 + rank, args.f0)
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
    conf = Conf(shape)
    osc = ddsp.OscillatorBank(conf).cuda()
    uniform = torch.rand(shape.batch, shape.frames, shape.hop, device="cuda") if args.noise == "resident" else None
    if args.tiling:
        assert ddsp._lib.lib().ddsp_test_hooks_enabled(), "--tiling is a tuning hook: run with DDSP_TEST_HOOKS=1"
        ddsp._lib.check(ddsp._lib.lib().ddsp_osc_set_tiling(args.tiling), "ddsp_osc_set_tiling")

    def step(i):
        y = osc(x)
        ddsp.noise_forward(x["H"], shape.hop, uniform=uniform, seed=1234 + rank, offset=i << 32, out=y, accumulate=True)
        return y

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_pass(first):
        """W untimed steps, then exactly K steps between two fences -> (seconds, per-launch kernel records, last output).  Inside the
        timed region only the DOMINANT kernel (the oscillator's synth launch: `roofline`) carries HIP events: an event pair costs the
        stream ~4 us, and pairs around all four launches of a step cost the headline 2.2 % (1.366 -> 1.336 ms same box)."""
        for i in range(args.warmup):
            y = step(first + i)
        ddsp._lib.profile_enable(2 * args.steps + 16, only=["osc_frame_synth"])
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            y = step(first + args.warmup + i)
        fence()
        el = time.perf_counter() - t0
        rec = ddsp._lib.profile_read()
        ddsp._lib.profile_enable(0)
        return el, rec, y

    # (0) one-time autotuning of the noise kernel's residency on THIS box (the count of its wavefronts per CU at which the chip's power
    # management starts to drop the clock differs between boxes: DESIGN.md section 5), with this very step; then idle for half a second
    cal_i = [1 << 20]

    def cal_step():
        step(cal_i[0])
        cal_i[0] += 1

    residency = ddsp.calibrate_noise_residency(cal_step) if not args.no_calibrate else None
    torch.cuda.synchronize()
    time.sleep(0.5)
    # (1) the same W + K from an idle GPU, reported as `from_idle`; (2) more of the same step until the GPU has been under load for
    # SETTLE_MS; (3) W + K again = the line's `value`: the sustained rate of the path, not the clock governor's ramp (settle_clock)
    t_settle = time.perf_counter()
    idle_elapsed, idle_records, y = timed_pass(0)
    done = args.warmup + args.steps
    left = SETTLE_MS - 1e3 * (time.perf_counter() - t_settle)
    extra = settle_clock(step, done, left) if left > 0 else 0
    done += extra
    settle_ms = 1e3 * (time.perf_counter() - t_settle)
    elapsed, records, y = timed_pass(done)
    timed_first_launch = (cal_i[0] - (1 << 20)) + done + args.warmup   # index (from 0) of the first timed launch of each kernel in this process (calibration steps included)
    assert bool(torch.isfinite(y).all()), "non-finite audio"
    own_elapsed = elapsed
    if dist is not None:
        tmax = torch.tensor([elapsed, idle_elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, idle_elapsed = float(tmax[0].item()), float(tmax[1].item())
    idle_kernel = {}
    for name, ms in idle_records:
        idle_kernel.setdefault(name, []).append(ms)
    # the shader clock this box's synth kernel runs at, measured in the kernel right after the timed region (every VALU
    # fraction below is quoted against the 2.4 GHz peak AND against the peak at this clock: boxes of the pool differ by 10 %)
    clock = measure_clock(x, shape, step) if rank == 0 else None
    # the synth kernel's average: the timed region's own events; the other kernels': K more steps each, after the clock probe
    kern_ms = kernel_breakdown(step, done + args.warmup + args.steps, args.steps, [k for k in KERNELS if k != "osc_frame_synth"])
    kern_ms["osc_frame_synth"] = float(np.mean([ms for name, ms in records if name == "osc_frame_synth"]))
    kern_ms = {k: kern_ms[k] for k in KERNELS if k in kern_ms}
    # every rank's own clock and per-kernel averages, gathered so that the first multi-GPU run diagnoses itself: a slow rank
    # (clock, thermal, a noisy neighbour on its XCDs) shows up by index instead of hiding inside the MAX
    diag = gather_rows(dist, [1e3 * own_elapsed / args.steps] + [kern_ms.get(k, float("nan")) for k in KERNELS], world, rank)
    del y

    if rank == 0:
        samples_per_step = world * shape.batch * shape.samples
        synth_ms = kern_ms.get("osc_frame_synth", float("nan"))
        # algorithmic bytes of the oscillator per output sample = 4 (y) + 4*(H+2)/hop (c, f0, a)
        bytes_per_sample = 4.0 + 4.0 * (shape.n_harmonics + 2) / shape.hop
        launch_samples = shape.batch * shape.samples
        achieved = launch_samples * bytes_per_sample / (synth_ms * 1e-3) / 1e9
        hs_per_s = launch_samples * shape.n_harmonics / (synth_ms * 1e-3)
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (not measurable live); used
        # only when the file was measured on exactly the kernel sources this run was built from (stamp written by
        # tools/summarise_profiles.py), otherwise traffic is null rather than stale
        pmc, pmc_name = load_pmc() if not (args.batch or args.tiling or args.harmonics) else (None, None)

        def traffic_of(prefix):
            if not pmc:
                return None
            k = [v for n, v in pmc.items() if n.startswith(prefix) and isinstance(v, dict) and "hbm_bytes_per_launch" in v]
            return max(x["hbm_bytes_per_launch"] for x in k) if k else None

        traffic = traffic_of("osc_chunk_synth_kernel") or traffic_of("osc_synth_kernel")
        traffic_src = f"profiles/{pmc_name} (FETCH_SIZE x2 + WRITE_SIZE)" if traffic else None
        traffic_committed = traffic
        if world == 1 and not args.no_live_pmc and not (args.batch or args.tiling or args.harmonics):
            live, how = live_traffic(args.f0)
            if live:
                traffic, traffic_src = live, how
            else:
                traffic_src = (traffic_src or "none") + f" [live measurement skipped: {how}]"
        # the kernel SURVEY §8(d) says can approach the HBM roof: 4 (y) + 4 F / hop (H) bytes per sample (the draw is made
        # in the kernel; the accumulate's read of y is the oscillator's output coming back, not counted as algorithmic)
        noise_ms = kern_ms.get("noise_frame", float("nan"))
        noise_bps = 4.0 + 4.0 * shape.n_noise_filters / shape.hop
        noise_achieved = launch_samples * noise_bps / (noise_ms * 1e-3) / 1e9
        noise_macs = shape.hop / 2.0 + shape.n_noise_filters          # direct form: truncated convolution + inverse DFT, per sample
        at_clock = (NOMINAL_GHZ / clock) if clock else None     # peak at this clock = nominal peak x clock / 2.4
        line = {
            "metric": "audio samples/sec/GPU + %HBM-roofline, 16kHz/100-harmonic/batch512",
            "value": samples_per_step * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch {shape.batch}/GPU x {world} GPU, {shape.sample_rate} Hz, {shape.n_harmonics} harmonics, "
                                   f"hop {shape.hop}, {shape.frames} frames (4 s), {shape.n_noise_filters} noise bands "
                                   f"(BASELINE.json configs[3] per-GPU shard; metric's batch512)",
                       "f0": args.f0, "noise_rng": args.noise, "parallelism": f"batch-sharded x{world}, no collective",
                       "step": "OscillatorBank.forward + FilteredNoise.forward accumulated (harmonics + noise)",
                       "arithmetic": "fp32 with an fp64 phase accumulator (torch CPU cumsum semantics)"},
            "samples_per_sec_per_gpu": samples_per_step * args.steps / elapsed / world,
            "noise_residency": {"waves_per_cu": ddsp._lib.lib().ddsp_noise_get_residency(), "library_default": 4,
                                "calibration_ms_per_step": residency["ms_per_step"] if residency else None,
                                "note": "ddsp_pytorch_amd.calibrate_noise_residency with this step, before everything else (0.7 s, untimed): "
                                        "wavefronts per CU of the hop-128 noise kernel; above a box-dependent count the chip drops its clock"},
            "clock_settle": {"note": "the timed region above follows >= %.0f ms of the same step back to back: the clock governor needs ~25 ms "
                                     "of load to reach its sustained clock, W = %d warm-up steps are %.0f ms.  `from_idle` is the same W + K "
                                     "started on an idle GPU (the first thing this process ran), max over ranks" % (
                                         SETTLE_MS, args.warmup, args.warmup * 1e3 * elapsed / args.steps),
                             "continuous_load_ms_before_warmup": settle_ms, "extra_steps": extra,
                             "timed_launches": {"first": timed_first_launch, "count": args.steps},
                             "from_idle": {"ms_per_step": 1e3 * idle_elapsed / args.steps,
                                           "value": samples_per_step * args.steps / idle_elapsed,
                                           "kernel_ms": {k: float(np.mean(v)) for k, v in idle_kernel.items()}}},
            "clock_ghz": clock, "clock_nominal_ghz": NOMINAL_GHZ,
            "clock_source": "in-kernel: shader-clock ticks over 100 MHz wall-clock ticks across one synth wavefront (ddsp_osc_clock); "
                            "median of 3 readings, each the synth launch after 11 more back-to-back steps, right after the timed region",
            "osc_plan": ddsp._lib.osc_plan(shape.batch, shape.frames, shape.n_harmonics, shape.hop, shape.sample_rate),
            "kernel_Mcycles": {k: v * 1e-3 * clock * 1e3 for k, v in kern_ms.items()} if clock else None,
            "roofline": {"bound": "hbm", "kernel": "osc_frame_synth", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_committed_profile": traffic_committed,
                         "algorithmic_bytes_per_launch": launch_samples * bytes_per_sample,
                         "algorithmic_bytes_per_sample": bytes_per_sample, "avg_launch_ms": synth_ms,
                         "note": "kernel is VALU-bound (SURVEY §8d): see valu",
                         "valu": {"harmonic_samples_per_s": hs_per_s, "lane_ops_per_harmonic_sample": SYNTH_OPS,
                                  "achieved_Tlaneops": hs_per_s * SYNTH_OPS / 1e12, "peak_Tlaneops": VALU_PEAK_TLANEOPS,
                                  "frac": hs_per_s * SYNTH_OPS / 1e12 / VALU_PEAK_TLANEOPS,
                                  "frac_at_clock": hs_per_s * SYNTH_OPS / 1e12 / VALU_PEAK_TLANEOPS * at_clock if at_clock else None,
                                  # SURVEY §8(d)'s own accounting: 27 algorithmic flops per harmonic-sample of the whole path
                                  # against the 157.3 TFLOP/s fp32 vector peak, (a) for this kernel's launch, (b) for the whole step
                                  "survey_flops_per_harmonic_sample": SURVEY_FLOPS_PER_HS, "survey_peak_TFLOPs": FP32_VECTOR_PEAK_TFLOPS,
                                  "survey_achieved_TFLOPs": hs_per_s * SURVEY_FLOPS_PER_HS / 1e12,
                                  "survey_frac": hs_per_s * SURVEY_FLOPS_PER_HS / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                                  "survey_frac_at_clock": hs_per_s * SURVEY_FLOPS_PER_HS / 1e12 / FP32_VECTOR_PEAK_TFLOPS * at_clock if at_clock else None,
                                  "survey_frac_whole_step": (launch_samples * shape.n_harmonics / (elapsed / args.steps))
                                                            * SURVEY_FLOPS_PER_HS / 1e12 / FP32_VECTOR_PEAK_TFLOPS},
                         "noise_frame": {"bound": "hbm", "kernel": "noise_frame", "achieved": noise_achieved, "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": noise_achieved / HBM_PEAK_GBS, "traffic": traffic_of("noise_wave_kernel"),
                                         "algorithmic_bytes_per_launch": launch_samples * noise_bps,
                                         "algorithmic_bytes_per_sample": noise_bps, "avg_launch_ms": noise_ms,
                                         "note": "wavefront-private form at hop 128 (round 3): truncated convolution hop/2 multiply-adds per sample on the vector "
                                                 "pipe + F per sample as split-bf16 matrix-core products + 10 Philox rounds per 4 samples -> VALU-bound",
                                         "valu": {"mac_per_sample": noise_macs,
                                                  "achieved_Tlaneops": launch_samples * noise_macs / (noise_ms * 1e-3) / 1e12,
                                                  "peak_Tlaneops": VALU_PEAK_TLANEOPS,
                                                  "frac": launch_samples * noise_macs / (noise_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS,
                                                  "frac_at_clock": launch_samples * noise_macs / (noise_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS * at_clock if at_clock else None}}},
            "kernel_ms": kern_ms,
            "kernel_ms_source": "osc_frame_synth: HIP events inside the timed region; each other kernel: K more steps with events around that kernel "
                                "only.  An event pair is not a barrier: a reading includes what was left of the preceding kernel (the noise kernel's "
                                "tail in osc_frame_totals: +0.03 ms against the rocprofv3 trace, profiles/rNN_kernel_stats.csv)",
            "per_rank_ms": [float(v) for v in diag[:, 0]],
            "per_rank_kernel_ms": {k: [float(v) for v in diag[:, 1 + j]] for j, k in enumerate(KERNELS)},
            "slowest_rank": int(np.argmax(diag[:, 0])),
            "collective_backend": dist.get_backend() if dist is not None else None,
            "rccl_ranks": dist.get_world_size() if dist is not None else 1,
        }
        if world == 1 and not args.no_secondary and not (args.batch or args.harmonics or args.tiling):
            # SECONDARY figures, timed AFTER (outside) the headline's timed region: BASELINE.json configs[1] and configs[2] on this GPU,
            # 10 steps each.  `value` above is the metric's configuration only.
            del x, osc
            torch.cuda.empty_cache()
            line["configs"] = {"note": "secondary figures: every one timed AFTER the headline's timed region, none is part of `value`",
                               "cfg1": cfg1_figures(),
                               "cfg2": time_config(syn.CFG2, 1002, 10, 2), "cfg3": time_config(syn.CFG3, 1003, 10, 2),
                               "musical": time_config(syn.CFG4_PER_GPU, 1004, 10, 2, f0_kind="musical"),
                               "live_callback": live_callback_ms(),
                               "train_step": train_step_figures()}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(shape)
            if "configs" in line:
                line["configs"]["cfg1"].update(cfg1_cpu())
        print(json.dumps(line), file=RESULT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
