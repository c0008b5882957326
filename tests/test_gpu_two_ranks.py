"""Two ranks of the HIP path on the GPU box: each rank is a fresh python process (tests/two_rank_worker.py),
both on cuda:0, gloo rendezvous on 127.0.0.1 -- the one-GPU rehearsal of `bench.py --gpus 2` as a committed test.

  * batch-sharded synthesis: concatenated shards == the unsharded run, bit for bit (rows are independent, no collective);
  * data-parallel training: replicas stay bit-identical after two `train_step`s with one flat gradient all-reduce, and equal
    the single-process step on the whole batch (mean-reduced loss: the mean of equal shards' gradients is the batch gradient);
  * `python bench.py --gpus 2` without a launcher starts its own ranks and prints one JSON line with n_gpus = 2.

Limitation (GRU section): both ranks share ONE GPU here, and each launches the persistent GRU recurrence, whose grid must
be co-resident.  The in-process guard (occupancy check + per-device event gate) does not reach across processes; if the
driver ever failed to co-schedule the two grids a rank would run into the 2 s spin bound.  The workers therefore run with
DDSP_GRU_DEBUG=1: such a time-out raises DdspHipError naming the cause, instead of a NaN / bit-equality assertion far away.
On real multi-GPU runs every rank owns its GPU and the situation does not arise.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import ddsp_pytorch_amd as ddsp  # noqa: E402
import two_rank_common as common  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(out_dir, world=2):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", DDSP_GRU_DEBUG="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), str(out_dir)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]


@pytest.fixture(scope="module")
def rank_outputs(tmp_path_factory):
    out = tmp_path_factory.mktemp("two_ranks")
    _run_ranks(out)
    return out


def test_sharded_synthesis_equals_unsharded_bitwise(rank_outputs):
    ctl, uniform = common.synth_problem()
    x = {k: torch.from_numpy(v).cuda() for k, v in ctl.items()}
    full = common.synthesize(ddsp, x, torch.from_numpy(uniform).cuda()).cpu()
    parts = torch.cat([torch.load(rank_outputs / f"y{r}.pt", weights_only=True) for r in range(2)], dim=0)
    assert parts.shape == full.shape == (common.SYNTH.batch, common.SYNTH.samples)
    assert torch.equal(parts, full)


def test_data_parallel_replicas_lock_step_and_equal_single_process(rank_outputs):
    sd0 = torch.load(rank_outputs / "sd0.pt", weights_only=True)
    sd1 = torch.load(rank_outputs / "sd1.pt", weights_only=True)
    assert all(torch.equal(sd0[k], sd1[k]) for k in sd0)          # one all-reduced bucket -> identical updates
    model, loss_fn, opt = common.make_trainer(ddsp)
    init = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    full = {k: v.cuda() for k, v in common.train_batch().items()}
    losses = [float(ddsp.train_step(model, loss_fn, opt, full)[0]) for _ in range(common.TRAIN_STEPS)]
    ref = {k: v.cpu() for k, v in model.state_dict().items()}
    moved = 0
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        d_ref, d_got = ref[k] - init[k], sd0[k] - init[k]
        scale = float(d_ref.abs().max())
        moved += scale > 0
        # (Adam divides by sqrt(v): where a gradient entry is close to zero, fp32 summation order -- two half batches all-reduced
        # against one full batch, split-K weight gradients of different heights -- decides its update; 1e-3 of the largest update
        # still catches a wrong average or a missing shard by three orders of magnitude)
        assert float((d_got - d_ref).abs().max()) <= 1e-3 * scale + 1e-9, k
    assert moved >= 30                                             # every trainable tensor received a gradient
    # the ranks' mean loss is the batch loss (equal shards)
    l0 = torch.load(rank_outputs / "loss0.pt", weights_only=True)
    l1 = torch.load(rank_outputs / "loss1.pt", weights_only=True)
    assert abs(0.5 * float(l0[0] + l1[0]) - losses[0]) <= 1e-4 * abs(losses[0])


def test_graphed_data_parallel_steps_equal_eager_ones(rank_outputs):
    """GraphedTrainStep on two ranks (forward + backward graph, one eager flat all-reduce, update graph) ends where the eager
    data-parallel steps end, on both ranks."""
    sd = torch.load(rank_outputs / "sd0.pt", weights_only=True)
    g0 = torch.load(rank_outputs / "sdg0.pt", weights_only=True)
    g1 = torch.load(rank_outputs / "sdg1.pt", weights_only=True)
    assert all(torch.equal(g0[k], g1[k]) for k in g0)
    for k in sd:
        if sd[k].dtype.is_floating_point:
            assert float((g0[k] - sd[k]).abs().max()) <= 1e-6 * max(1e-3, float(sd[k].abs().max())), k


def test_overlapped_bucketed_reduction_equals_the_flat_one(rank_outputs):
    """OverlappedGradientReducer (buckets all-reduced while the backward still runs) on two ranks of the HIP path."""
    sd = torch.load(rank_outputs / "sd0.pt", weights_only=True)
    o0 = torch.load(rank_outputs / "sdo0.pt", weights_only=True)
    o1 = torch.load(rank_outputs / "sdo1.pt", weights_only=True)
    assert all(torch.equal(o0[k], o1[k]) for k in o0)
    for k in sd:
        if sd[k].dtype.is_floating_point:
            assert float((o0[k] - sd[k]).abs().max()) <= 1e-6 * max(1e-3, float(sd[k].abs().max())), k


def test_bench_self_launches_two_ranks():
    """bench.py --gpus 2 with no WORLD_SIZE: the parent (GPU-free) starts two ranks under torch.distributed.run, both on
    this box's one GPU (--one-device, gloo), small batch; rank 0 prints the one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "16", "--backend", "gloo", "--one-device"], env=env, capture_output=True, text=True,
                       timeout=420)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["samples_per_sec_per_gpu"] * 2 - line["value"]) <= 1e-6 * line["value"]
    assert "roofline" in line and "cpu_baseline" not in line      # the CPU baseline is an N = 1 figure
    # the N > 1 line diagnoses itself: every rank's own clock and per-kernel averages, the slowest rank by index, the
    # collective backend and how many ranks its communicator holds (the driver's runs say "nccl" = RCCL here)
    assert len(line["per_rank_ms"]) == 2 and all(v > 0 for v in line["per_rank_ms"])
    assert line["slowest_rank"] in (0, 1) and line["per_rank_ms"][line["slowest_rank"]] == max(line["per_rank_ms"])
    assert line["ms_per_step"] >= max(line["per_rank_ms"]) * (1 - 1e-6)          # the MAX over the ranks' clocks
    assert set(line["per_rank_kernel_ms"]) == {"osc_frame_totals", "osc_scan", "osc_frame_synth", "noise_frame"}
    assert all(len(v) == 2 and all(t > 0 for t in v) for v in line["per_rank_kernel_ms"].values())
    assert line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo"
    assert "configs" not in line                                   # the secondary configurations are an N = 1 figure too


def test_bench_train_mode_two_ranks_times_the_allreduce():
    """bench.py --mode train --gpus 2: the line carries the gradient all-reduce timed on its own (allreduce_ms, the 19.35 MB
    flat bucket at the full model; a small model here) and the per-rank step times."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DDSP_GRU_DEBUG"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--backend", "gloo", "--one-device"], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][0]
    assert line["n_gpus"] == 2 and line["allreduce_ms"] > 0 and line["config"]["allreduce_bytes"] > 0
    assert len(line["per_rank_ms"]) == 2 and len(line["per_rank_allreduce_ms"]) == 2 and line["rccl_ranks"] == 2


@pytest.mark.parametrize("mode", ["synth", "train"])
def test_bench_one_rank_through_rccl(mode):
    """The collectives of the N > 1 path on the backend the driver's multi-GPU runs use (`nccl` = RCCL), with the one rank this box
    allows (RCCL refuses two ranks on one device): process-group init bound to the device, barriers, the MAX / SUM all-reduces of
    the timing and the diagnostics, and -- train mode -- the flat gradient bucket's all-reduce.  What stays unmeasured is only what
    needs a second GPU: the xGMI transfers themselves."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = ["--mode", "train", "--batch", "2", "--steps", "2", "--warmup", "1"] if mode == "train" else ["--batch", "16", "--steps", "3", "--warmup", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-dist",
                        "--no-cpu-baseline", "--no-secondary"] + args, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 1 and line["collective_backend"] == "nccl" and line["rccl_ranks"] == 1
    assert line["value"] > 0 and len(line["per_rank_ms"]) == 1 and line["per_rank_ms"][0] > 0
    if mode == "train":
        assert line["allreduce_ms"] > 0 and line["config"]["allreduce_bytes"] > 0
