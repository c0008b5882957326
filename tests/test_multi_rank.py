"""World-size-2 gloo test (CPU) of the batch-sharded multi-GPU path: every rank owns a contiguous slice of
the batch rows, there is no data-path collective, and the timing reduction (MAX over ranks) works.
The per-rank synthesis runs through the CPU oracle here (no GPU in this container); on the GPU box the same
sharding is exercised by bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddsp_pytorch_amd import synthetic as syn
    from ddsp_pytorch_amd.sharding import shard_rows
    from oracle import oracle
    shape = syn.SynthShape("mr", 5, 16000, 64, 12, 20, 17)
    ctl = syn.make_controls(shape, 77, "musical")          # the same global batch on every rank
    lo, hi = shard_rows(shape.batch, rank, world)
    y = oracle.osc_forward(ctl["f0"][lo:hi], ctl["c"][lo:hi], ctl["a"][lo:hi], shape.hop, shape.sample_rate)
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)               # bench.py's elapsed-time reduction
    assert float(t) == 0.5 + world - 1
    dist.destroy_process_group()


def test_batch_sharding_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from ddsp_pytorch_amd import synthetic as syn
    from oracle import oracle
    shape = syn.SynthShape("mr", 5, 16000, 64, 12, 20, 17)
    ctl = syn.make_controls(shape, 77, "musical")
    full = oracle.osc_forward(ctl["f0"], ctl["c"], ctl["a"], shape.hop, shape.sample_rate)
    parts = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(parts, full)                      # rows are independent: sharded == unsharded, bit for bit


def test_shard_rows_partition():
    from ddsp_pytorch_amd.sharding import shard_rows
    for batch in (0, 1, 5, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_self_launch_relays_rank_failure():
    """`python bench.py --gpus 2` without a launcher: the parent (which imports neither torch nor the HIP library) starts the
    ranks under torch.distributed.run as a child process.  No GPU in this container, so every rank stops with "needs a
    GPU": the parent must hand that failure on as a non-zero exit code (on the GPU box the same path is run for real by
    tests/test_gpu_two_ranks.py)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in (r.stdout + r.stderr)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]     # no bench line from a failed run
