"""Rank body of tests/test_gpu_two_ranks.py: started as a FRESH python process per rank (RANK / WORLD_SIZE /
MASTER_* in the environment), every rank on cuda:0 of the one-GPU box, gloo for the rendezvous -- the committed form of
the `--one-device` rehearsal.  Everything on the data path is the HIP library (no oracle in here).

argv: out_dir
  1. batch-sharded synthesis (BASELINE.json configs[3]): this rank's contiguous rows of one global batch
     through OscillatorBank + FilteredNoise -> y{rank}.pt (the parent compares the concatenation with the unsharded run)
  2. data-parallel training step (configs[4]): identical replicas, this rank's half of the batch, ONE flat all-reduce
     of the gradients through `train_step` -> sd{rank}.pt; the same through `GraphedTrainStep` -> sdg{rank}.pt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    import ddsp_pytorch_amd as ddsp
    import two_rank_common as common

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- 1. sharded synthesis ------------------------------------------------------------------------------------
    ctl, uniform = common.synth_problem()
    lo, hi = ddsp.sharding.shard_rows(common.SYNTH.batch, rank, world)
    x = {k: torch.from_numpy(v[lo:hi]).cuda() for k, v in ctl.items()}
    y = common.synthesize(ddsp, x, torch.from_numpy(uniform[lo:hi]).cuda())
    torch.cuda.synchronize()
    torch.save(y.cpu(), os.path.join(out_dir, f"y{rank}.pt"))
    # bench.py's timing reduction: MAX over ranks of a per-rank scalar
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 0.5 + world - 1

    # ---- 2. data-parallel step -----------------------------------------------------------------------------------
    model, loss_fn, opt = common.make_trainer(ddsp)
    full = common.train_batch()
    lo, hi = ddsp.sharding.shard_rows(common.TRAIN_ROWS, rank, world)
    shard = {k: v[lo:hi].cuda() for k, v in full.items()}
    losses = []
    for _ in range(common.TRAIN_STEPS):
        loss, nbytes = ddsp.train_step(model, loss_fn, opt, shard)
        losses.append(float(loss))
    assert nbytes == 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(out_dir, f"sd{rank}.pt"))
    torch.save(torch.tensor(losses), os.path.join(out_dir, f"loss{rank}.pt"))

    # ---- 3. the same steps as hipGraph replays (GraphedTrainStep: graph, ONE eager all-reduce, graph) ---------------
    model_g, loss_g, opt_g = common.make_trainer(ddsp)
    graphed = ddsp.GraphedTrainStep(model_g, loss_g, opt_g, shard)
    for _ in range(common.TRAIN_STEPS):
        graphed.step(shard)
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in model_g.state_dict().items()}, os.path.join(out_dir, f"sdg{rank}.pt"))

    # ---- 4. the same steps with the bucketed reducer whose all-reduces start during the backward ---------------------
    model_o, loss_o, opt_o = common.make_trainer(ddsp)
    reducer = ddsp.OverlappedGradientReducer(model_o.parameters(), bucket_bytes=16 << 10)
    assert len(reducer.buckets) >= 3
    for _ in range(common.TRAIN_STEPS):
        ddsp.train_step(model_o, loss_o, opt_o, shard, reducer=reducer)
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in model_o.state_dict().items()}, os.path.join(out_dir, f"sdo{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
