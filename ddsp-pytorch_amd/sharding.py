"""Batch sharding for multi-GPU synthesis: batch rows are independent, so rank r of W
synthesises a contiguous slice of rows and no collective is needed on the data path."""
from __future__ import annotations


def shard_rows(batch: int, rank: int, world: int) -> tuple[int, int]:
    """[lo, hi) rows of a `batch`-row problem owned by `rank` (contiguous, sizes differ by at most one)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} / world {world}")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
