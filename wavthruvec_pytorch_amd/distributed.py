"""Data-parallel Conditional-BatchNorm for the Vec2Wav generator: batch shards + one tiny all-reduce per stage.

Every op of the generator is per-sample except the train-mode BatchNorm statistics (modules.py:23), so the
batch is split into contiguous shards, one process per GPU, and per stage only `[sum_c | sumsq_c | count]`
(2C+1 fp64 values, <= 4 KiB) is all-reduced - RCCL over xGMI when the tensors live on GPUs
(`backend="nccl"` is RCCL on PyTorch-ROCm), gloo on CPU in the tests.  Five latency-bound collectives per forward;
they cannot be fused because stage i+1's statistics depend on stage i's normalised output.
The reference itself has no SyncBN: parity target = N-rank output == single-process run on the
concatenated global batch.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first `global_batch % world_size` ranks get one extra sample."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of {world_size}')
    base, extra = divmod(global_batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank: int, world_size: int):
    """Slice every tensor of `tensors` along dim 0 to this rank's shard."""
    B = tensors[0].shape[0]
    lo, hi = shard_bounds(B, rank, world_size)
    return tuple(t[lo:hi].contiguous() for t in tensors)


class BNStatSync:
    """Callable handed to the generator: sums the per-stage statistics array over the process group in place."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, single_rank_collective: Optional[bool] = None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised: call init_process_group first')
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        # A one-rank group needs no exchange and issues none (no collective, no stream joins around it) - unless asked to: a one-rank
        # all-reduce is legal, it is the code a multi-GPU job runs, and on a one-GPU box the only way to execute and time it
        # (bench.py --force-pg / its `stat_sync` block and the -m gpu tests pass single_rank_collective=True).
        self.single_rank_collective = bool(single_rank_collective)
        self.calls = 0

    def __call__(self, stats: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1 or self.single_rank_collective:
            self.calls += 1
            if stats.is_cuda and self.backend == 'gloo':
                # gloo has no device collectives: bounce the <= 4 KiB array through the host (tests / single-GPU multi-rank)
                host = stats.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                stats.copy_(host)
            else:
                dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        return stats
