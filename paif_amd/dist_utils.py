"""Multi-GPU plumbing for the replicas-only data-parallel path: one process per GPU, each rank runs the
hot path on its own shard of the batch; there is NO data-path collective for configs 2-4 (every op is per-sample, and the
batch-global min/max is defined per rank, exactly what wrapping the reference in DDP would do).  The only collectives are
the timing barrier and the max-over-ranks of the elapsed time (RCCL on GPUs, gloo in the CPU tests), plus the optional
2-float min/max all-reduce below that reproduces single-process large-batch semantics."""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(rank, world, per_rank):
    """Samples [start, start + per_rank) of rank `rank` (weak scaling: per-rank batch fixed)."""
    return rank * per_rank, per_rank


def max_over_ranks(seconds, dist=None, device="cpu"):
    """The bench contract's elapsed time: MAX over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def global_minmax(mn, mx, dist=None):
    """Optional `global_minmax=True` mode: min/max of the glue's batch-global normalisation across ALL ranks
    (one all-reduce of 2 floats as min over [mn, -mx])."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mn, mx
    t = torch.stack([mn.reshape(()), -mx.reshape(())])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t[0], -t[1]
