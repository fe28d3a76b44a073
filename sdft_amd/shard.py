"""Channel sharding for one-process-per-GPU runs (torch.distributed; backend nccl = RCCL, or gloo).

The unit of parallelism across GPUs is the independent channel/stream (one plan each, no shared
mutable state -- reference sdft.h:145-182), so the data path needs no collective: every rank
analyses its own contiguous block of channels.  Collectives are used only to bracket timing
(barrier) and to gather scalars (max elapsed time, sample counts).
"""

from __future__ import annotations

from typing import List, Tuple


def channel_block(channels: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [first, first+count) of `channels` owned by `rank`; remainders go to the
    lowest ranks so block sizes differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(int(channels), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def all_blocks(channels: int, world_size: int) -> List[Tuple[int, int]]:
    return [channel_block(channels, world_size, r) for r in range(world_size)]


def weak_scaling_channels(per_gpu: int, world_size: int) -> int:
    """BASELINE config 5: a fixed number of channels per GPU (64), so total = 64 * N."""
    return int(per_gpu) * int(world_size)


def barrier(device=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if device is not None and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device])
        else:
            dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of one scalar (the job is as slow as its slowest rank)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = f"cuda:{device}" if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = f"cuda:{device}" if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def min_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    dev = f"cuda:{device}" if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def run_census(local_seconds: float, local_frac: float, device=None, placement_gbs: float = 0.0, placed: bool = True,
               first_frac: float = 0.0) -> dict:
    """What a multi-GPU record needs to be checked without reading logs: how many ranks took part in the collectives
    of which backend on how many distinct devices, the spread of the per-rank step time and roofline fraction, and -- so
    that a slow rank is explained by its memory, not guessed -- the spread of the store-only rate of the ranks' matrices
    (placement_gbs: 0 on a rank that took its first allocation), how many ranks had their matrix placed by the library,
    and the spread of the roofline fraction into the ranks' FIRST allocations."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "ranks": 1, "world_size": 1}
    backend = dist.get_backend()
    on_gpu = backend == "nccl"
    dev_index = int(device) if device is not None else 0
    # one-hot of the device index, summed: the number of non-zero slots is the number of distinct GPUs the ranks sit on
    dev = f"cuda:{device}" if (device is not None and on_gpu) else "cpu"
    hot = torch.zeros(64, dtype=torch.float64, device=dev)
    hot[dev_index % 64] = 1.0
    dist.all_reduce(hot, op=dist.ReduceOp.SUM)
    return {
        "backend": "rccl (torch.distributed nccl)" if on_gpu else backend,
        "ranks": int(round(sum_over_ranks(1.0, device))),
        "world_size": dist.get_world_size(),
        "distinct_local_devices": int((hot > 0).sum().item()),
        "seconds_min": min_over_ranks(local_seconds, device),
        "seconds_max": max_over_ranks(local_seconds, device),
        "roofline_frac_min": min_over_ranks(local_frac, device),
        "roofline_frac_max": max_over_ranks(local_frac, device),
        "placement_gbs_min": min_over_ranks(placement_gbs, device),
        "placement_gbs_max": max_over_ranks(placement_gbs, device),
        "ranks_placed": int(round(sum_over_ranks(1.0 if placed else 0.0, device))),
        "first_allocation_frac_min": min_over_ranks(first_frac, device),
        "first_allocation_frac_max": max_over_ranks(first_frac, device),
    }


def job_throughput(local_units: float, local_seconds: float, device=None) -> Tuple[float, float]:
    """(total units over all ranks) / (max seconds over ranks) -> (units per second, seconds)."""
    total = sum_over_ranks(local_units, device)
    secs = max_over_ranks(local_seconds, device)
    return (total / secs if secs > 0 else 0.0), secs
