"""Multi-GPU plumbing: the hot path shards by independent stream / frame, so ranks exchange no pixel data.

Only two things cross ranks: a barrier around the timed region and the max of the per-rank elapsed times.  Both go
through torch.distributed with the "gloo" backend, on the GPU box and in the CPU tests alike: two host scalars, no RCCL traffic."""
from __future__ import annotations


def streams_of_rank(n_streams: int, world: int, rank: int) -> list[int]:
    """Stream s is decoded on rank s % world (one decoder instance per GPU, libavcodec-style frame/stream parallelism)."""
    return [s for s in range(n_streams) if s % world == rank]


def barrier(dist, world: int, sync=None) -> None:
    if sync is not None:
        sync()
    if world > 1:
        dist.barrier()
    if sync is not None:
        sync()


def max_over_ranks(dist, torch, world: int, value: float, device) -> float:
    if world <= 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
