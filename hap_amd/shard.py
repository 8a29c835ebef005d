"""Frame sharding across the GPUs of one node.

Hap frames carry no inter-frame state (reference hap.c has no globals), so the stream is split
by frame index with no data-path collective: rank r of W owns frames r, r+W, r+2W, ...
The only collectives are the barrier and the MAX-reduce of the elapsed time that bench.py needs.
"""
import torch
import torch.distributed as dist


def frames_for_rank(total_frames, rank, world):
    """Round-robin frame ownership (C4: 60 frames over 8 GPUs -> 8/8/8/8/7/7/7/7)."""
    return list(range(rank, total_frames, world))


def owner_of_frame(frame_index, world):
    return frame_index % world


def chunk_group_for_rank(chunk_count, rank, world):
    """C5-style split of ONE huge frame: contiguous chunk groups per GPU."""
    lo = chunk_count * rank // world
    hi = chunk_count * (rank + 1) // world
    return range(lo, hi)


def max_over_ranks(seconds, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_sizes(local_sizes, device="cpu"):
    """All ranks learn every rank's encoded-frame sizes (variable-length outputs): the
    'exchange sizes first' step of the optional final gather."""
    if not (dist.is_available() and dist.is_initialized()):
        return [list(local_sizes)]
    world = dist.get_world_size()
    n = torch.tensor([len(local_sizes)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n)
    width = int(max(c.item() for c in counts))
    mine = torch.zeros(width, dtype=torch.int64, device=device)
    mine[: len(local_sizes)] = torch.tensor(list(local_sizes), dtype=torch.int64, device=device)
    rows = [torch.zeros(width, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(rows, mine)
    return [rows[r][: int(counts[r].item())].tolist() for r in range(world)]
