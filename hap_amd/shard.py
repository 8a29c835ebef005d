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


# ----------------------------------------------------------------------------------------------
# One huge frame split over the GPUs by chunk groups.
#
# Encode: rank r block-compresses and packs its band of block rows as a Hap frame of its own with
# chunks/W chunks; the variable-sized band frames travel to the root (sizes first, then one
# grouped send/recv -- direct xGMI links into the root, no ring) where HapGpuJoinChunkGroups
# concatenates the chunk lists into one ordinary frame.
# Decode: every rank holds the frame, decodes its chunk group in place in a full-size buffer
# (HapGpuDecodeChunkGroup); the equal- or unequal-sized slices are then exchanged.
# The only data-path messages are these optional gathers; the codec itself never communicates.
# ----------------------------------------------------------------------------------------------

def band_for_rank(block_rows, chunk_count, rank, world):
    """Rows of 4x4 blocks [lo, hi) and the chunk count of rank's band.  Chunk boundaries must
    fall on band boundaries: chunk_count and block_rows both divisible by world."""
    if chunk_count % world or block_rows % world:
        raise ValueError("chunk count %d and block rows %d must both be divisible by the %d ranks"
                         % (chunk_count, block_rows, world))
    rows = block_rows // world
    return rank * rows, (rank + 1) * rows, chunk_count // world


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _through_host():
    """gloo moves device tensors for all_reduce / broadcast only: under gloo (CPU tests, and the one-GPU dry run of
    bench.py's multi-rank path) point-to-point messages and all_gather go through host copies.  Under RCCL (nccl)
    everything stays on the device."""
    return dist.get_backend() == "gloo"


def gather_variable(local, root=0):
    """Gathers 1-D uint8 tensors of different lengths on `root`: returns the list there, None
    elsewhere.  Sizes are exchanged first, then every rank posts one send and the root W-1
    receives in a single group."""
    rank, world = _world()
    if world == 1:
        return [local]
    staged = _through_host() and local.is_cuda
    sizes = gather_sizes([int(local.numel())], device="cpu" if staged else local.device)
    wire = local.cpu() if staged else local
    ops, parts = [], None
    if rank == root:
        parts = [wire if r == root else torch.empty(sizes[r][0], dtype=torch.uint8, device=wire.device)
                 for r in range(world)]
        ops = [dist.P2POp(dist.irecv, parts[r], r) for r in range(world) if r != root and sizes[r][0]]
    elif local.numel():
        ops = [dist.P2POp(dist.isend, wire, root)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if staged and parts is not None:
        parts = [local if r == root else parts[r].to(local.device) for r in range(world)]
    return parts


def exchange_slices(buf, bounds, root=None):
    """`buf` is a full-size 1-D buffer of which this rank has filled [bounds[rank], bounds[rank+1]).
    root=None: every rank ends up with every slice; otherwise only `root` does."""
    rank, world = _world()
    if world == 1:
        return buf
    if root is None:
        for r in range(world):
            if bounds[r + 1] > bounds[r]:
                dist.broadcast(buf[bounds[r]: bounds[r + 1]], src=r)
        return buf
    ops, landing = [], {}
    staged = _through_host() and buf.is_cuda
    if rank == root:
        for r in range(world):
            if r != root and bounds[r + 1] > bounds[r]:
                piece = buf[bounds[r]: bounds[r + 1]]
                if staged:
                    landing[r] = torch.empty(piece.numel(), dtype=buf.dtype)
                ops.append(dist.P2POp(dist.irecv, landing.get(r, piece), r))
    elif bounds[rank + 1] > bounds[rank]:
        piece = buf[bounds[rank]: bounds[rank + 1]]
        ops = [dist.P2POp(dist.isend, piece.cpu() if staged else piece, root)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for r, host in landing.items():
        buf[bounds[r]: bounds[r + 1]].copy_(host)
    return buf


def encode_frame_sharded(encode_band, join, root=0):
    """encode_band() -> 1-D uint8 tensor holding this rank's band frame (any length);
    join(list of bytes) -> joined frame.  Returns the joined frame on root, None elsewhere."""
    band = encode_band()
    parts = gather_variable(band, root)
    if parts is None:
        return None
    return join([p.cpu().numpy().tobytes() for p in parts])


def decode_frame_sharded(chunk_layout, decode_group, out, root=None):
    """chunk_layout: decoded offsets of the texture's chunks (n + 1 entries);
    decode_group(first, count, out) decodes those chunks into `out` in place.
    Every rank decodes its contiguous chunk group; slices are then exchanged (see exchange_slices)."""
    rank, world = _world()
    n = len(chunk_layout) - 1
    groups = [chunk_group_for_rank(n, r, world) for r in range(world)]
    mine = groups[rank]
    if len(mine):
        decode_group(mine.start, len(mine), out)
    bounds = [chunk_layout[n * r // world] for r in range(world + 1)]
    return exchange_slices(out[: chunk_layout[n]], bounds, root)
