"""The N>1 path on CPU: two gloo ranks shard a frame stream, agree on the elapsed-time MAX and
exchange their (variable) encoded sizes -- the only collectives bench.py's multi-GPU mode uses."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hap_amd import shard  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.frames_for_rank(60, rank, world)
    sizes = [1000 * f + rank for f in mine]
    t = shard.max_over_ranks(0.25 + rank)
    rows = shard.gather_sizes(sizes)
    dist.barrier()
    out.put((rank, mine, t, rows))
    dist.destroy_process_group()


def test_two_rank_frame_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    frames = sorted(got[0][1] + got[1][1])
    assert frames == list(range(60))                       # every frame exactly once
    assert all(shard.owner_of_frame(f, world) == 0 for f in got[0][1])
    assert got[0][2] == got[1][2] == 1.25                  # MAX over ranks
    assert got[0][3] == got[1][3]
    assert got[0][3][1] == [1000 * f + 1 for f in got[1][1]]


def test_partition_helpers():
    assert [len(shard.frames_for_rank(60, r, 8)) for r in range(8)] == [8, 8, 8, 8, 7, 7, 7, 7]
    groups = [list(shard.chunk_group_for_rank(64, r, 8)) for r in range(8)]
    assert sum(groups, []) == list(range(64)) and all(len(g) == 8 for g in groups)
    assert shard.max_over_ranks(3.5) == 3.5 and shard.gather_sizes([1, 2]) == [[1, 2]]
