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


# ---- one frame split by chunk groups: gloo ranks, the CPU checker as the codec ----
def _chunk_group_worker(rank, world, port, out, raw_band):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _data as D
    import _libs as L
    import hap_amd
    ora = L.oracle_api()
    row_bytes = 16 * 64                                        # one row of 64 DXT5 blocks
    block_rows, chunks = 12 * world, 4 * world
    tex = bytearray(D.stream_bytes(row_bytes * block_rows, "mixed", seed=9))
    if raw_band is not None:                                   # one band gains nothing -> stored as-is
        lo, hi, _ = shard.band_for_rank(block_rows, chunks, raw_band, world)
        tex[lo * row_bytes: hi * row_bytes] = D.stream_bytes((hi - lo) * row_bytes, "random", seed=4)
    tex = bytes(tex)
    lo, hi, band_chunks = shard.band_for_rank(block_rows, chunks, rank, world)

    def encode_band():
        r, frame = ora.encode([tex[lo * row_bytes: hi * row_bytes]], [L.FMT_DXT5], [L.COMP_SNAPPY], [band_chunks])
        assert r == 0
        return torch.frombuffer(bytearray(frame), dtype=torch.uint8)

    def join(frames):
        r, joined = hap_amd.HapGpuJoinChunkGroups(frames)
        assert r == 0
        return joined

    frame = shard.encode_frame_sharded(encode_band, join, root=0)
    assert (frame is None) == (rank != 0)
    box = [frame]
    dist.broadcast_object_list(box, src=0)
    frame = box[0]
    r, layout = hap_amd.HapGpuGetFrameTextureChunkLayout(frame, 0)
    assert r == 0 and layout[-1] == len(tex)

    def decode_group(first, count, dst):
        # stand-in for HapGpuDecodeChunkGroup: only the group's byte range is produced
        r, data, _fmt = ora.decode(frame, 0, len(tex))
        assert r == 0
        a, b = layout[first], layout[first + count]
        dst[a:b] = torch.frombuffer(bytearray(data[a:b]), dtype=torch.uint8)

    everywhere = shard.decode_frame_sharded(layout, decode_group, torch.zeros(len(tex), dtype=torch.uint8), root=None)
    at_root = shard.decode_frame_sharded(layout, decode_group, torch.zeros(len(tex), dtype=torch.uint8), root=0)
    whole = ora.decode(frame, 0, len(tex))
    dist.barrier()
    out.put((rank, len(layout) - 1, everywhere.numpy().tobytes() == tex, at_root.numpy().tobytes() == tex,
             whole[0] == 0 and whole[1] == tex))
    dist.destroy_process_group()


def _run_chunk_groups(world, raw_band):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_chunk_group_worker, args=(r, world, port, q, raw_band)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


def test_two_rank_chunk_group_encode_and_decode():
    got = _run_chunk_groups(2, None)
    assert [g[1] for g in got] == [8, 8]                   # 4 chunks per band, joined
    assert all(g[2] and g[4] for g in got)                  # every rank holds the whole texture; frame is a valid Hap frame
    assert got[0][3] and not got[1][3]                      # root-only gather


def test_three_rank_chunk_groups_with_a_band_stored_as_is():
    got = _run_chunk_groups(3, 1)
    assert [g[1] for g in got] == [9, 9, 9]                 # 4 + 1 (as-is band -> one chunk) + 4
    assert all(g[2] and g[4] for g in got) and got[0][3]


def test_bench_starts_its_own_ranks_and_splits_the_stream():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks of itself (torch.distributed.run,
    127.0.0.1), splits the 60-frame stream f -> rank f mod N and reduces the elapsed time with MAX: the launch path
    of the driver's SCALE runs, here with gloo and a sleep in place of the codec (--selftest-cpu)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    done = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--selftest-cpu"], capture_output=True,
                          text=True, timeout=300, env=env)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [json.loads(x) for x in done.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1                                        # rank 0 alone prints
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["frames_per_rank"] == [30, 30]
    assert line["frames_per_step"] == 60 and 25.0 < line["ms_per_step"] < 200.0      # MAX over ranks of ~30 ms
    done = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--scaling", "weak", "--frames", "7",
                           "--selftest-cpu"], capture_output=True, text=True, timeout=300, env=env)
    line = [json.loads(x) for x in done.stdout.splitlines() if x.startswith("{")][0]
    assert line["frames_per_rank"] == [7, 7] and line["frames_per_step"] == 14 and line["scaling"] == "weak"
    # a launcher that started a different number of ranks than --gpus says is an error, not a silent n_gpus
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    done = subprocess.run([sys.executable, bench, "--gpus", "2", "--selftest-cpu"], capture_output=True, text=True,
                          timeout=120, env=env2)
    assert done.returncode != 0 and "rank(s) were started" in done.stderr
    sys.path.insert(0, os.path.dirname(bench))
    import bench as B
    assert [len(B.frames_of_rank(60, r, 8, "strong")) for r in range(8)] == [8, 8, 8, 8, 7, 7, 7, 7]
    assert sorted(f for r in range(8) for f in B.frames_of_rank(60, r, 8, "strong")) == list(range(60))
    assert B.frames_of_rank(60, 3, 8, "weak") == list(range(180, 240))
