"""One huge frame split over the GPUs of a node by chunk groups.

  python tools/bench_chunk_groups.py                       # 1 GPU: the band is the whole frame
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29533 tools/bench_chunk_groups.py   # 8 GPUs: 2048 pixel rows per GPU

Rank r block-compresses and packs rows [r*H/W, (r+1)*H/W) as a band frame (chunks/W chunks), band
frames are gathered on rank 0 (RCCL send/recv) and joined (HapGpuJoinChunkGroups); then every rank
decodes its chunk group of the joined frame in place and the slices are gathered on rank 0.
Prints one JSON line on rank 0 (not the driver's bench line: that is bench.py)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=16384)
    ap.add_argument("--height", type=int, default=16384)
    ap.add_argument("--chunks", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if "RANK" in os.environ:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    import hap_amd
    from hap_amd import shard, synth

    w, h = args.width, args.height
    fmts, chunks = [0x01, 0x8DBB], [args.chunks, args.chunks]           # Hap Q Alpha
    lo, hi, band_chunks = shard.band_for_rank(h // 4, args.chunks, rank, world)
    rows = (hi - lo) * 4
    ctx = hap_amd.Context(local_rank)
    # the band is generated as a picture of its own (deterministic per rank)
    band = synth.rgba_frame(w, rows, 1000 + rank, device=dev)
    tex_bytes = [(w // 4) * (rows // 4) * b for b in (16, 8)]
    cap = hap_amd.HapMaxEncodedLength(tex_bytes, fmts, [band_chunks] * 2)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    def encode_band():
        r, used, res = ctx.encode_frames_rgba([band], w, rows, w * 4, fmts, [1, 1], [band_chunks] * 2, [out],
                                              flags=hap_amd.ENCODE_FRAGMENT_INDEX)
        assert r == 0, (r, res)
        return out[: used[0]]

    def join(parts):
        r, joined = hap_amd.HapGpuJoinChunkGroups(parts)
        assert r == 0, r
        return joined

    t = {}
    encode_band()
    fence(); t0 = time.perf_counter()
    for _ in range(args.reps):
        piece = encode_band()
    fence(); t["encode_bands_ms"] = (time.perf_counter() - t0) / args.reps * 1e3
    t0 = time.perf_counter()
    parts = shard.gather_variable(piece, root=0)
    fence(); t["gather_band_frames_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    frame = join([p.cpu().numpy() for p in parts]) if parts is not None else None
    t["join_on_host_ms"] = (time.perf_counter() - t0) * 1e3
    box = [frame]
    if dist.is_initialized():
        dist.broadcast_object_list(box, src=0)
    frame = box[0]
    dframe = torch.frombuffer(bytearray(frame), dtype=torch.uint8).to(dev)

    ok = True
    for idx in (0, 1):
        r, layout = hap_amd.HapGpuGetFrameTextureChunkLayout(frame, idx)
        assert r == 0
        whole = torch.zeros(layout[-1], dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def decode_group(first, count, dst):
            r, used, fmt = ctx.decode_chunk_group(dframe, idx, first, count, dst)
            assert r == 0 and fmt == fmts[idx], (r, fmt)

        group = shard.chunk_group_for_rank(len(layout) - 1, rank, world)
        decode_group(group.start, len(group), whole)
        fence(); t0 = time.perf_counter()
        for _ in range(args.reps):
            decode_group(group.start, len(group), whole)
        fence(); t["decode_groups_tex%d_ms" % idx] = (time.perf_counter() - t0) / args.reps * 1e3
        t0 = time.perf_counter()
        shard.exchange_slices(whole, [layout[(len(layout) - 1) * r // world] for r in range(world + 1)], root=0)
        fence(); t["gather_slices_tex%d_ms" % idx] = (time.perf_counter() - t0) * 1e3
        # every rank checks its own slice against a direct block compression of its band
        want = torch.empty(tex_bytes[idx], dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        r, _u = ctx.compress_rgba(band, w, rows, w * 4, fmts[idx], want)
        a, b = layout[group.start], layout[group.start + len(group)]
        ok = ok and r == 0 and bool(torch.equal(whole[a:b], want))
    flag = torch.tensor([1 if ok else 0], device=dev)
    if dist.is_initialized():
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        for k in list(t):
            v = torch.tensor([t[k]], dtype=torch.float64, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            t[k] = float(v.item())
    if rank == 0:
        rgba_gb = w * h * 4 / 1e9
        print(json.dumps({"workload": "one %dx%d Hap Q Alpha frame, %d+%d chunks, split by chunk groups" % (w, h, args.chunks, args.chunks),
                          "n_gpus": world, "parity": bool(flag.item()), "frame_bytes": len(frame),
                          "encode_rgba_gbps": round(rgba_gb / (t["encode_bands_ms"] / 1e3), 1),
                          "decode_rgba_gbps": round(rgba_gb / ((t["decode_groups_tex0_ms"] + t["decode_groups_tex1_ms"]) / 1e3), 1),
                          "ms": {k: round(v, 3) for k, v in t.items()}}))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
