#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: RGBA GB/s (+ frames/s), 8K Hap Q
encode + decode, batched 60-frame stream (configs[3] / "C4": 7680x4320, scaled YCoCg-DXT5,
24 chunks, Snappy), frames resident in HBM when the timed region starts.

One "step" = one pass of the hot path over one 60-frame batch:
    RGBA (HBM) -> [block_encode] -> YCoCg-DXT5 texture -> [snappy_compress, frame_pack,
    frame_gather] -> Hap Q frame (HBM) -> [decode_plan, snappy_decode] -> texture (HBM)
value = frames * W*H*4 bytes / seconds for encode+decode together (decimal GB/s).

Multi-GPU: one process per GPU (torchrun), frames are independent so every rank runs the
same pipeline on its own 60 frames (weak scaling), no data-path collective; the only
torch.distributed traffic is the barrier and the MAX-reduce of the elapsed time.

Extra objects on the JSON line: "roofline" for the dominant kernel (HIP events recorded on the
library's own stream around every launch of the timed region) and "cpu_baseline" (the
unmodified reference hap.c + libsnappy from oracle/_ref when present, else the C port in
oracle/, plus the oracle's scalar block encoder -- the reference has no RGBA stage).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CONFIGS = {
    # name: (width, height, texture formats, chunk counts, default frames per batch)
    "C2": (3840, 2160, [0x83F0], [1], 60),
    "C3": (3840, 2160, [0x83F3], [8], 60),
    "C4": (7680, 4320, [0x01], [24], 60),
    "C5": (16384, 16384, [0x01, 0x8DBB], [64, 64], 4),
}
BLOCK_BYTES = {0x83F0: 8, 0x8DBB: 8, 0x83F3: 16, 0x01: 16}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C4", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0, help="frames per batch (default: the config's)")
    ap.add_argument("--no-fragment-index", action="store_true")
    ap.add_argument("--frag-log2", type=int, default=0, help="Snappy fragment size (log2 bytes); 0 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--foreign-frames", type=int, default=24,
                    help="also time decoding of N frames made by the CPU reference encoder (no fragment table); 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import hap_amd
    from hap_amd import synth

    w, h, fmts, chunks, nf_default = CONFIGS[args.config]
    nf = args.frames or nf_default
    ctx = hap_amd.Context(local_rank)
    if args.frag_log2:
        ctx.set_fragment_log2(args.frag_log2)
    flags = 0 if args.no_fragment_index else hap_amd.ENCODE_FRAGMENT_INDEX
    tex_bytes = [(w // 4) * (h // 4) * BLOCK_BYTES[f] for f in fmts]
    cap = hap_amd.HapMaxEncodedLength(tex_bytes, fmts, chunks)
    rgba_bytes = w * h * 4

    rgba = [synth.rgba_frame(w, h, rank * nf + i, device=dev) for i in range(nf)]
    frames = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(nf)]
    dec = [[torch.empty(tb, dtype=torch.uint8, device=dev) for _ in range(nf)] for tb in tex_bytes]
    comps = [1] * len(fmts)
    # the buffers stay where they are for the whole run: resolve their addresses once, as a C client would
    rgba = hap_amd.BufferList(rgba)
    frames = hap_amd.BufferList(frames)
    dec = [hap_amd.BufferList(d) for d in dec]
    torch.cuda.synchronize()

    used_box = [None]

    def step():
        r, used, results = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, comps, chunks, frames, flags=flags)
        if r != 0:
            raise RuntimeError("encode failed: %r %r" % (r, results[:4]))
        used_box[0] = used
        for idx in range(len(fmts)):
            r, dused, dfmts, dres = ctx.decode_frames(frames, used, idx, dec[idx])
            if r != 0 or dused[0] != tex_bytes[idx]:
                raise RuntimeError("decode failed: %r %r" % (r, dres[:4]))

    def fence():
        torch.cuda.synchronize()
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    ctx.collect_profile()            # drop anything recorded so far
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.collect_profile()
    ctx.set_profiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # separate encode / decode wall rates (untimed extra pass, events on the library's stream)
    ctx.timer_start()
    r, used, _ = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, comps, chunks, frames, flags=flags)
    enc_ms = ctx.timer_stop()
    ctx.timer_start()
    for idx in range(len(fmts)):
        ctx.decode_frames(frames, used, idx, dec[idx])
    dec_ms = ctx.timer_stop()

    # the size-for-speed option (HAPGPU_ENCODE_COARSE_MATCHES), reported beside the default; never `value`
    coarse = None
    if world == 1 and hasattr(hap_amd, "ENCODE_COARSE_MATCHES"):
        cflags = flags | hap_amd.ENCODE_COARSE_MATCHES
        ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, comps, chunks, frames, flags=cflags)
        ctx.timer_start()
        r, cused, _ = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, comps, chunks, frames, flags=cflags)
        for idx in range(len(fmts)):
            ctx.decode_frames(frames, cused, idx, dec[idx])
        c_ms = ctx.timer_stop()
        coarse = {"rgba_GBps": round(nf * rgba_bytes / (c_ms * 1e-3) / 1e9, 2), "ms": round(c_ms, 3),
                  "snappy_ratio": round(sum(cused) / nf / sum(tex_bytes), 4),
                  "note": "encode+decode with 32-bit granular element streams for every format"}
        r, used, _ = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, comps, chunks, frames, flags=flags)

    # DXT -> RGBA, untimed extra: what a player without texture units needs after HapDecode
    rgba_out = torch.empty(w * h * 4, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    ctx.collect_profile()
    for i in range(min(nf, 16)):
        ctx.decompress_rgba(dec[0][i], fmts[0], w, h, rgba=rgba_out, alpha=(dec[1][i] if len(fmts) > 1 else None))
    prof_bd = ctx.collect_profile().get("block_decode", (0, 0.0))
    ctx.set_profiling(False)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_frames = nf * world * args.steps
    value = total_frames * rgba_bytes / elapsed / 1e9
    frame_bytes = sum(used) / nf
    ratio = frame_bytes / sum(tex_bytes)

    # ---- roofline of the dominant kernel: algorithmic bytes per launch / mean launch time ----
    blocks = (w // 4) * (h // 4)
    bsum = sum(tex_bytes)
    algo = {
        "block_encode": None,      # filled per launch below (one launch per frame per texture)
        "snappy_compress": nf * (bsum + frame_bytes),
        "frame_pack": None,
        "frame_gather": nf * 2 * frame_bytes,
        "decode_plan": None,
        "snappy_decode": nf * (frame_bytes * (tex_bytes[0] / bsum) + tex_bytes[0]) if len(fmts) == 1 else None,
    }
    # block encode: one launch per texture format over the whole batch
    algo["block_encode"] = nf * sum(blocks * (64 + BLOCK_BYTES[f]) for f in fmts) / len(fmts)
    if len(fmts) > 1:
        algo["snappy_decode"] = nf * (frame_bytes + bsum) / len(fmts)
    kernels = {}
    for name, (launches, ms) in prof.items():
        if launches:
            kernels[name] = {"launches": int(launches), "ms_total": round(ms, 4), "ms_avg": round(ms / launches, 5)}
            if algo.get(name):
                kernels[name]["algorithmic_GBps"] = round(algo[name] / (ms / launches * 1e-3) / 1e9, 1)
    dom = max((k for k in kernels if algo.get(k)), key=lambda k: kernels[k]["ms_total"])
    achieved = kernels[dom]["algorithmic_GBps"]
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(args.config, dom, nf),
                "algorithmic_bytes_per_launch": int(algo[dom]), "avg_launch_ms": kernels[dom]["ms_avg"]}

    if world > 1:
        args.no_cpu_baseline = True       # the CPU / foreign / host-pointer legs are reported at N=1 only
    host_path = None
    if not args.no_cpu_baseline:
        try:
            host_path = host_pointer_path(ctx, rgba, frames, used, fmts, comps, chunks, tex_bytes, cap, w, h, flags)
        except Exception as exc:
            host_path = {"error": repr(exc)}
    foreign = None
    if args.foreign_frames and not args.no_cpu_baseline:
        try:
            foreign = decode_foreign(ctx, dev, fmts, chunks, dec, tex_bytes, cap, min(args.foreign_frames, nf), rgba_bytes)
        except Exception as exc:
            foreign = {"error": repr(exc)}
    cpu = None
    if not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(w, h, fmts, chunks, rgba, dec, tex_bytes, cap, args.cpu_seconds)
        except Exception as exc:       # the baseline is reported, never required
            cpu = {"error": repr(exc)}

    line = {
        "metric": "RGBA GB/s + frames/sec, 8K Hap Q encode+decode" if args.config == "C4"
                  else "RGBA GB/s + frames/sec, %s encode+decode" % args.config,
        "value": round(value, 2), "unit": "GB/s", "fps": round(total_frames / elapsed, 1),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: %dx%d %s, %s chunks, Snappy, %d-frame batch per GPU, device-resident" % (
            args.config, w, h, "+".join("%#x" % f for f in fmts), "+".join(map(str, chunks)), nf),
            "frames_per_step_per_gpu": nf, "fragment_index": not args.no_fragment_index,
            "snappy_ratio": round(ratio, 4), "parallelism": "frame-shard x%d" % world},
        "encode_only": {"rgba_GBps": round(nf * rgba_bytes / (enc_ms * 1e-3) / 1e9, 2), "ms": round(enc_ms, 3)},
        "decode_only": {"rgba_GBps": round(nf * rgba_bytes / (dec_ms * 1e-3) / 1e9, 2), "ms": round(dec_ms, 3),
                        "texture_GBps": round(nf * bsum / (dec_ms * 1e-3) / 1e9, 2)},
        "texture_to_rgba": ({"us_per_frame": round(prof_bd[1] / prof_bd[0] * 1e3, 2),
                             "algorithmic_GBps": round((sum(tex_bytes) + rgba_bytes) / (prof_bd[1] / prof_bd[0] * 1e-3) / 1e9, 1)}
                            if prof_bd[0] else None),
        "coarse_matches_option": coarse,
        "decode_of_reference_encoded_frames": foreign,
        "host_pointer_path": host_path,
        "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def host_pointer_path(ctx, rgba, frames, used, fmts, comps, chunks, tex_bytes, cap, w, h, flags, n=4):
    """Same calls with pageable HOST buffers (what a plain hap.h client passes): PCIe-inclusive, never `value`."""
    import numpy as np
    n = min(n, len(rgba))
    host_rgba = [rgba[i].cpu().numpy() for i in range(n)]
    host_frames = [np.zeros(cap, dtype=np.uint8) for _ in range(n)]
    ctx.encode_frames_rgba(host_rgba, w, h, w * 4, fmts, comps, chunks, host_frames, flags=flags)   # warm: scratch, page faults
    t0 = time.perf_counter()
    r, hused, res = ctx.encode_frames_rgba(host_rgba, w, h, w * 4, fmts, comps, chunks, host_frames, flags=flags)
    t_enc = time.perf_counter() - t0
    if r != 0:
        raise RuntimeError("host encode failed %r" % res)
    outs = [np.zeros(tex_bytes[0], dtype=np.uint8) for _ in range(n)]
    ctx.decode_frames(host_frames, hused, 0, outs)
    t0 = time.perf_counter()
    r, dused, _f, dres = ctx.decode_frames(host_frames, hused, 0, outs)
    t_dec = time.perf_counter() - t0
    if r != 0:
        raise RuntimeError("host decode failed %r" % dres)
    rgba_bytes = w * h * 4
    res = {"frames": n, "encode_rgba_GBps": round(n * rgba_bytes / t_enc / 1e9, 2), "decode_rgba_GBps": round(n * rgba_bytes / t_dec / 1e9, 2),
           "encode_ms_per_frame": round(t_enc / n * 1e3, 2), "decode_ms_per_frame": round(t_dec / n * 1e3, 2),
           "note": "pageable host memory over PCIe, includes staging copies; warm (second call)"}
    # the same with page-locked host buffers (hipHostMalloc'ed by torch): the library's copies become real DMA
    try:
        pin_rgba = [rgba[i].cpu().pin_memory() for i in range(n)]
        pin_frames = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(n)]
        pin_outs = [torch.empty(tex_bytes[0], dtype=torch.uint8).pin_memory() for _ in range(n)]
        ctx.encode_frames_rgba(pin_rgba, w, h, w * 4, fmts, comps, chunks, pin_frames, flags=flags)
        t0 = time.perf_counter()
        r, pused, _res = ctx.encode_frames_rgba(pin_rgba, w, h, w * 4, fmts, comps, chunks, pin_frames, flags=flags)
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        r2, _du, _f, _dr = ctx.decode_frames(pin_frames, pused, 0, pin_outs)
        t_dec = time.perf_counter() - t0
        if r == 0 and r2 == 0:
            res["pinned"] = {"encode_rgba_GBps": round(n * rgba_bytes / t_enc / 1e9, 2),
                             "decode_rgba_GBps": round(n * rgba_bytes / t_dec / 1e9, 2)}
    except Exception as exc:      # pinning is optional
        res["pinned"] = {"error": repr(exc)}
    return res


def decode_foreign(ctx, dev, fmts, chunks, dec, tex_bytes, cap, n, rgba_bytes):
    """Frames produced by the CPU reference encoder (libsnappy streams, no fragment table) decoded by
    the GPU: the generic one-wave-per-chunk path.  Reported beside the headline, never part of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import _libs as L
    ref = L.ref_lib()
    lib, prefix = (ref, "refbase") if ref is not None else (L.oracle_lib(), "oraclebase")
    enc = getattr(lib, prefix + "_encode"); enc.restype = C.c_double
    count = len(fmts)
    tex_host = [[dec[i][f].cpu().numpy() for i in range(count)] for f in range(n)]
    ptrs = (C.c_void_p * (n * count))(*[tex_host[f][i].ctypes.data for f in range(n) for i in range(count)])
    lens = (C.c_ulong * count)(*tex_bytes)
    cf = (C.c_uint * count)(*fmts); cc = (C.c_uint * count)(*([1] * count)); ck = (C.c_uint * count)(*chunks)
    out = np.zeros(cap * n, dtype=np.uint8)
    used = (C.c_ulong * n)()
    if enc(C.c_uint(count), ptrs, lens, cf, cc, ck, C.c_uint(n), out.ctypes.data_as(C.c_void_p), C.c_ulong(cap), used,
           C.c_uint(n), C.c_uint(1)) < 0:
        raise RuntimeError("reference encode failed")
    frames = [torch.from_numpy(out[i * cap: i * cap + used[i]].copy()).to(dev) for i in range(n)]
    outs = [torch.empty(tex_bytes[0], dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize()
    ctx.decode_frames(frames, [int(u) for u in used], 0, outs)          # warm-up
    ctx.timer_start()
    r, dused, _f, dres = ctx.decode_frames(frames, [int(u) for u in used], 0, outs)
    ms = ctx.timer_stop()
    ok = r == 0 and all(torch.equal(outs[i], dec[0][i]) for i in range(n))
    return {"frames": n, "ms": round(ms, 3), "rgba_GBps": round(n * rgba_bytes / (ms * 1e-3) / 1e9, 2),
            "texture_GBps": round(n * tex_bytes[0] / (ms * 1e-3) / 1e9, 2), "bit_exact": bool(ok),
            "encoder": "reference hap.c + libsnappy 1.1.8" if ref is not None else "oracle/ C port"}


def measured_traffic(config, kernel, frames):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass (FETCH_SIZE and
    WRITE_SIZE collected in separate runs by tools/prof_traffic.sh; FETCH doubled as the MI355X
    guide prescribes for wide streaming reads).  None when no measurement exists for this config."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % config.lower())
    try:
        with open(path) as f:
            t = json.load(f)
        k = t["kernels"][kernel]
    except (OSError, KeyError, ValueError):
        return None
    scale = 1.0 if k["per"] == "frame" else frames / float(t["frames_per_launch"])
    return int((2.0 * k["fetch"] + k["write"]) * 1024 * scale)


def cpu_baseline(w, h, fmts, chunks, rgba, dec, tex_bytes, cap, budget_s):
    """Same frames through the CPU path on this box's host cores; a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import _libs as L
    cores = os.cpu_count() or 1
    ref = L.ref_lib()
    lib, prefix, kind = (ref, "refbase", "reference") if ref is not None else (L.oracle_lib(), "oraclebase", "port")
    enc = getattr(lib, prefix + "_encode"); enc.restype = C.c_double
    decf = getattr(lib, prefix + "_decode"); decf.restype = C.c_double
    ora = L.oracle_lib()
    ora.oraclebase_bc_encode.restype = C.c_double

    sample = min(len(rgba), 4)                 # distinct frames copied to the host
    threads = min(cores, 256)
    nwork = max(sample, min(threads, 128))     # frames in flight: pointers cycle over the sample
    count = len(fmts)
    tex_host = [[dec[i][f].cpu().numpy() for i in range(count)] for f in range(sample)]
    rgba_host = rgba[0].cpu().numpy()
    # block encode (oracle's scalar C, the reference has none): rows of one frame over all threads
    out = np.zeros(tex_bytes[0], dtype=np.uint8)
    t_bc = 0.0
    for fmt in fmts:
        t_bc += ora.oraclebase_bc_encode(rgba_host.ctypes.data_as(C.c_void_p), C.c_uint(w), C.c_uint(h),
                                         C.c_size_t(w * 4), C.c_uint(fmt), out.ctypes.data_as(C.c_void_p),
                                         C.c_uint(threads), C.c_uint(1))
    # container + Snappy: one frame per thread (the reference's encode is serial per frame)
    ptrs = (C.c_void_p * (nwork * count))(*[tex_host[f % sample][i].ctypes.data for f in range(nwork) for i in range(count)])
    lens = (C.c_ulong * count)(*tex_bytes)
    cf = (C.c_uint * count)(*fmts); cc = (C.c_uint * count)(*([1] * count)); ck = (C.c_uint * count)(*chunks)
    enc_threads = min(threads, nwork)
    outbuf = np.zeros(cap * enc_threads, dtype=np.uint8)
    used = (C.c_ulong * nwork)()
    t_enc = enc(C.c_uint(count), ptrs, lens, cf, cc, ck, C.c_uint(nwork), outbuf.ctypes.data_as(C.c_void_p),
                C.c_ulong(cap), used, C.c_uint(enc_threads), C.c_uint(1))
    if t_enc < 0:
        raise RuntimeError("cpu encode failed %r" % t_enc)
    frames = []
    for f in range(sample):
        one = np.zeros(cap, dtype=np.uint8)
        u = (C.c_ulong * 1)()
        p1 = (C.c_void_p * count)(*[tex_host[f][i].ctypes.data for i in range(count)])
        enc(C.c_uint(count), p1, lens, cf, cc, ck, C.c_uint(1), one.ctypes.data_as(C.c_void_p), C.c_ulong(cap), u,
            C.c_uint(1), C.c_uint(1))
        frames.append(one[: u[0]].copy())
    fptrs = (C.c_void_p * nwork)(*[frames[f % sample].ctypes.data for f in range(nwork)])
    flens = (C.c_ulong * nwork)(*[len(frames[f % sample]) for f in range(nwork)])
    decp = getattr(lib, prefix + "_decode_parallel"); decp.restype = C.c_double
    stride = max(tex_bytes)
    dout = np.zeros(stride * enc_threads, dtype=np.uint8)
    t_dec = 0.0
    for idx in range(count):
        t = decp(fptrs, flens, C.c_uint(nwork), C.c_uint(idx), dout.ctypes.data_as(C.c_void_p),
                 C.c_ulong(stride), C.c_uint(enc_threads), C.c_uint(1))
        if t < 0:
            raise RuntimeError("cpu decode failed %r" % t)
        t_dec += t
    t_enc /= nwork
    t_dec /= nwork
    # one hardware thread, one frame (the reference as a client would call it serially)
    one_out = np.zeros(cap, dtype=np.uint8)
    one_used = (C.c_ulong * 1)()
    p1 = (C.c_void_p * count)(*[tex_host[0][i].ctypes.data for i in range(count)])
    t_enc1 = enc(C.c_uint(count), p1, lens, cf, cc, ck, C.c_uint(1), one_out.ctypes.data_as(C.c_void_p), C.c_ulong(cap),
                 one_used, C.c_uint(1), C.c_uint(1))
    f1 = (C.c_void_p * 1)(frames[0].ctypes.data)
    l1 = (C.c_ulong * 1)(len(frames[0]))
    t_dec1 = sum(decp(f1, l1, C.c_uint(1), C.c_uint(idx), dout.ctypes.data_as(C.c_void_p), C.c_ulong(stride),
                      C.c_uint(1), C.c_uint(1)) for idx in range(count))
    sample_n = sample
    sample = 1      # t_enc / t_dec are already per frame
    rgba_bytes = w * h * 4
    per_frame = t_bc + t_enc / sample + t_dec / sample
    return {"value": round(rgba_bytes / per_frame / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": kind,
            "threads": {"block_encode": threads, "hap_encode": enc_threads, "hap_decode": enc_threads},
            "sample": "%d frames in flight (%d distinct) of this workload: RGBA->DXT by oracle/bc_oracle.c (the reference "
                      "has no block encoder) + HapEncode + HapDecode by %s, %d threads, amortised per frame" % (
                          nwork, sample_n, "unmodified reference hap.c + libsnappy 1.1.8" if kind == "reference" else "oracle/ C port", threads),
            "ms_per_frame": {"block_encode": round(t_bc * 1e3, 2), "hap_encode": round(t_enc / sample * 1e3, 2),
                             "hap_decode": round(t_dec / sample * 1e3, 2)},
            "container_only_rgba_GBps": round(rgba_bytes / (t_enc / sample + t_dec / sample) / 1e9, 3),
            "single_thread_ms_per_frame": {"hap_encode": round(t_enc1 * 1e3, 2), "hap_decode": round(t_dec1 * 1e3, 2)}}


if __name__ == "__main__":
    main()
