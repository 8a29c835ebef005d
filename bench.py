#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: RGBA GB/s (+ frames/s), 8K Hap Q
encode + decode, batched 60-frame stream (configs[3] / "C4": 7680x4320, scaled YCoCg-DXT5,
24 chunks, Snappy), frames resident in HBM when the timed region starts.

One "step" = one pass of the hot path over one 60-frame batch:
    RGBA (HBM) -> [block_encode] -> YCoCg-DXT5 texture -> [snappy_compress, frame_pack,
    frame_gather] -> Hap Q frame (HBM) -> [decode_plan, snappy_decode] -> texture (HBM)
value = frames * W*H*4 bytes / seconds for encode+decode together (decimal GB/s).

Multi-GPU: one process per GPU (`python bench.py --gpus N` starts the N ranks itself when no
launcher did), frames are independent: the 60-frame stream is split f -> GPU f mod N (SURVEY 8e,
"strong" scaling, 8/8/8/8/7/7/7/7 at N = 8) with no data-path collective; the only
torch.distributed traffic of the headline is the barrier and the MAX-reduce of the elapsed time.
At N > 1 the line also carries the weak-scaling figure (a whole stream per GPU) and the C5
chunk-group split of one 16K frame without and with the RCCL gather.  At N = 1 it carries a "c5"
object: the north-star's 16K Hap Q Alpha target config timed the same way.

Extra objects on the JSON line: "roofline" for the dominant kernel (HIP events recorded on the
library's own stream around every launch of the timed region) and "cpu_baseline" (the
unmodified reference hap.c + libsnappy from oracle/_ref when present, else the C port in
oracle/, plus the oracle's scalar block encoder -- the reference has no RGBA stage; medians of
>= 10 repetitions).  Beside them, never part of `value`: "c5" / "c2" / "c3" (the other BASELINE
configs), "c1" (configs[0]: one 1080p DXT1 frame through plain hap.h, one call, next to the
reference on one CPU thread), "per_call_hap_h" (one hap.h call per 8K frame from a C client),
"bc7_opaque" (an opaque format), frames made by the reference encoder, host-pointer rates.
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
    "C1": (1920, 1080, [0x83F0], [1], 1),
    "C2": (3840, 2160, [0x83F0], [1], 60),
    "C3": (3840, 2160, [0x83F3], [8], 60),
    "C4": (7680, 4320, [0x01], [24], 60),
    "C5": (16384, 16384, [0x01, 0x8DBB], [64, 64], 4),
}
BLOCK_BYTES = {0x83F0: 8, 0x8DBB: 8, 0x83F3: 16, 0x01: 16}
ROUND_TAG = "r06"        # profiles/<round>_traffic_<cfg>.json is what roofline.traffic quotes


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C4", choices=sorted(CONFIGS))
    ap.add_argument("--frames", type=int, default=0, help="frames of the stream (default: the config's)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the stream is split over the ranks, frame f -> rank f mod N (SURVEY 8e); "
                         "weak: every rank gets a whole stream of its own")
    ap.add_argument("--no-fragment-index", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="the step as one blocking encode call + one blocking decode call on one context (rounds 1-4); default: "
                         "pipelined -- batch k + 1's encode is launched (HapGpuEncodeFramesRGBABegin) before batch k is decoded "
                         "on a second context, so the GPU never waits for the host between calls")
    ap.add_argument("--frag-log2", type=int, default=0, help="Snappy fragment size (log2 bytes); 0 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--foreign-frames", type=int, default=24,
                    help="also time decoding of N frames made by the CPU reference encoder (no fragment table); 0 = skip")
    ap.add_argument("--c5-frames", type=int, default=4,
                    help="also time N frames of C5 (16K Hap Q Alpha, the north-star's target config) at N=1; 0 = skip")
    ap.add_argument("--no-extras", action="store_true",
                    help="the timed pipeline only: no size-for-speed option, texture->RGBA, host-pointer, foreign-frame, CPU "
                         "or C5 legs (what tools/prof_bench.sh profiles, so that every dispatch belongs to the headline)")
    ap.add_argument("--extras-deadline", type=int, default=240,
                    help="N > 1: seconds the legs beside the headline (other scaling mode, chunk groups with the RCCL gather) may take "
                         "before rank 0 prints the line without them")
    ap.add_argument("--one-gpu-ranks", action="store_true",
                    help="dry run of the multi-rank path on ONE GPU (tests): every rank uses device 0 and the collectives "
                         "run over gloo (host copies) instead of RCCL; the codec is the real one.  Never a measurement")
    ap.add_argument("--devices-from-c", type=int, default=0, metavar="N",
                    help="the stream dealt out over N contexts on min(N, visible) GPUs by ONE process through the C entry points "
                         "HapGpuEncodeFramesRGBAOnDevices / HapGpuDecodeFramesOnDevices (frame f -> context f mod N, a host thread "
                         "per context, no collective): the multi-GPU road of a C client, same line shape as --gpus N")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launch / rank / reduction logic only, gloo on CPU, a sleep in place of the codec (tests)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of ourselves, one per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def frames_of_rank(total, rank, world, scaling):
    """Frame numbers (seeds of the synthetic stream) this rank works on."""
    if scaling == "weak":
        return list(range(rank * total, (rank + 1) * total))
    return list(range(rank, total, world))          # hap_amd.shard.frames_for_rank: 60 over 8 -> 8/8/8/8/7/7/7/7


class Stream:
    """One rank's share of a synthetic stream with every buffer resident in HBM, and the timed step over it."""

    def __init__(self, hap_amd, ctx, dev, config, frame_ids, flags, ctx_dec=None, ctx_enc2=None):
        from hap_amd import synth
        self.hap, self.ctx = hap_amd, ctx
        # pipelined steps: the decode calls go to a context of their own (own stream, own scratch), and the frames of
        # consecutive batches to alternating buffers -- batch k + 1 is being written while batch k is read
        self.ctx_dec = ctx_dec
        # a second ENCODE context (optional): batches alternate between the two, so that one batch's encode kernel starts
        # while the other's is in its tail -- what small batches lose to the last, partly filled round of workgroups
        self.ctx_enc2 = ctx_enc2
        self.w, self.h, self.fmts, self.chunks, _n = CONFIGS[config]
        self.nf = len(frame_ids)
        self.flags = flags
        w, h = self.w, self.h
        self.tex_bytes = [(w // 4) * (h // 4) * BLOCK_BYTES[f] for f in self.fmts]
        self.cap = hap_amd.HapMaxEncodedLength(self.tex_bytes, self.fmts, self.chunks)
        self.rgba_bytes = w * h * 4
        self.comps = [1] * len(self.fmts)
        # the buffers stay where they are for the whole run: resolve their addresses once, as a C client would
        self.rgba = hap_amd.BufferList([synth.rgba_frame(w, h, i, device=dev) for i in frame_ids])
        self.frames = hap_amd.BufferList([torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in frame_ids])
        self.frames_b = hap_amd.BufferList([torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in frame_ids]) \
            if ctx_dec is not None else None
        self.frames_c = hap_amd.BufferList([torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in frame_ids]) \
            if ctx_enc2 is not None else None
        self.dec = [hap_amd.BufferList([torch.empty(tb, dtype=torch.uint8, device=dev) for _ in frame_ids])
                    for tb in self.tex_bytes]
        # entry f * T + t = texture t of frame f: the order HapGpuDecodeFrameTextures takes
        self.dec_all = hap_amd.BufferList([self.dec[t][f] for f in range(self.nf) for t in range(len(self.fmts))])
        self.used = None
        torch.cuda.synchronize()

    def encode(self, flags=None):
        r, used, results = self.ctx.encode_frames_rgba(self.rgba, self.w, self.h, self.w * 4, self.fmts, self.comps,
                                                       self.chunks, self.frames, flags=self.flags if flags is None else flags)
        if r != 0:
            raise RuntimeError("encode failed: %r %r" % (r, results[:4]))
        return used

    def decode(self, used, ctx=None, frames=None):
        ctx = ctx or self.ctx
        frames = frames if frames is not None else self.frames
        if len(self.fmts) > 1:
            # every texture of every frame in one batch (HapGpuDecodeFrameTextures)
            nt = len(self.fmts)
            r, dused, _dfmts, dres = ctx.decode_frame_textures(frames, used, nt, self.dec_all)
            if r != 0 or dused[:nt] != [self.tex_bytes[t] for t in range(nt)]:
                raise RuntimeError("decode failed: %r %r" % (r, dres[:4]))
            return
        r, dused, _dfmts, dres = ctx.decode_frames(frames, used, 0, self.dec[0])
        if r != 0 or dused[0] != self.tex_bytes[0]:
            raise RuntimeError("decode failed: %r %r" % (r, dres[:4]))

    def begin(self, frames, ctx=None):
        r = (ctx or self.ctx).encode_frames_rgba_begin(self.rgba, self.w, self.h, self.w * 4, self.fmts, self.comps, self.chunks, frames,
                                                       flags=self.flags)
        if r != 0:
            raise RuntimeError("encode (first half) failed: %r" % r)

    def finish(self, ctx=None):
        r, used, results = (ctx or self.ctx).encode_finish()
        if r != 0:
            raise RuntimeError("encode failed: %r %r" % (r, results[:4]))
        return used

    def step(self):
        if self.nf == 0:
            return
        self.used = self.encode()
        self.decode(self.used)

    def timed(self, steps, warmup, fence, pipelined=False):
        """K steps between fences; returns (seconds, per-class HIP-event profile of the timed region).
        pipelined: K encodes and K decodes all the same, every decode reads the frames its batch's encode wrote -- but
        batch k + 1's encode is on the GPU's queue before the host turns to batch k's decode (its header read-back,
        its plan, its launches, its completion), and the decode runs on a second context."""
        for _ in range(warmup):
            self.step()
        if pipelined and self.nf:
            self.decode(self.used, self.ctx_dec)                # the decode context's scratch and code, untimed
        ctxs = [self.ctx] + ([self.ctx_dec] if pipelined else []) + ([self.ctx_enc2] if pipelined and self.ctx_enc2 is not None else [])
        for c in ctxs:
            c.set_profiling(True)
            c.collect_profile()            # drop anything recorded so far
        fence()
        t0 = time.perf_counter()
        if not pipelined:
            for _ in range(steps):
                self.step()
        elif self.nf and self.ctx_enc2 is not None:
            # two encode contexts: batches k and k + 1 are both on the GPU's queues while batch k - 1 is decoded
            sets = [self.frames, self.frames_b, self.frames_c]
            encs = [self.ctx, self.ctx_enc2]
            for k in range(min(2, steps)):
                self.begin(sets[k % 3], encs[k & 1])
            for k in range(steps):
                self.used = self.finish(encs[k & 1])
                if k + 2 < steps:
                    self.begin(sets[(k + 2) % 3], encs[k & 1])
                self.decode(self.used, self.ctx_dec, sets[k % 3])
            self.last_frames = sets[(steps - 1) % 3]
        elif self.nf:
            sets = [self.frames, self.frames_b]
            self.begin(sets[0])
            for k in range(steps):
                self.used = self.finish()
                if k + 1 < steps:
                    self.begin(sets[(k + 1) & 1])
                self.decode(self.used, self.ctx_dec, sets[k & 1])
            self.last_frames = sets[(steps - 1) & 1]
        for c in ctxs[1:]:
            c.synchronize()
        fence()
        elapsed = time.perf_counter() - t0
        prof = {}
        for c in ctxs:
            for name, (n, ms) in c.collect_profile().items():
                a = prof.get(name, (0, 0.0))
                prof[name] = (a[0] + n, a[1] + ms)
            c.set_profiling(False)
        return elapsed, prof

    def bit_exact(self, reference=False):
        """After a timed region: every texture the last step decoded equals what the block encoder makes of the same
        RGBA frame (the encoder itself is pinned to oracle/bc_oracle.c by the -m gpu tests at these sizes).
        reference: frame 0 as the last step wrote it is also decoded by the checker on the CPU -- the unmodified
        reference (oracle/_ref) when it is there, else its restatement -- and must give the same textures."""
        ok = True
        for idx, fmt in enumerate(self.fmts):
            want = torch.empty(self.tex_bytes[idx], dtype=torch.uint8, device=self.dec[idx][0].device)
            for i in range(self.nf):
                torch.cuda.synchronize()
                r = self.ctx.compress_rgba(self.rgba[i], self.w, self.h, self.w * 4, fmt, want)
                self.ctx.synchronize()
                ok = ok and r[0] == 0 and bool(torch.equal(want, self.dec[idx][i]))
        if reference and self.nf and self.used:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import _libs as L
                api = L.ref_api() or L.oracle_api()
                frames = getattr(self, "last_frames", None) or self.frames
                frame = frames[0][: self.used[0]].cpu().numpy()
                for idx, fmt in enumerate(self.fmts):
                    rc, out, f2 = api.decode_np(frame, idx, self.tex_bytes[idx])
                    ok = ok and rc == 0 and f2 == fmt and bool((out == self.dec[idx][0].cpu().numpy()).all())
                self.reference_checked = "reference hap.c" if L.ref_api() is not None else "oracle/ restatement"
            except Exception as exc:           # the checker is test infrastructure: its absence is reported, not fatal
                self.reference_checked = "unavailable: %r" % (exc,)
        return ok

    def split_rates(self):
        """separate encode / decode wall times of one untimed extra pass (events on the library's stream)"""
        self.ctx.timer_start()
        used = self.encode()
        enc_ms = self.ctx.timer_stop()
        self.ctx.timer_start()
        self.decode(used)
        dec_ms = self.ctx.timer_stop()
        return enc_ms, dec_ms

    def kernel_table(self, prof, steps, config):
        """Per kernel class: launches, time, algorithmic GB/s.  Algorithmic bytes per step follow SURVEY 8d:
        block encode 64 + sum(b) per block (the RGBA is counted once however many launches read it), Snappy compress
        b(1 + c), decode b(1 + c), gather 2 c b."""
        nf = self.nf
        blocks = (self.w // 4) * (self.h // 4)
        bsum = sum(self.tex_bytes)
        frame_bytes = sum(self.used) / max(nf, 1)
        per_step = {"block_encode": nf * blocks * (64 + sum(BLOCK_BYTES[f] for f in self.fmts)),
                    # RGBA -> compressed fragments in one kernel: SURVEY 8d "fused encode", 64 + c b per block
                    "encode_fused": nf * (blocks * 64 + frame_bytes),
                    "snappy_compress": nf * (bsum + frame_bytes),
                    "frame_gather": nf * 2 * frame_bytes,
                    "snappy_decode": nf * (frame_bytes + bsum)}
        kernels = {}
        for name, (launches, ms) in prof.items():
            if not launches:
                continue
            k = {"launches": int(launches), "ms_total": round(ms, 4), "ms_avg": round(ms / launches, 5)}
            if name == "frame_gather" and per_step[name] * steps / (ms * 1e-3) / 1e9 > 1.5 * HBM_PEAK_GBPS:
                # (batches of a dozen single-texture frames and more: the compressor wrote the fragments where they
                # belong, the gather pass moves the group tables only)
                k["note"] = "group tables only: the fragments were placed by the compressor"
            elif name in per_step:
                k["algorithmic_bytes_per_launch"] = int(per_step[name] * steps / launches)
                k["algorithmic_GBps"] = round(per_step[name] * steps / (ms * 1e-3) / 1e9, 1)
            kernels[name] = k
        return kernels, frame_bytes / bsum

    def roofline(self, kernels, config, kernel=None):
        cands = [k for k in kernels if "algorithmic_GBps" in kernels[k]]
        dom = kernel or max(cands, key=lambda k: kernels[k]["ms_total"])
        achieved = kernels[dom]["algorithmic_GBps"]
        traffic, source = measured_traffic(config, dom, self.nf)
        r = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": source,
             "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"],
             "avg_launch_ms": kernels[dom]["ms_avg"]}
        # what these kernels actually sit on: vector instruction issue.  SQ_INSTS_VALU (the round's PMC pass over this
        # command) x 4 cycles / (1024 SIMDs x 2.4 GHz) = the time the launch's vector instructions take issued back to
        # back on every SIMD of the chip; as a share of the measured launch time
        valu = measured_valu(config, dom, self.nf)
        if valu and kernels[dom]["ms_avg"]:
            r["issue_bound"] = round(valu * 4.0 / (1024 * 2.4e9) / (kernels[dom]["ms_avg"] * 1e-3), 4)
            r["issue_bound_note"] = ("SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz) / avg_launch_ms; above 1: the kernel's mix issues "
                                     "faster than 4 cycles an instruction (tools/micro/valu_rates.hip: add / and / mov 2.7, the rest 3.4-4.3; "
                                     "3.6 on average -> issue_bound_at_3p6)")
            r["issue_bound_at_3p6"] = round(valu * 3.6 / (1024 * 2.4e9) / (kernels[dom]["ms_avg"] * 1e-3), 4)
        return r


def devices_from_c(args):
    """One process, N contexts: frame f -> context f mod N (hap_devices.c).  The timed step is one
    HapGpuEncodeFramesRGBAOnDevices call + one HapGpuDecodeFramesOnDevices call over the whole stream; `value` counts every
    frame of every step, the clock is this process's (the calls return when every context has finished)."""
    import time
    import hap_amd
    from hap_amd import synth
    from hap_amd.api import encode_frames_rgba_on_devices, decode_frames_on_devices
    w, h, fmts, chunks, nf_default = CONFIGS[args.config]
    nf = args.frames or nf_default
    n_ctx = args.devices_from_c
    visible = torch.cuda.device_count()
    devices = [c % visible for c in range(n_ctx)]
    ctxs = [hap_amd.Context(d) for d in devices]
    flags = 0 if args.no_fragment_index else hap_amd.ENCODE_FRAGMENT_INDEX
    tex_bytes = [(w // 4) * (h // 4) * (8 if f in (0x83F0, 0x8DBB) else 16) for f in fmts]
    cap = hap_amd.HapMaxEncodedLength(tex_bytes, fmts, chunks) + (1 << 20)
    rgba, frames, decs = [], [], []
    for f in range(nf):
        dev = torch.device("cuda", devices[f % n_ctx])
        rgba.append(synth.rgba_frame(w, h, f, device=dev))
        frames.append(torch.zeros(cap, dtype=torch.uint8, device=dev))
        decs.append([torch.zeros(b, dtype=torch.uint8, device=dev) for b in tex_bytes])
    for d in range(visible):
        torch.cuda.synchronize(d)

    def step():
        r, used, res = encode_frames_rgba_on_devices(ctxs, rgba, w, h, w * 4, fmts, [1] * len(fmts), chunks, frames, flags=flags)
        assert r == 0 and res == [0] * nf, (r, res)
        for t in range(len(fmts)):
            r, du, df, dr = decode_frames_on_devices(ctxs, frames, used, t, [d[t] for d in decs])
            assert r == 0 and dr == [0] * nf, (r, dr)
        return used

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        used = step()
    elapsed = time.perf_counter() - t0
    # the last step's textures against the block encoder's (each on its own device)
    ok = True
    for f in range(0, nf, max(1, nf // 8)):
        for t, fmt in enumerate(fmts):
            want = torch.zeros(tex_bytes[t], dtype=torch.uint8, device=rgba[f].device)
            torch.cuda.synchronize(rgba[f].device)
            assert ctxs[f % n_ctx].compress_rgba(rgba[f], w, h, w * 4, fmt, want) == (0, tex_bytes[t])
            ok = ok and bool(torch.equal(want, decs[f][t]))
    rgba_bytes = w * h * 4
    total = nf * args.steps
    line = {"metric": "RGBA GB/s + frames/sec, 8K Hap Q encode+decode" if args.config == "C4"
                      else "RGBA GB/s + frames/sec, %s encode+decode" % args.config,
            "value": round(total * rgba_bytes / elapsed / 1e9, 2), "unit": "GB/s", "fps": round(total / elapsed, 1),
            "n_gpus": min(n_ctx, visible), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "step": "one HapGpuEncodeFramesRGBAOnDevices call + one HapGpuDecodeFramesOnDevices call per texture over the whole stream "
                    "(blocking: the calls return when every context has finished)",
            "config": {"workload": "%s: %dx%d %s, %s chunks, Snappy, %d-frame stream, device-resident" % (
                           args.config, w, h, "+".join("%#x" % f for f in fmts), "+".join(map(str, chunks)), nf),
                       "frames_per_step": nf, "contexts": n_ctx, "devices": sorted(set(devices)), "fragment_index": not args.no_fragment_index,
                       "snappy_ratio": round(sum(used) / nf / sum(tex_bytes), 4),
                       "parallelism": "one process: frame f -> context f mod %d (hap_devices.c), a host thread per context, "
                                      "no data-path collective" % n_ctx},
            "bit_exact": ok, "roofline": None, "cpu_baseline": None}
    if min(n_ctx, visible) < n_ctx:
        line["dry_run"] = "%d contexts share %d GPU(s): a functional check of the C road, not a scaling measurement" % (n_ctx, visible)
    print(json.dumps(line))
    sys.stdout.flush()


def main():
    args = parse_args()
    if args.devices_from_c:
        return devices_from_c(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_under_torchrun(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if "RANK" in os.environ:                                      # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if args.one_gpu_ranks:
            local_rank = 0
        if not args.selftest_cpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if (args.selftest_cpu or args.one_gpu_ranks) else "nccl", rank=rank, world_size=world)
        world = dist.get_world_size()
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) were started" % (args.gpus, world))
    if args.selftest_cpu:
        return selftest_cpu(args, dist, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import hap_amd

    nf_total = args.frames or CONFIGS[args.config][4]
    ctx = hap_amd.Context(local_rank)
    if args.frag_log2:
        ctx.set_fragment_log2(args.frag_log2)
    flags = 0 if args.no_fragment_index else hap_amd.ENCODE_FRAGMENT_INDEX

    def fence():
        torch.cuda.synchronize()
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    pipelined = not args.serial
    ctx_dec = hap_amd.Context(local_rank) if pipelined else None
    stream = Stream(hap_amd, ctx, dev, args.config, frames_of_rank(nf_total, rank, world, args.scaling), flags, ctx_dec=ctx_dec)
    elapsed, prof = stream.timed(args.steps, args.warmup, fence, pipelined=pipelined)
    elapsed = max_over_ranks(elapsed)
    serial = None
    timed_bit_exact = stream.bit_exact(reference=(rank == 0))     # what the timed region's last step left, before anything else runs
    if pipelined:
        # the same K steps as blocking calls on one context: what rounds 1-4 timed, and the region the per-kernel events
        # are taken from -- in the pipelined region the encode kernel of batch k + 1 and the decode kernels of batch k
        # share the GPU, and an event pair around one of them also times its share of the other
        pipelined_prof = prof
        e2, prof = stream.timed(args.steps, 1, fence)
        serial = max_over_ranks(e2)
    total_frames = (nf_total if args.scaling == "strong" else nf_total * world) * args.steps
    rgba_bytes = stream.rgba_bytes
    value = total_frames * rgba_bytes / elapsed / 1e9

    other = None
    watchdog = None
    if world > 1:
        # What follows is reported beside the headline, never required for it -- and it is the only part of this file
        # with collectives on the data path (the optional RCCL gather of the chunk-group leg).  The headline is safe
        # from it: an exception in a leg is reported in its place, and if a leg has not come back after
        # --extras-deadline seconds (a rank that failed alone leaves the others waiting in a collective) rank 0 prints
        # the line as it stands and every rank leaves.
        import threading
        core = {"metric": "RGBA GB/s + frames/sec, 8K Hap Q encode+decode" if args.config == "C4"
                          else "RGBA GB/s + frames/sec, %s encode+decode" % args.config,
                "value": round(value, 2), "unit": "GB/s", "fps": round(total_frames / elapsed, 1), "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "%s, %d-frame stream, device-resident" % (args.config, nf_total),
                           "parallelism": "frame f -> GPU f mod %d, no data-path collective" % world},
                "bit_exact": bool(timed_bit_exact),
                "extras": "not finished within %d s: the line was printed without them" % args.extras_deadline}

        def give_up():
            if rank == 0:
                print(json.dumps(core))
                sys.stdout.flush()
            os._exit(0)

        watchdog = threading.Timer(args.extras_deadline, give_up)
        watchdog.daemon = True
        watchdog.start()
        groups = None
        try:
            # the other scaling mode beside the headline (same kernels, same step; a second timed region)
            mode = "weak" if args.scaling == "strong" else "strong"
            del stream.rgba, stream.frames, stream.dec, stream.dec_all
            torch.cuda.empty_cache()
            del stream.frames_b
            s2 = Stream(hap_amd, ctx, dev, args.config, frames_of_rank(nf_total, rank, world, mode), flags, ctx_dec=ctx_dec)
            e2, _p2 = s2.timed(args.steps, 1, fence, pipelined=pipelined)
            e2 = max_over_ranks(e2)
            f2 = (nf_total if mode == "strong" else nf_total * world) * args.steps
            other = {"scaling": mode, "value": round(f2 * rgba_bytes / e2 / 1e9, 2), "unit": "GB/s", "fps": round(f2 / e2, 1),
                     "ms_per_step": round(e2 / args.steps * 1e3, 3), "frames_per_step": f2 // args.steps}
            del s2
            torch.cuda.empty_cache()
        except Exception as exc:
            other = {"error": repr(exc)}
        try:
            groups = c5_chunk_groups(hap_amd, ctx, dist, dev, rank, world, fence)
        except Exception as exc:
            groups = {"error": repr(exc)}
    else:
        groups = None

    if rank != 0:
        if dist is not None:
            dist.barrier()
            if watchdog is not None:
                watchdog.cancel()
            dist.destroy_process_group()
        return

    line = {
        "metric": "RGBA GB/s + frames/sec, 8K Hap Q encode+decode" if args.config == "C4"
                  else "RGBA GB/s + frames/sec, %s encode+decode" % args.config,
        "value": round(value, 2), "unit": "GB/s", "fps": round(total_frames / elapsed, 1),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "step": ("pipelined: K encodes + K decodes inside the timed region, batch k + 1's encode launched (HapGpuEncodeFramesRGBABegin) "
                 "before batch k is decoded on a second context; frames in alternating buffers") if pipelined
                else "serial: one blocking encode call + one blocking decode call per step, one context",
    }
    total_frames_per_step = total_frames // args.steps
    if serial is not None:
        line["serial_step"] = {"ms_per_step": round(serial / args.steps * 1e3, 3),
                               "value": round(total_frames * rgba_bytes / serial / 1e9, 2), "unit": "GB/s",
                               "note": "the same K steps as blocking calls on one context (the step of rounds 1-4); the per-kernel "
                                       "events and the roofline below are this region's"}
    w, h, fmts, chunks = stream.w, stream.h, stream.fmts, stream.chunks
    if world == 1:
        kernels, ratio = stream.kernel_table(prof, args.steps, args.config)
        enc_ms, dec_ms = stream.split_rates()
        nf, bsum = stream.nf, sum(stream.tex_bytes)
        line["config"] = {"workload": "%s: %dx%d %s, %s chunks, Snappy, %d-frame stream, device-resident" % (
            args.config, w, h, "+".join("%#x" % f for f in fmts), "+".join(map(str, chunks)), nf_total),
            "frames_per_step": nf_total, "fragment_index": not args.no_fragment_index,
            "snappy_ratio": round(ratio, 4), "parallelism": "single GPU"}
        line["encode_only"] = {"rgba_GBps": round(nf * rgba_bytes / (enc_ms * 1e-3) / 1e9, 2), "ms": round(enc_ms, 3)}
        line["decode_only"] = {"rgba_GBps": round(nf * rgba_bytes / (dec_ms * 1e-3) / 1e9, 2), "ms": round(dec_ms, 3),
                               "texture_GBps": round(nf * bsum / (dec_ms * 1e-3) / 1e9, 2)}
        if args.no_extras:
            args.no_cpu_baseline, args.c5_frames = True, 0
        line["texture_to_rgba"] = None if args.no_extras else texture_to_rgba(stream, dev)
        line["coarse_matches_option"] = None if args.no_extras else coarse_option(stream, hap_amd)
        line["smaller_files_option"] = None if args.no_extras else smaller_option(stream, hap_amd)
        if not args.no_extras:
            try:
                line["separate_passes"] = separate_passes(hap_amd, stream, local_rank, args.config, fence)
            except Exception as exc:
                line["separate_passes"] = {"error": repr(exc)}
        stream.used = stream.encode()
        stream.decode(stream.used)
        line["bit_exact"] = bool(timed_bit_exact and stream.bit_exact())
        line["bit_exact_checked"] = ("every texture the timed region's last step decoded == the block encoder's; frame 0 of that step "
                                     "decoded by: %s" % getattr(stream, "reference_checked", "-"))
        if not args.no_extras:
            try:
                line["frames_to_rgba"] = frames_to_rgba(hap_amd, stream, dev)
            except Exception as exc:
                line["frames_to_rgba"] = {"error": repr(exc)}
        extras = {}
        if not args.no_cpu_baseline:
            for name, fn in (("host_pointer_path", lambda: host_pointer_path(ctx, stream.rgba, stream.frames, stream.used, fmts,
                                                                                stream.comps, chunks, stream.tex_bytes, stream.cap, w, h, flags)),
                             ("decode_of_reference_encoded_frames",
                              lambda: decode_foreign(ctx, dev, fmts, chunks, stream.dec, stream.tex_bytes, stream.cap,
                                                     min(args.foreign_frames, nf), rgba_bytes) if args.foreign_frames else None),
                             ("cpu_baseline", lambda: cpu_baseline(w, h, fmts, chunks, stream.rgba, stream.dec, stream.tex_bytes,
                                                                    stream.cap, args.cpu_seconds))):
                try:
                    extras[name] = fn()
                except Exception as exc:       # these legs are reported, never required
                    extras[name] = {"error": repr(exc)}
        line["decode_of_reference_encoded_frames"] = extras.get("decode_of_reference_encoded_frames")
        line["host_pointer_path"] = extras.get("host_pointer_path")
        line["roofline"] = stream.roofline(kernels, args.config)
        line["roofline"]["events_from"] = ("the serial region of the same K steps (each kernel runs by itself there; in the pipelined region the "
                                           "encode kernel of one batch and the decode kernels of the other share the GPU)") if serial is not None \
            else "the timed region"
        line["roofline"]["profiled_as"] = "python bench.py --config %s --steps 20 --warmup 5 --frames %d --no-extras --serial (tools/prof_bench.sh -> profiles/%s_kernel_stats_%s.csv)" % (
            args.config, nf_total, ROUND_TAG, args.config.lower())
        line["kernels"] = kernels
        if not args.no_extras:
            ms_full = elapsed / args.steps * 1e3 if pipelined else None
            ms_full_serial = (serial if serial is not None else elapsed) / args.steps * 1e3
            for name, fn in (("small_batch", lambda: small_batch(hap_amd, ctx, ctx_dec, dev, args.config, flags, fence, ms_full,
                                                                   ms_full_serial, nf_total)),
                             ("plain_frames_batched", lambda: plain_frames_batched(hap_amd, ctx, dev, args.config, nf, fence)),
                             ("fine_chunks_option", lambda: fine_chunks_option(hap_amd, ctx, dev, args.config, nf, fence))):
                try:
                    line[name] = fn()
                except Exception as exc:
                    line[name] = {"error": repr(exc)}
        if serial is not None:
            line["kernels_in_pipelined_region_ms_avg"] = {k: round(ms / n, 5) for k, (n, ms) in pipelined_prof.items() if n}
        line["cpu_baseline"] = extras.get("cpu_baseline")
        if not args.no_extras:
            try:
                stream.used = stream.encode()
                stream.decode(stream.used)
                line["per_call_hap_h"] = per_call_object(hap_amd, stream, dev)
            except Exception as exc:
                line["per_call_hap_h"] = {"error": repr(exc)}
        if args.c5_frames and args.config != "C5":
            del stream
            torch.cuda.empty_cache()
            try:
                line["c5"] = side_config(hap_amd, ctx, dev, "C5", args.c5_frames, flags, fence,
                                         reference_frames=0 if args.no_cpu_baseline else 1, ctx_dec=ctx_dec)
            except Exception as exc:
                line["c5"] = {"error": repr(exc)}
            for small in ("C2", "C3"):
                try:
                    line[small.lower()] = side_config(hap_amd, ctx, dev, small, CONFIGS[small][4], flags, fence, ctx_dec=ctx_dec)
                except Exception as exc:
                    line[small.lower()] = {"error": repr(exc)}
            for name, fn in (("c1", lambda: c1_object(hap_amd, ctx, dev)), ("bc7_opaque", lambda: opaque_object(hap_amd, ctx, dev, fence))):
                torch.cuda.empty_cache()
                try:
                    line[name] = fn()
                except Exception as exc:
                    line[name] = {"error": repr(exc)}
    else:
        per_rank = [len(frames_of_rank(nf_total, r, world, args.scaling)) for r in range(world)]
        line["config"] = {"workload": "%s: %dx%d %s, %s chunks, Snappy, %d-frame stream, device-resident" % (
            args.config, w, h, "+".join("%#x" % f for f in fmts), "+".join(map(str, chunks)), nf_total),
            "frames_per_step": total_frames // args.steps, "frames_per_rank": per_rank,
            "fragment_index": not args.no_fragment_index,
            "parallelism": "frame f -> GPU f mod %d, no data-path collective" % world if args.scaling == "strong"
                           else "a whole stream per GPU, no data-path collective"}
        line["other_scaling_mode"] = other
        line["c5_chunk_groups"] = groups
        line["rccl_ranks_seen"] = int(dist.get_world_size())
        line["collective_backend"] = str(dist.get_backend())
        if args.one_gpu_ranks:
            line["dry_run"] = "all ranks on one GPU, gloo collectives: a functional check of the multi-rank path, not a measurement"
        kernels, _ratio = stream.kernel_table(prof, args.steps, args.config)
        line["roofline"] = stream.roofline(kernels, args.config)
        line["roofline"]["note"] = "rank 0's launches (%d frames per step)" % stream.nf
        line["cpu_baseline"] = None
    if watchdog is not None:
        watchdog.cancel()
    if world == 1:
        line["summary"] = summary_of(line)
    print(json.dumps(line))
    sys.stdout.flush()
    if dist is not None:
        # (the line is out: a rank that never arrives must not keep this one from leaving)
        leave = None
        if world > 1:
            import threading
            leave = threading.Timer(60.0, lambda: os._exit(0))
            leave.daemon = True
            leave.start()
        dist.barrier()
        if leave is not None:
            leave.cancel()
        dist.destroy_process_group()


def summary_of(line):
    """The numbers a reader of the line's LAST two thousand characters must find (the driver's record keeps the parsed
    core and a tail of that length): per-kernel milliseconds of the headline config, the serial step, the small batch,
    frames without the private table, frames of the reference encoder, and the other configs' headline figures."""
    def g(o, *path):
        for k in path:
            o = o.get(k) if isinstance(o, dict) else None
        return o
    def r(v, n=4):
        return round(v, n) if isinstance(v, (int, float)) else v
    k = line.get("kernels") or {}
    c5 = line.get("c5") or {}
    ref, ref5 = line.get("decode_of_reference_encoded_frames") or {}, c5.get("decode_of_reference_encoded_frames") or {}
    out = {
        "c4": {"ms_step": line.get("ms_per_step"), "ms_step_serial": g(line, "serial_step", "ms_per_step"),
               "kernels_ms": {n: r(v.get("ms_avg")) for n, v in k.items()},
               "encode_only_ms": g(line, "encode_only", "ms"), "decode_only_ms": g(line, "decode_only", "ms"),
               "ratio": g(line, "config", "snappy_ratio"), "roofline_frac": g(line, "roofline", "frac"),
               "issue_bound": g(line, "roofline", "issue_bound"),
               "decode_frac": r((g(k, "snappy_decode", "algorithmic_GBps") or 0) / HBM_PEAK_GBPS)},
        "small_batch": {"ms_step": g(line, "small_batch", "pipelined", "ms_per_step"),
                        "implied_strong_scaling_at_8": g(line, "small_batch", "implied_strong_scaling_at_8")},
        "plain_frames_batched": {"decode_ms": g(line, "plain_frames_batched", "decode_ms"),
                                 "encode_ms": g(line, "plain_frames_batched", "encode_ms")},
        "per_call_hap_h": {"plain_decode_ms": g(line, "per_call_hap_h", "plain_frames", "decode_ms_per_call"),
                           "table_decode_ms": g(line, "per_call_hap_h", "with_private_table", "decode_ms_per_call")},
        "reference_frames": {"frames": ref.get("frames"), "ms": ref.get("ms"), "one_frame_ms": ref.get("one_frame_ms"),
                             "frac": ref.get("frac_of_hbm_peak"), "bit_exact": ref.get("bit_exact")},
        "c5": {"value": c5.get("value"), "ms_step": c5.get("ms_per_step"), "ms_step_serial": g(c5, "serial_step", "ms_per_step"),
               "ratio": c5.get("snappy_ratio"), "decode_frac": g(c5, "roofline", "frac"), "decode_ms": g(c5, "roofline", "avg_launch_ms"),
               "issue_bound": g(c5, "roofline", "issue_bound"),
               "decode_by_layout_frac": {n: v.get("frac") for n, v in (c5.get("decode_by_layout") or {}).items() if isinstance(v, dict)},
               "encode_only_ms": g(c5, "encode_only", "ms"), "kernels_ms": {n: r(v.get("ms_avg")) for n, v in (c5.get("kernels") or {}).items()},
               "reference_frame_ms": ref5.get("ms"), "bit_exact": c5.get("bit_exact")},
        "c2": {"value": g(line, "c2", "value"), "ratio": g(line, "c2", "snappy_ratio"), "decode_frac": g(line, "c2", "roofline", "frac")},
        "c3": {"value": g(line, "c3", "value"), "ratio": g(line, "c3", "snappy_ratio"), "decode_frac": g(line, "c3", "roofline", "frac")},
        "bit_exact": line.get("bit_exact"), "value": line.get("value"),
    }
    return out


def selftest_cpu(args, dist, rank, world):
    """The launch path without a GPU: same spawning, rank split, barrier and MAX-reduce, gloo instead of RCCL and a
    sleep of 1 ms per frame instead of the codec.  Used by tests/test_sharding_gloo.py; never a measurement."""
    nf_total = args.frames or CONFIGS[args.config][4]
    mine = frames_of_rank(nf_total, rank, world, args.scaling)

    def fence():
        if dist is not None:
            dist.barrier()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3 * len(mine))
    fence()
    elapsed = time.perf_counter() - t0
    counts = [len(mine)]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        box = [None] * world
        dist.all_gather_object(box, len(mine))
        counts = box
    if rank == 0:
        total = (nf_total if args.scaling == "strong" else nf_total * world) * args.steps
        print(json.dumps({"selftest": True, "n_gpus": world, "scaling": args.scaling, "frames_per_rank": counts,
                          "frames_per_step": total // args.steps, "ms_per_step": round(elapsed / args.steps * 1e3, 3)}))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def texture_to_rgba(stream, dev):
    """DXT -> RGBA, untimed extra: what a player without texture units needs after HapDecode"""
    ctx, w, h, fmts = stream.ctx, stream.w, stream.h, stream.fmts
    rgba_out = torch.empty(w * h * 4, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # (one untimed call first: the first kernel that touches a fresh allocation pays for its pages -- r03's figure of
    # 56 us per frame was the mean of one such call and fifteen of 38 us)
    ctx.decompress_rgba(stream.dec[0][0], fmts[0], w, h, rgba=rgba_out, alpha=(stream.dec[1][0] if len(fmts) > 1 else None))
    ctx.set_profiling(True)
    ctx.collect_profile()
    for i in range(min(stream.nf, 16)):
        ctx.decompress_rgba(stream.dec[0][i], fmts[0], w, h, rgba=rgba_out, alpha=(stream.dec[1][i] if len(fmts) > 1 else None))
    n, ms = ctx.collect_profile().get("block_decode", (0, 0.0))
    ctx.set_profiling(False)
    if not n:
        return None
    return {"us_per_frame": round(ms / n * 1e3, 2),
            "algorithmic_GBps": round((sum(stream.tex_bytes) + stream.rgba_bytes) / (ms / n * 1e-3) / 1e9, 1)}


def frames_to_rgba(hap_amd, stream, dev):
    """Hap frames -> RGBA8 pictures in one call (HapGpuDecodeFramesRGBA: second stage undone into scratch, then the
    block decoder), untimed extra.  Checked here against the two-step path (decoded textures ->
    HapGpuDecompressRGBA), which the GPU tests check against the oracle."""
    ctx, w, h, fmts, nf = stream.ctx, stream.w, stream.h, stream.fmts, stream.nf
    pics = hap_amd.BufferList([torch.empty(stream.rgba_bytes, dtype=torch.uint8, device=dev) for _ in range(nf)])
    torch.cuda.synchronize()
    r, res = ctx.decode_frames_rgba(stream.frames, stream.used, len(fmts), pics, w, h)
    if r or any(res):
        return {"error": "HapResult %d" % r}
    ctx.timer_start()
    ctx.decode_frames_rgba(stream.frames, stream.used, len(fmts), pics, w, h)
    ms = ctx.timer_stop()
    two_step = torch.empty(stream.rgba_bytes, dtype=torch.uint8, device=dev)
    same = True
    for i in (0, nf - 1):
        ctx.decompress_rgba(stream.dec[0][i], fmts[0], w, h, rgba=two_step, alpha=(stream.dec[1][i] if len(fmts) > 1 else None))
        same = same and bool(torch.equal(two_step, pics[i]))
    return {"frames": nf, "ms": round(ms, 3), "fps": round(nf / (ms * 1e-3), 1),
            "rgba_GBps": round(nf * stream.rgba_bytes / (ms * 1e-3) / 1e9, 1), "same_as_two_steps": same}


def separate_passes(hap_amd, stream, device_index, config, fence, steps=6):
    """The same step with the block encoder as a pass of its own, the compressor reading the texture it wrote and a gather
    pass placing the fragments (a context made under HAP_AMD_NO_FUSION / HAP_AMD_NO_PLACING: what the calls that start
    from textures, two-texture frames and small batches run): reported beside the default, never `value`.  Carries the
    block encoder's own roofline (64 + b bytes per block)."""
    keys = ("HAP_AMD_NO_FUSION", "HAP_AMD_NO_PLACING")
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ[k] = "1"
    try:
        ctx2 = hap_amd.Context(device_index)
    finally:
        for k in keys:
            if old[k] is None:
                del os.environ[k]
            else:
                os.environ[k] = old[k]
    ctx1, stream.ctx = stream.ctx, ctx2
    try:
        elapsed, prof = stream.timed(steps, 2, fence)
        kernels, _ratio = stream.kernel_table(prof, steps, config)
        # (traffic: this round's PMC passes over the same command under the same two variables, tools/prof_bench.sh <cfg> <n> sep)
        roof = stream.roofline(kernels, config + "sep", kernel="block_encode") if "block_encode" in kernels else None
        same = stream.bit_exact()
    finally:
        stream.ctx = ctx1
        ctx2.close()
    ms = elapsed / steps * 1e3
    return {"rgba_GBps": round(stream.nf * stream.rgba_bytes / (ms * 1e-3) / 1e9, 2), "ms_per_step": round(ms, 3), "bit_exact": same,
            "kernels_ms": {k: v["ms_avg"] for k, v in kernels.items()}, "roofline": roof}


def coarse_option(stream, hap_amd):
    """the size-for-speed option (HAPGPU_ENCODE_COARSE_MATCHES), reported beside the default; never `value`"""
    if not hasattr(hap_amd, "ENCODE_COARSE_MATCHES"):
        return None
    cflags = stream.flags | hap_amd.ENCODE_COARSE_MATCHES
    stream.encode(cflags)
    stream.ctx.timer_start()
    cused = stream.encode(cflags)
    stream.decode(cused)
    c_ms = stream.ctx.timer_stop()
    return {"rgba_GBps": round(stream.nf * stream.rgba_bytes / (c_ms * 1e-3) / 1e9, 2), "ms": round(c_ms, 3),
            "snappy_ratio": round(sum(cused) / stream.nf / sum(stream.tex_bytes), 4),
            "note": "encode+decode with 32-bit granular element streams for every format"}


def smaller_option(stream, hap_amd):
    """the speed-for-size option (HAPGPU_ENCODE_SMALLER_FILES: 64 KiB fragments, no private table), reported beside the
    default; never `value`"""
    if not hasattr(hap_amd, "ENCODE_SMALLER_FILES"):
        return None
    sflags = hap_amd.ENCODE_SMALLER_FILES
    stream.encode(sflags)
    stream.ctx.timer_start()
    sused = stream.encode(sflags)
    enc_ms = stream.ctx.timer_stop()
    stream.decode(sused)
    stream.ctx.timer_start()
    stream.decode(sused)
    dec_ms = stream.ctx.timer_stop()
    return {"snappy_ratio": round(sum(sused) / stream.nf / sum(stream.tex_bytes), 4),
            "encode_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
            "rgba_GBps": round(stream.nf * stream.rgba_bytes / ((enc_ms + dec_ms) * 1e-3) / 1e9, 2),
            "note": "encode / decode of the same frames with 64 KiB Snappy fragments and no fragment table"}


def small_batch(hap_amd, ctx, ctx_dec, dev, config, flags, fence, ms_full, ms_full_serial, nf_full, frames=8, steps=40):
    """What each rank of an 8-GPU strong-scaling run of the stream works on: `frames` frames per step (60 over 8 ranks:
    8/8/8/8/7/7/7/7).  Fixed costs per call -- header read-back, host plan, small launches, completion round trips --
    are a third of such a step when the calls block; the pipelined step hides them under the other call's kernels.
    implied_strong_scaling_at_8 = (ms per step of the whole stream / 8) / (ms per step of 8 frames): what one GPU's
    numbers predict for the efficiency of the 8-GPU run (nothing else is shared between the ranks)."""
    s = Stream(hap_amd, ctx, dev, config, list(range(frames)), flags, ctx_dec=ctx_dec)
    res = {"frames_per_step": frames, "steps": steps}
    best = {}
    for mode in (["pipelined"] if ctx_dec is not None else []) + ["serial"]:
        e, prof = min((s.timed(steps, 3, fence, pipelined=(mode == "pipelined")) for _ in range(2)), key=lambda r: r[0])
        ms = e / steps * 1e3
        best[mode] = ms
        res[mode] = {"ms_per_step": round(ms, 4), "fps": round(frames * steps / e, 1),
                     "rgba_GBps": round(frames * steps * s.rgba_bytes / e / 1e9, 2)}
        if mode == "serial":
            res[mode]["kernels_ms_per_step"] = {k: round(t / steps, 4) for k, (n, t) in prof.items() if n}
            res[mode]["not_in_kernels_ms"] = round(ms - sum(t for _k, (n, t) in prof.items() if n) / steps, 4)
    res["bit_exact"] = s.bit_exact()
    share = nf_full / float(frames)
    if "pipelined" in best and ms_full:
        res["implied_strong_scaling_at_8"] = round(ms_full / share / best["pipelined"], 4)
    if ms_full_serial:
        res["implied_strong_scaling_at_8_serial_calls"] = round(ms_full_serial / share / best["serial"], 4)
    return res


def plain_frames_batched(hap_amd, ctx, dev, config, frames, fence, steps=6):
    """The frames plain hap.h HapEncode writes by default -- nothing the Hap specification does not name, no private
    table -- through the batched calls: same pictures, same step (blocking calls), beside the headline; never `value`.
    Their chunks are concatenations of independent 8 KiB fragments which the decoder has to find (block scan)."""
    s = Stream(hap_amd, ctx, dev, config, list(range(frames)), 0)
    elapsed, prof = min((s.timed(steps, 2, fence), s.timed(steps, 0, fence)), key=lambda r: r[0])
    kernels, ratio = s.kernel_table(prof, steps, config)
    enc_ms, dec_ms = s.split_rates()
    ok = s.bit_exact(reference=True)
    return {"frames_per_step": frames, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "rgba_GBps": round(frames * steps * s.rgba_bytes / elapsed / 1e9, 2), "snappy_ratio": round(ratio, 4),
            "encode_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
            "decode_texture_GBps": round(frames * sum(s.tex_bytes) / (dec_ms * 1e-3) / 1e9, 1),
            "bit_exact": ok, "frame_0_decoded_by": getattr(s, "reference_checked", "-"),
            "kernels_ms": {k: v["ms_avg"] for k, v in kernels.items()}}


def fine_chunks_option(hap_amd, ctx, dev, config, frames, fence, steps=6):
    """HAPGPU_ENCODE_FINE_CHUNKS: one chunk per 8 KiB fragment in the tables every Hap parser reads, nothing private in the
    frame.  Same pictures, same step (blocking calls), beside the headline; never `value`."""
    w, h, fmts, _chunks, _n = CONFIGS[config]
    tex_bytes = [(w // 4) * (h // 4) * BLOCK_BYTES[f] for f in fmts]
    fine = [hap_amd.fine_chunk_count(tb, f) for tb, f in zip(tex_bytes, fmts)]
    name = config + "fine"
    CONFIGS[name] = (w, h, fmts, fine, frames)
    try:
        s = Stream(hap_amd, ctx, dev, name, list(range(frames)), hap_amd.ENCODE_FINE_CHUNKS)
        elapsed, prof = min((s.timed(steps, 2, fence), s.timed(steps, 0, fence)), key=lambda r: r[0])
        kernels, ratio = s.kernel_table(prof, steps, name)
        enc_ms, dec_ms = s.split_rates()
        ok = s.bit_exact(reference=True)
        one = hap_amd.BufferList([s.frames[0]])
        one_out = hap_amd.BufferList([s.dec[0][0]])
        ctx.decode_frames(one, s.used[:1], 0, one_out)
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.decode_frames(one, s.used[:1], 0, one_out)
        one_ms = (time.perf_counter() - t0) / 20 * 1e3
    finally:
        del CONFIGS[name]
    return {"chunks": fine, "frames_per_step": frames, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "rgba_GBps": round(frames * steps * s.rgba_bytes / elapsed / 1e9, 2), "snappy_ratio": round(ratio, 4),
            "encode_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
            "decode_texture_GBps": round(frames * sum(s.tex_bytes) / (dec_ms * 1e-3) / 1e9, 1),
            "one_frame_decode_call_ms": round(one_ms, 4),
            "bit_exact": ok, "frame_0_decoded_by": getattr(s, "reference_checked", "-"),
            "kernels_ms": {k: v["ms_avg"] for k, v in kernels.items()}}


def decode_by_layout(s, steps=4):
    """The decode launch of a two-texture stream, one texture at a time (HapGpuDecodeFrames, index 0 / 1): what the colour
    texture and the alpha plane cost by themselves (in the step they are units of ONE launch)."""
    out = {}
    for idx, fmt in enumerate(s.fmts):
        s.ctx.decode_frames(s.frames, s.used, idx, s.dec[idx])
        s.ctx.set_profiling(True)
        s.ctx.collect_profile()
        for _ in range(steps):
            r = s.ctx.decode_frames(s.frames, s.used, idx, s.dec[idx])
            if r[0] != 0:
                return {"error": "HapResult %d" % r[0]}
        n, ms = s.ctx.collect_profile().get("snappy_decode", (0, 0.0))
        s.ctx.set_profiling(False)
        if not n:
            continue
        # compressed bytes of this texture: its section of every frame (the frame's tables say; cheaper: by ratio of
        # the decoder's own counters is not available -- the section length is read from the frame header on the host)
        comp = 0
        for f in range(s.nf):
            head = s.frames[f][:64].cpu().numpy().tobytes()
            comp += _section_bytes(head, idx, len(s.fmts))
        alg = s.nf * s.tex_bytes[idx] + comp
        out["%#x" % fmt] = {"ms_avg": round(ms / n, 4), "algorithmic_GBps": round(alg / (ms / n * 1e-3) / 1e9, 1),
                             "frac": round(alg / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "snappy_ratio": round(comp / float(s.nf * s.tex_bytes[idx]), 4)}
    return out


def _section_bytes(head, idx, count):
    """bytes of texture section `idx` of a frame, from the frame's first bytes (sections: hap.c:137-212)"""
    def sect(at):
        ln = int.from_bytes(head[at:at + 3], "little")
        if ln == 0:
            return 8, int.from_bytes(head[at + 4:at + 8], "little")
        return 4, ln
    if count == 1:
        return sect(0)[1]
    h0, _l0 = sect(0)
    h1, l1 = sect(h0)
    if idx == 0:
        return l1
    # the second section starts behind the first: its header is not in the prefix -- total minus the first
    return _l0 - (h1 + l1) - 8


def side_config(hap_amd, ctx, dev, config, frames, flags, fence, steps=6, reference_frames=0, ctx_dec=None):
    """The other BASELINE.json configs beside the headline, same step and timing rules: C5 = the north-star's target,
    16384x16384 Hap Q Alpha (YCoCg-DXT5 + RGTC1, 64 + 64 chunks, two-texture frame); C2 / C3 = the 4K configs.
    `value` is the pipelined step's when a decode context is given (as for the headline), the per-kernel events and the
    roofline those of the serial region."""
    s = Stream(hap_amd, ctx, dev, config, list(range(frames)), flags, ctx_dec=ctx_dec)
    # (two timed regions of `steps` steps, the faster one reported: with a few milliseconds per region one stall of
    # the box -- seen once: 15 ms -- would otherwise be the number)
    elapsed, prof = min((s.timed(steps, 2, fence), s.timed(steps, 0, fence)), key=lambda r: r[0])
    serial_elapsed = elapsed
    if ctx_dec is not None:
        pipe_ok = True
        elapsed = min(s.timed(steps, 1, fence, pipelined=True)[0] for _ in range(2))
        pipe_ok = s.bit_exact()
    kernels, ratio = s.kernel_table(prof, steps, config)
    enc_ms, dec_ms = s.split_rates()
    total = frames * steps
    res = {"workload": "%s: %dx%d %s, %s chunks, Snappy, %d frames per step, device-resident" % (
                config, s.w, s.h, "+".join("%#x" % f for f in s.fmts), "+".join(map(str, s.chunks)), frames),
            "value": round(total * s.rgba_bytes / elapsed / 1e9, 2), "unit": "GB/s", "fps": round(total / elapsed, 2),
            "steps": steps, "timed_regions": 2, "ms_per_step": round(elapsed / steps * 1e3, 3), "snappy_ratio": round(ratio, 4),
            "compressed_bytes_per_step": int(sum(s.used)),
            "bit_exact": bool(s.bit_exact() and (ctx_dec is None or pipe_ok)),
            "step": "pipelined" if ctx_dec is not None else "serial",
            "serial_step": {"ms_per_step": round(serial_elapsed / steps * 1e3, 3),
                            "value": round(total * s.rgba_bytes / serial_elapsed / 1e9, 2)},
            "encode_only": {"rgba_GBps": round(frames * s.rgba_bytes / (enc_ms * 1e-3) / 1e9, 2), "ms": round(enc_ms, 3)},
            "decode_only": {"rgba_GBps": round(frames * s.rgba_bytes / (dec_ms * 1e-3) / 1e9, 2), "ms": round(dec_ms, 3),
                            "texture_GBps": round(frames * sum(s.tex_bytes) / (dec_ms * 1e-3) / 1e9, 2)},
            "roofline": s.roofline(kernels, config, kernel="snappy_decode"),
            "kernels": kernels}
    if len(s.fmts) > 1:
        try:
            res["decode_by_layout"] = decode_by_layout(s)
        except Exception as exc:
            res["decode_by_layout"] = {"error": repr(exc)}
    if reference_frames:
        # the same textures as the reference encoder writes them (libsnappy streams, no table): every existing Hap file
        try:
            s.used = s.encode()
            s.decode(s.used)
            res["decode_of_reference_encoded_frames"] = decode_foreign(ctx, dev, s.fmts, s.chunks, s.dec, s.tex_bytes, s.cap,
                                                                       min(reference_frames, frames), s.rgba_bytes,
                                                                       with_whole_stream=False)
        except Exception as exc:
            res["decode_of_reference_encoded_frames"] = {"error": repr(exc)}
    return res


def c5_chunk_groups(hap_amd, ctx, dist, dev, rank, world, fence, reps=3):
    """One 16K Hap Q Alpha frame split over the ranks by chunk groups: rank r encodes its band of block
    rows as a frame of chunks/W chunks; band frames are gathered on rank 0 (RCCL send/recv over xGMI) and joined
    (HapGpuJoinChunkGroups); then every rank decodes its chunk group of the joined frame in place and the slices are
    gathered on rank 0.  Reported without and with the gathers.  All ranks call this; rank 0 gets the dict."""
    from hap_amd import shard, synth
    w = h = 16384
    nchunks = 64
    fmts = [0x01, 0x8DBB]
    try:
        lo, hi, band_chunks = shard.band_for_rank(h // 4, nchunks, rank, world)
    except ValueError as exc:
        return {"skipped": str(exc)}
    rows = (hi - lo) * 4
    band = synth.rgba_frame(w, rows, 1000 + rank, device=dev)
    tex_bytes = [(w // 4) * (rows // 4) * b for b in (16, 8)]
    cap = hap_amd.HapMaxEncodedLength(tex_bytes, fmts, [band_chunks] * 2)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def encode_band():
        r, used, res = ctx.encode_frames_rgba([band], w, rows, w * 4, fmts, [1, 1], [band_chunks] * 2, [out],
                                              flags=hap_amd.ENCODE_FRAGMENT_INDEX)
        if r != 0:
            raise RuntimeError("band encode failed %r %r" % (r, res))
        return out[: used[0]]

    t = {}
    encode_band()
    fence(); t0 = time.perf_counter()
    for _ in range(reps):
        piece = encode_band()
    fence(); t["encode_bands_ms"] = (time.perf_counter() - t0) / reps * 1e3
    shard.gather_variable(piece, root=0)                                   # warm the send/recv path
    fence(); t0 = time.perf_counter()
    parts = shard.gather_variable(piece, root=0)
    fence(); t["gather_band_frames_ms"] = (time.perf_counter() - t0) * 1e3
    # join on the device: the band frames arrived over xGMI and stay in HBM (tables through the host, payloads D2D).
    # Like the other legs: one untimed call first (scratch arenas, the output buffer), then the timed ones.
    joined_bytes = 0
    dframe = None
    if parts is not None:
        dframe = torch.empty(sum(int(p.numel()) for p in parts) + 64, dtype=torch.uint8, device=dev)
        part_bytes = [int(p.numel()) for p in parts]
        torch.cuda.synchronize()
        ctx.join_chunk_groups(parts, part_bytes, dframe)
    fence(); t0 = time.perf_counter()
    if parts is not None:
        for _ in range(reps):
            r, joined_bytes = ctx.join_chunk_groups(parts, part_bytes, dframe)
            if r != 0:
                raise RuntimeError("join failed %r" % r)
    fence(); t["join_on_device_ms"] = (time.perf_counter() - t0) / reps * 1e3
    if dframe is not None:
        dframe = dframe[:joined_bytes]
    n = torch.tensor([joined_bytes], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=0)
    if dframe is None:
        dframe = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(dframe, src=0)
    # (the collective runs on the backend's own stream and the library reads the frame on ITS own: without this the
    # header fetch below can see the buffer before the broadcast has landed -- found by the two-ranks-on-one-GPU test)
    torch.cuda.synchronize()
    frame = dframe
    ok = True
    for idx in (0, 1):
        r, layout = hap_amd.HapGpuGetFrameTextureChunkLayout(frame, idx)
        if r != 0:
            raise RuntimeError("layout failed %r" % r)
        whole = torch.zeros(layout[-1], dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        group = shard.chunk_group_for_rank(len(layout) - 1, rank, world)

        def decode_group():
            r, _used, fmt = ctx.decode_chunk_group(dframe, idx, group.start, len(group), whole)
            if r != 0 or fmt != fmts[idx]:
                raise RuntimeError("group decode failed %r" % r)
        decode_group()
        fence(); t0 = time.perf_counter()
        for _ in range(reps):
            decode_group()
        fence(); t["decode_groups_tex%d_ms" % idx] = (time.perf_counter() - t0) / reps * 1e3
        bounds = [layout[(len(layout) - 1) * r // world] for r in range(world + 1)]
        shard.exchange_slices(whole, bounds, root=0)
        fence(); t0 = time.perf_counter()
        shard.exchange_slices(whole, bounds, root=0)
        fence(); t["gather_slices_tex%d_ms" % idx] = (time.perf_counter() - t0) * 1e3
        want = torch.empty(tex_bytes[idx], dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        r, _u = ctx.compress_rgba(band, w, rows, w * 4, fmts[idx], want)
        a, b = layout[group.start], layout[group.start + len(group)]
        ok = ok and r == 0 and bool(torch.equal(whole[a:b], want))
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    for k in list(t):
        v = torch.tensor([t[k]], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        t[k] = float(v.item())
    rgba_gb = w * h * 4 / 1e9
    enc = t["encode_bands_ms"]
    dec = t["decode_groups_tex0_ms"] + t["decode_groups_tex1_ms"]
    enc_g = enc + t["gather_band_frames_ms"] + t["join_on_device_ms"]
    dec_g = dec + t["gather_slices_tex0_ms"] + t["gather_slices_tex1_ms"]
    return {"workload": "one 16384x16384 Hap Q Alpha frame, 64+64 chunks, split by chunk groups over %d GPUs" % world,
            "bit_exact": bool(flag.item()), "frame_bytes": int(n.item()),
            "without_gather": {"encode_rgba_GBps": round(rgba_gb / (enc / 1e3), 1), "decode_rgba_GBps": round(rgba_gb / (dec / 1e3), 1)},
            "with_rccl_gather_to_rank0": {"encode_rgba_GBps": round(rgba_gb / (enc_g / 1e3), 1),
                                          "decode_rgba_GBps": round(rgba_gb / (dec_g / 1e3), 1)},
            "ms": {k: round(v, 3) for k, v in t.items()}}


_PERCALL = {}


def percall_lib():
    """tools/percall_loop.c compiled against hap_amd/libhap_amd.so: a plain C client that calls hap.h once per frame
    (no Python or ctypes time inside the loop).  None when no C compiler is at hand: the callers then loop in Python
    and say so."""
    if "lib" not in _PERCALL:
        _PERCALL["lib"] = None
        try:
            import atexit
            import shutil
            import subprocess
            import tempfile
            d = tempfile.mkdtemp(prefix="hap_percall_")
            atexit.register(shutil.rmtree, d, True)
            so = os.path.join(d, "libpercall.so")
            subprocess.run(["gcc", "-O2", "-std=c99", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tools", "percall_loop.c"), "-o", so, "-L", os.path.join(ROOT, "hap_amd"),
                            "-l:libhap_amd.so", "-Wl,-rpath," + os.path.join(ROOT, "hap_amd")], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            lib = C.CDLL(so)
            lib.percall_decode.restype = C.c_double
            lib.percall_encode.restype = C.c_double
            _PERCALL["lib"] = lib
        except Exception:
            pass
    return _PERCALL["lib"]


def _addr(buf):
    return buf.data_ptr() if hasattr(buf, "data_ptr") else buf.ctypes.data


def percall_times(hap_amd, textures, tex_bytes, fmts, chunks, frames, cap, outs, reps, table):
    """hap.h HapEncode (texture bytes in, as the reference takes them) and HapDecode, ONE CALL PER FRAME, every buffer
    where the caller put it (device tensors or host arrays).  textures[f][t]; returns (encode s, decode s per texture
    index, used bytes, ok).  `table`: HAP_AMD_FRAGMENT_INDEX for the calls (plain hap.h writes the private table on
    request only)."""
    n, count = len(textures), len(fmts)
    old = os.environ.get("HAP_AMD_FRAGMENT_INDEX")
    os.environ["HAP_AMD_FRAGMENT_INDEX"] = "1" if table else "0"
    try:
        lib = percall_lib()
        used = (C.c_ulong * n)()
        if lib is not None:
            tp = (C.c_void_p * (n * count))(*[_addr(textures[f][t]) for f in range(n) for t in range(count)])
            tb = (C.c_ulong * count)(*tex_bytes)
            cf = (C.c_uint * count)(*fmts); cc = (C.c_uint * count)(*([1] * count)); ck = (C.c_uint * count)(*chunks)
            fp = (C.c_void_p * n)(*[_addr(b) for b in frames])
            lib.percall_encode(tp, tb, C.c_uint(count), cf, cc, ck, C.c_uint(n), fp, C.c_ulong(cap), used, C.c_uint(1))   # warm
            t_enc = lib.percall_encode(tp, tb, C.c_uint(count), cf, cc, ck, C.c_uint(n), fp, C.c_ulong(cap), used, C.c_uint(reps))
            if t_enc < 0:
                raise RuntimeError("HapEncode failed: %r" % t_enc)
            fb = (C.c_ulong * n)(*[int(u) for u in used])
            t_dec = []
            for t in range(count):
                op = (C.c_void_p * n)(*[_addr(outs[t][f]) for f in range(n)])
                lib.percall_decode(fp, fb, C.c_uint(n), C.c_uint(t), op, C.c_ulong(tex_bytes[t]), C.c_uint(1))
                d = lib.percall_decode(fp, fb, C.c_uint(n), C.c_uint(t), op, C.c_ulong(tex_bytes[t]), C.c_uint(reps))
                if d < 0:
                    raise RuntimeError("HapDecode failed: %r" % d)
                t_dec.append(d / reps)
            return t_enc / reps, t_dec, [int(u) for u in used], "C"
        # no compiler: the same calls through ctypes (their time includes Python's)
        def enc_all():
            for f in range(n):
                r, u = hap_amd.HapEncode(list(textures[f]), fmts, [1] * count, chunks, outputBuffer=frames[f], outputBufferBytes=cap)
                if r != 0:
                    raise RuntimeError("HapEncode failed: %r" % r)
                used[f] = u
        enc_all()
        t0 = time.perf_counter()
        for _ in range(reps):
            enc_all()
        t_enc = (time.perf_counter() - t0) / reps
        t_dec = []
        for t in range(count):
            def dec_all():
                for f in range(n):
                    r, _u, _fm = hap_amd.HapDecode(frames[f][: used[f]], t, outputBuffer=outs[t][f])
                    if r != 0:
                        raise RuntimeError("HapDecode failed: %r" % r)
            dec_all()
            t0 = time.perf_counter()
            for _ in range(reps):
                dec_all()
            t_dec.append((time.perf_counter() - t0) / reps)
        return t_enc, t_dec, [int(u) for u in used], "python"
    finally:
        if old is None:
            del os.environ["HAP_AMD_FRAGMENT_INDEX"]
        else:
            os.environ["HAP_AMD_FRAGMENT_INDEX"] = old


def c1_object(hap_amd, ctx, dev):
    """BASELINE.json configs[0]: HapDecode() of ONE 1920x1080 Hap1 (DXT1) frame with a single Snappy chunk -- and the
    HapEncode that made it -- through plain hap.h, one call, device and host pointers, beside the unmodified reference
    on one host thread.  One 1 MB chunk is 127 fragments of work and two completion round trips per call: this is the
    case where a CPU can win, and the line says which way it went."""
    import numpy as np
    from hap_amd import synth
    w, h, fmts, chunks, _n = CONFIGS["C1"]
    tb = [(w // 4) * (h // 4) * 8]
    cap = hap_amd.HapMaxEncodedLength(tb, fmts, chunks)
    rgba = synth.rgba_frame(w, h, 0, device=dev)
    tex = torch.empty(tb[0], dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    if ctx.compress_rgba(rgba, w, h, w * 4, fmts[0], tex) != (0, tb[0]):
        raise RuntimeError("block encode failed")
    ctx.synchronize()
    out = {"workload": "C1: one %dx%d Hap1 (DXT1) frame, 1 chunk, Snappy, plain hap.h HapEncode / HapDecode, one call" % (w, h),
           "texture_bytes": tb[0]}
    tex_host = tex.cpu().numpy()
    reps = 20
    for where in ("device_pointers", "host_pointers"):
        leg = {}
        for table in (False, True):
            if where == "device_pointers":
                textures = [[tex]]
                frames = [torch.empty(cap, dtype=torch.uint8, device=dev)]
                outs = [[torch.zeros(tb[0], dtype=torch.uint8, device=dev)]]
                torch.cuda.synchronize()
            else:
                textures = [[tex_host]]
                frames = [np.zeros(cap, dtype=np.uint8)]
                outs = [[np.zeros(tb[0], dtype=np.uint8)]]
            t_enc, t_dec, used, loop = percall_times(hap_amd, textures, tb, fmts, chunks, frames, cap, outs, reps, table)
            got = outs[0][0].cpu().numpy() if hasattr(outs[0][0], "cpu") else outs[0][0]
            leg["with_private_table" if table else "plain_frame"] = {
                "encode_ms": round(t_enc * 1e3, 4), "decode_ms": round(t_dec[0] * 1e3, 4), "frame_bytes": used[0],
                "bit_exact": bool(np.array_equal(got, tex_host))}
            leg["loop"] = loop
        out[where] = leg
    # the unmodified reference, one thread, same texture (tests/_libs is checker infrastructure: outside every GPU timing)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _libs as L
        ref = L.ref_lib()
        lib, prefix, kind = (ref, "refbase", "reference hap.c + libsnappy 1.1.8") if ref is not None else (L.oracle_lib(), "oraclebase", "oracle/ C port")
        enc = getattr(lib, prefix + "_encode"); enc.restype = C.c_double
        decp = getattr(lib, prefix + "_decode_parallel"); decp.restype = C.c_double
        ptr = (C.c_void_p * 1)(tex_host.ctypes.data)
        lens = (C.c_ulong * 1)(tb[0]); cf = (C.c_uint * 1)(fmts[0]); cc = (C.c_uint * 1)(1); ck = (C.c_uint * 1)(chunks[0])
        frame = np.zeros(cap, dtype=np.uint8); used = (C.c_ulong * 1)()
        dout = np.zeros(tb[0], dtype=np.uint8)
        te, td = [], []
        for _ in range(15):
            te.append(enc(C.c_uint(1), ptr, lens, cf, cc, ck, C.c_uint(1), frame.ctypes.data_as(C.c_void_p), C.c_ulong(cap), used, C.c_uint(1), C.c_uint(1)))
            fp = (C.c_void_p * 1)(frame.ctypes.data); fl = (C.c_ulong * 1)(used[0])
            td.append(decp(fp, fl, C.c_uint(1), C.c_uint(0), dout.ctypes.data_as(C.c_void_p), C.c_ulong(tb[0]), C.c_uint(1), C.c_uint(1)))
        te.sort(); td.sort()
        cpu = {"encoder": kind, "threads": 1, "encode_ms": round(te[len(te) // 2] * 1e3, 4), "decode_ms": round(td[len(td) // 2] * 1e3, 4),
               "frame_bytes": int(used[0]), "repetitions": 15, "statistic": "median"}
        out["cpu_reference_one_thread"] = cpu
        gpu_dec = out["device_pointers"]["plain_frame"]["decode_ms"]
        gpu_dec_t = out["device_pointers"]["with_private_table"]["decode_ms"]
        host_dec = out["host_pointers"]["plain_frame"]["decode_ms"]
        out["who_wins_decode"] = {
            "device_pointers_plain_frame": "gpu" if gpu_dec < cpu["decode_ms"] else "cpu",
            "device_pointers_with_private_table": "gpu" if gpu_dec_t < cpu["decode_ms"] else "cpu",
            "host_pointers_plain_frame": "gpu" if host_dec < cpu["decode_ms"] else "cpu"}
        out["who_wins_encode"] = {"device_pointers": "gpu" if out["device_pointers"]["plain_frame"]["encode_ms"] < cpu["encode_ms"] else "cpu",
                                  "host_pointers": "gpu" if out["host_pointers"]["plain_frame"]["encode_ms"] < cpu["encode_ms"] else "cpu"}
    except Exception as exc:
        out["cpu_reference_one_thread"] = {"error": repr(exc)}
    return out


def per_call_object(hap_amd, stream, dev, n=16, reps=3):
    """What a drop-in client does: one hap.h HapEncode (texture bytes in) and one HapDecode per frame, device pointers,
    on the headline's frames -- next to the batched calls of the timed region.  Frames written by plain hap.h carry no
    private table unless HAP_AMD_FRAGMENT_INDEX=1 asks for it: both are reported."""
    n = min(n, stream.nf)
    fmts, chunks, tb, cap = stream.fmts, stream.chunks, stream.tex_bytes, stream.cap
    textures = [[stream.dec[t][f] for t in range(len(fmts))] for f in range(n)]       # the textures the last step decoded
    want = [[textures[f][t].clone() for t in range(len(fmts))] for f in range(n)]
    frames = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(n)]
    outs = [[torch.zeros(tb[t], dtype=torch.uint8, device=dev) for _ in range(n)] for t in range(len(fmts))]
    torch.cuda.synchronize()
    res = {"workload": "hap.h HapEncode (texture bytes in) + HapDecode, one call per frame, device pointers, %d frames of the stream" % n}
    for table in (False, True):
        t_enc, t_dec, used, loop = percall_times(hap_amd, want, tb, fmts, chunks, frames, cap, outs, reps, table)
        torch.cuda.synchronize()
        ok = all(bool(torch.equal(outs[t][f], want[f][t])) for f in range(n) for t in range(len(fmts)))
        dec = sum(t_dec)
        res["with_private_table" if table else "plain_frames"] = {
            "encode_ms_per_call": round(t_enc / n * 1e3, 4), "decode_ms_per_call": round(dec / n * 1e3, 4),
            "encode_texture_GBps": round(n * sum(tb) / t_enc / 1e9, 2), "decode_texture_GBps": round(n * sum(tb) / dec / 1e9, 2),
            "decode_rgba_GBps": round(n * stream.rgba_bytes / dec / 1e9, 2),
            "snappy_ratio": round(sum(used) / n / sum(tb), 4), "bit_exact": ok}
        res["loop"] = loop
    return res


def opaque_object(hap_amd, ctx, dev, fence, fmt=0x8E8C, n=30, steps=4):
    """an opaque 16-byte-block format (BC7: the library has no block encoder for it, hap.c:369-375 takes the
    bytes as they are) at 8K size through the batched calls: the position-per-lane compressor and the generic fragment
    decoder, not the block kernels of the headline."""
    from hap_amd import synth
    w, h, chunks = 7680, 4320, [24]
    tb = [(w // 4) * (h // 4) * 16]
    cap = hap_amd.HapMaxEncodedLength(tb, [fmt], chunks)
    # texture-like bytes: the YCoCg-DXT5 blocks of synthetic pictures stand in for BC7 blocks (opaque to the codec)
    texs = []
    for i in range(n):
        t = torch.empty(tb[0], dtype=torch.uint8, device=dev)
        picture = synth.rgba_frame(w, h, i, device=dev)
        torch.cuda.synchronize()              # (torch made the picture on ITS stream; the library reads it on its own)
        if ctx.compress_rgba(picture, w, h, w * 4, 0x01, t) != (0, tb[0]):
            raise RuntimeError("block encode failed")
        texs.append(t)
        del picture
    ctx.synchronize()
    tex_l = hap_amd.BufferList(texs)
    frames = hap_amd.BufferList([torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(n)])
    outs = hap_amd.BufferList([torch.zeros(tb[0], dtype=torch.uint8, device=dev) for _ in range(n)])
    torch.cuda.synchronize()
    flags = hap_amd.ENCODE_FRAGMENT_INDEX

    def enc():
        r, used, res = ctx.encode_frames([[t] for t in tex_l], [fmt], [1], chunks, frames, flags=flags)
        if r != 0:
            raise RuntimeError("opaque encode failed %r" % (res[:2],))
        return used

    def dec(used):
        r, du, _f, dres = ctx.decode_frames(frames, used, 0, outs)
        if r != 0 or du[0] != tb[0]:
            raise RuntimeError("opaque decode failed %r" % (dres[:2],))
    def measure(fl):
        nonlocal flags
        flags = fl
        used = enc(); dec(used)
        best_e = best_d = None
        for _ in range(steps):
            ctx.timer_start(); used = enc(); e = ctx.timer_stop()
            ctx.timer_start(); dec(used); d = ctx.timer_stop()
            best_e = e if best_e is None else min(best_e, e)
            best_d = d if best_d is None else min(best_d, d)
        torch.cuda.synchronize()
        ok = all(bool(torch.equal(outs[i], texs[i])) for i in range(n))
        ratio = sum(used) / n / tb[0]
        return {"snappy_ratio": round(ratio, 4), "bit_exact": ok,
                "encode_ms": round(best_e, 3), "decode_ms": round(best_d, 3),
                "encode_texture_GBps": round(n * tb[0] / (best_e * 1e-3) / 1e9, 1), "decode_texture_GBps": round(n * tb[0] / (best_d * 1e-3) / 1e9, 1),
                "decode_algorithmic_GBps": round(n * tb[0] * (1 + ratio) / (best_d * 1e-3) / 1e9, 1),
                "decode_frac_of_hbm_peak": round(n * tb[0] * (1 + ratio) / (best_d * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    res = {"workload": "%dx%d opaque 16-byte blocks (format %#x, BC7), 24 chunks, Snappy, %d frames per call, device-resident" % (w, h, fmt, n),
           "statistic": "best of %d calls" % steps}
    res.update(measure(hap_amd.ENCODE_FRAGMENT_INDEX))
    # the size-for-speed option the library keeps for exactly these formats: elements on 32-bit positions
    if hasattr(hap_amd, "ENCODE_COARSE_MATCHES"):
        res["coarse_matches_option"] = measure(hap_amd.ENCODE_FRAGMENT_INDEX | hap_amd.ENCODE_COARSE_MATCHES)
    return res


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


def decode_foreign(ctx, dev, fmts, chunks, dec, tex_bytes, cap, n, rgba_bytes, with_whole_stream=True):
    """Frames produced by the CPU reference encoder (libsnappy streams, no fragment table) decoded by
    the GPU: block scan, then one wavefront per 64 KiB block; the one-wavefront-per-chunk path beside it
    (DECODE_NO_BLOCK_SCAN), for a batch and for one frame.  Reported beside the headline, never part of it."""
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
    del out
    # entry f * T + t = texture t of frame f (one call decodes every texture of every frame)
    outs = [torch.empty(tex_bytes[t], dtype=torch.uint8, device=dev) for _f in range(n) for t in range(count)]
    torch.cuda.synchronize()
    import hap_amd

    def timed(count_frames, flags):
        fr, us, ou = frames[:count_frames], [int(u) for u in used[:count_frames]], outs[:count_frames * count]
        call = (lambda: ctx.decode_frame_textures(fr, us, count, ou, flags)) if count > 1 else (lambda: ctx.decode_frames(fr, us, 0, ou, flags))
        call()                                                          # warm-up
        best = None
        for _ in range(3):
            ctx.timer_start()
            r = call()[0]
            ms = ctx.timer_stop()
            if r != 0:
                return None
            best = ms if best is None else min(best, ms)
        return best

    def same(count_frames):
        return all(torch.equal(outs[f * count + t], dec[t][f]) for f in range(count_frames) for t in range(count))
    ms = timed(n, 0)
    ok = ms is not None and same(n)
    for o in outs:
        o.zero_()
    one = timed(1, 0)
    ok = ok and one is not None and same(1)
    whole = timed(n, hap_amd.DECODE_NO_BLOCK_SCAN) if with_whole_stream else None
    whole_one = timed(1, hap_amd.DECODE_NO_BLOCK_SCAN) if with_whole_stream else None
    rnd = lambda v: None if v is None else round(v, 3)
    bsum = sum(tex_bytes)
    res = {"frames": n, "ms": rnd(ms), "rgba_GBps": round(n * rgba_bytes / (ms * 1e-3) / 1e9, 2) if ms else None,
           "texture_GBps": round(n * bsum / (ms * 1e-3) / 1e9, 2) if ms else None, "bit_exact": bool(ok),
           "one_frame_ms": rnd(one), "snappy_ratio": round(sum(int(u) for u in used) / n / bsum, 4),
           "encoder": "reference hap.c + libsnappy 1.1.8" if ref is not None else "oracle/ C port"}
    if ms:
        alg = n * (bsum + sum(int(u) for u in used) / n)
        res["algorithmic_GBps"] = round(alg / (ms * 1e-3) / 1e9, 1)
        res["frac_of_hbm_peak"] = round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    if with_whole_stream:
        res["without_block_scan"] = {"ms": rnd(whole), "one_frame_ms": rnd(whole_one)}
    return res


def measured_traffic(config, kernel, frames):
    """HBM bytes per launch of `kernel` from this round's rocprofv3 PMC passes over THIS command
    (tools/prof_traffic.sh <config>: FETCH_SIZE and WRITE_SIZE in separate runs; FETCH doubled as the MI355X guide
    prescribes for wide streaming reads; the raw per-dispatch CSV summaries are committed beside the json under
    profiles/).  Counters cannot be read from inside the timed process, so the bench line carries the profile's
    figure scaled to the frames per launch of this run, and names its source.  (None, None) when the round has no
    profile for the config."""
    name = "%s_traffic_%s.json" % (ROUND_TAG, config.lower())
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            t = json.load(f)
        k = t["kernels"][kernel]
    except (OSError, KeyError, ValueError):
        return None, None
    scale = frames / float(t["frames_per_launch"])
    return int((2.0 * k["fetch_kib"] + k["write_kib"]) * 1024 * scale), "profiles/" + name


def measured_valu(config, kernel, frames):
    """Vector instructions per launch of `kernel` (SQ_INSTS_VALU of the same rocprofv3 PMC passes, scaled to this run's
    frames per launch), or None."""
    path = os.path.join(ROOT, "profiles", "%s_traffic_%s.json" % (ROUND_TAG, config.lower()))
    try:
        with open(path) as f:
            t = json.load(f)
        return float(t["kernels"][kernel]["valu_insts"]) * frames / float(t["frames_per_launch"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


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
    # Every figure below is the MEDIAN of `reps` >= 10 repetitions, buffers pre-faulted by an untimed
    # pass, worker threads pinned one per CPU of the process's affinity mask (oracle/cpu_baseline.c).
    def median(fn, reps):
        ts = sorted(fn() for _ in range(reps))
        if ts[0] < 0:
            raise RuntimeError("cpu leg failed %r" % ts[0])
        return ts[len(ts) // 2]
    pin = getattr(lib, prefix + "_pinning", None)
    pinned = bool(pin()) if pin is not None else False
    # block encode (oracle's scalar C, the reference has none): rows of one frame over all threads
    out = np.zeros(max(tex_bytes), dtype=np.uint8)

    def bc_once():
        t = 0.0
        for fmt in fmts:
            t += ora.oraclebase_bc_encode(rgba_host.ctypes.data_as(C.c_void_p), C.c_uint(w), C.c_uint(h),
                                          C.c_size_t(w * 4), C.c_uint(fmt), out.ctypes.data_as(C.c_void_p),
                                          C.c_uint(threads), C.c_uint(1))
        return t
    # container + Snappy: one frame per thread (the reference's encode is serial per frame)
    ptrs = (C.c_void_p * (nwork * count))(*[tex_host[f % sample][i].ctypes.data for f in range(nwork) for i in range(count)])
    lens = (C.c_ulong * count)(*tex_bytes)
    cf = (C.c_uint * count)(*fmts); cc = (C.c_uint * count)(*([1] * count)); ck = (C.c_uint * count)(*chunks)
    enc_threads = min(threads, nwork)
    outbuf = np.zeros(cap * enc_threads, dtype=np.uint8)
    used = (C.c_ulong * nwork)()

    def enc_once():
        return enc(C.c_uint(count), ptrs, lens, cf, cc, ck, C.c_uint(nwork), outbuf.ctypes.data_as(C.c_void_p),
                   C.c_ulong(cap), used, C.c_uint(enc_threads), C.c_uint(1))
    t_first = time.perf_counter()
    bc_once(); enc_once()                       # untimed: page faults, thread start-up
    one_pass = time.perf_counter() - t_first
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

    def dec_once():
        t = 0.0
        for idx in range(count):
            d = decp(fptrs, flens, C.c_uint(nwork), C.c_uint(idx), dout.ctypes.data_as(C.c_void_p),
                     C.c_ulong(stride), C.c_uint(enc_threads), C.c_uint(1))
            if d < 0:
                return d
            t += d
        return t
    dec_once()
    # repetitions: at least 10, more while the budget lasts (one pass of all three legs took `one_pass` seconds)
    reps = int(max(10, min(30, (budget_s * 0.3) / max(one_pass * 1.5, 1e-3))))
    # each leg with its threads pinned and left to the scheduler: the CPU gets the better median
    set_pin = [getattr(l_, n_, None) for l_, n_ in ((lib, prefix + "_set_pinning"), (ora, "oraclebase_set_pinning"))]
    legs = {}
    for mode in ((1, 0) if all(set_pin) else (int(pinned),)):
        for f in set_pin:
            if f is not None:
                f(C.c_int(mode))
        for name, fn in (("block_encode", bc_once), ("hap_encode", enc_once), ("hap_decode", dec_once)):
            t = median(fn, reps)
            if name not in legs or t < legs[name][0]:
                legs[name] = (t, bool(mode))
    for f in set_pin:
        if f is not None:
            f(C.c_int(1))
    t_bc = legs["block_encode"][0]
    t_enc = legs["hap_encode"][0] / nwork
    t_dec = legs["hap_decode"][0] / nwork
    pinned = {k: v[1] for k, v in legs.items()}
    # one hardware thread, one frame (the reference as a client would call it serially)
    one_out = np.zeros(cap, dtype=np.uint8)
    one_used = (C.c_ulong * 1)()
    p1 = (C.c_void_p * count)(*[tex_host[0][i].ctypes.data for i in range(count)])
    f1 = (C.c_void_p * 1)(frames[0].ctypes.data)
    l1 = (C.c_ulong * 1)(len(frames[0]))
    reps1 = 10
    t_enc1 = median(lambda: enc(C.c_uint(count), p1, lens, cf, cc, ck, C.c_uint(1), one_out.ctypes.data_as(C.c_void_p), C.c_ulong(cap),
                                one_used, C.c_uint(1), C.c_uint(1)), reps1)
    t_dec1 = median(lambda: sum(decp(f1, l1, C.c_uint(1), C.c_uint(idx), dout.ctypes.data_as(C.c_void_p), C.c_ulong(stride),
                                     C.c_uint(1), C.c_uint(1)) for idx in range(count)), reps1)
    sample_n = sample
    sample = 1      # t_enc / t_dec are already per frame
    rgba_bytes = w * h * 4
    per_frame = t_bc + t_enc / sample + t_dec / sample
    return {"value": round(rgba_bytes / per_frame / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": kind,
            "threads": {"block_encode": threads, "hap_encode": enc_threads, "hap_decode": enc_threads},
            "sample": "%d frames in flight (%d distinct) of this workload: RGBA->DXT by oracle/bc_oracle.c (the reference "
                      "has no block encoder) + HapEncode + HapDecode by %s, %d threads, amortised per frame; median of %d "
                      "repetitions per leg after an untimed pass" % (
                          nwork, sample_n, "unmodified reference hap.c + libsnappy 1.1.8" if kind == "reference" else "oracle/ C port", threads, reps),
            "repetitions": reps, "statistic": "median of each leg, pinned and unpinned threads, the faster of the two",
            "threads_pinned": pinned,
            "ms_per_frame": {"block_encode": round(t_bc * 1e3, 2), "hap_encode": round(t_enc / sample * 1e3, 2),
                             "hap_decode": round(t_dec / sample * 1e3, 2)},
            "container_only_rgba_GBps": round(rgba_bytes / (t_enc / sample + t_dec / sample) / 1e9, 3),
            "single_thread_ms_per_frame": {"hap_encode": round(t_enc1 * 1e3, 2), "hap_decode": round(t_dec1 * 1e3, 2)}}


if __name__ == "__main__":
    main()
