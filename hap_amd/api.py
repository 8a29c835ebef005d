"""Python mirror of include/hap.h and include/hap_gpu.h (same names, argument
meaning and result codes as /root/reference/source/hap.h:40-152).

Buffers may be bytes / bytearray / numpy arrays (host) or objects exposing
`data_ptr()` + `numel()`/`nbytes` (torch CUDA tensors -> used in place)."""
import ctypes as C
import os

from ._lib import CALLBACK, WORK_FN, lib


class HapTextureFormat:
    RGB_DXT1 = 0x83F0
    RGBA_DXT5 = 0x83F3
    YCoCg_DXT5 = 0x01
    A_RGTC1 = 0x8DBB
    RGBA_BPTC_UNORM = 0x8E8C
    RGB_BPTC_UNSIGNED_FLOAT = 0x8E8F
    RGB_BPTC_SIGNED_FLOAT = 0x8E8E


HapCompressorNone, HapCompressorSnappy = 0, 1


class HapResult:
    No_Error, Bad_Arguments, Buffer_Too_Small, Bad_Frame, Internal_Error = range(5)


ENCODE_FRAGMENT_INDEX = 0x1
ENCODE_COARSE_MATCHES = 0x2
ENCODE_SMALLER_FILES = 0x4
ENCODE_FINE_CHUNKS = 0x8
DECODE_IGNORE_FRAGMENT_INDEX = 0x1
DECODE_IGNORE_HALF_TILES = 0x2
DECODE_NO_BLOCK_SCAN = 0x4
DECODE_NO_FIELD_GUESS = 0x8
DECODE_GUESS_FIELDS = 0x10
KERNEL_CLASSES = ["block_encode", "snappy_compress", "frame_pack", "frame_gather", "decode_plan", "snappy_decode",
                  "block_decode", "block_scan", "encode_fused"]


def _addr_len(buf):
    """(address, nbytes, keepalive) of a host or device buffer."""
    if buf is None:
        return None, 0, None
    if hasattr(buf, "data_ptr"):                      # torch tensor (host or device)
        return buf.data_ptr(), buf.numel() * buf.element_size(), buf
    if hasattr(buf, "ctypes") and hasattr(buf, "nbytes"):   # numpy
        return buf.ctypes.data, buf.nbytes, buf
    if isinstance(buf, (bytes, bytearray, memoryview)):
        raw = (C.c_ubyte * max(1, len(buf))).from_buffer_copy(bytes(buf) or b"\0")
        return C.addressof(raw), len(buf), raw
    if isinstance(buf, C.Array):
        return C.addressof(buf), C.sizeof(buf), buf
    raise TypeError("unsupported buffer type %r" % type(buf))


def HapMaxEncodedLength(lengths, textureFormats, chunkCounts):
    n = len(lengths)
    return lib.HapMaxEncodedLength(n, (C.c_ulong * n)(*lengths), (C.c_uint * n)(*textureFormats),
                                   (C.c_uint * n)(*chunkCounts))


def HapEncode(inputBuffers, textureFormats, compressors, chunkCounts, outputBuffer=None, outputBufferBytes=None):
    """Returns (result, frame bytes | used). With outputBuffer=None a host buffer of
    HapMaxEncodedLength() is allocated and the frame returned as bytes."""
    n = len(inputBuffers)
    infos = [_addr_len(b) for b in inputBuffers]
    ptrs = (C.c_void_p * n)(*[i[0] for i in infos])
    lens = (C.c_ulong * n)(*[i[1] for i in infos])
    own = outputBuffer is None
    if own:
        if outputBufferBytes is None:
            outputBufferBytes = HapMaxEncodedLength([i[1] for i in infos], textureFormats, chunkCounts)
        outputBuffer = (C.c_ubyte * max(1, outputBufferBytes))()
    oaddr, olen, _keep = _addr_len(outputBuffer)
    if outputBufferBytes is None:
        outputBufferBytes = olen
    used = C.c_ulong(0)
    r = lib.HapEncode(n, ptrs, lens, (C.c_uint * n)(*textureFormats), (C.c_uint * n)(*compressors),
                      (C.c_uint * n)(*chunkCounts), oaddr, outputBufferBytes, C.byref(used))
    if own:
        return r, (C.string_at(outputBuffer, used.value) if r == 0 else None)
    return r, used.value


def _serial_callback():
    def cb(fn, p, count, info):
        for i in range(count):
            fn(p, i)
    return CALLBACK(cb)


def HapDecode(inputBuffer, index=0, callback=None, outputBuffer=None, outputBufferBytes=1 << 20):
    """Returns (result, decoded bytes | used, textureFormat)."""
    iaddr, ilen, _k = _addr_len(inputBuffer)
    own = outputBuffer is None
    if own:
        outputBuffer = (C.c_ubyte * max(1, outputBufferBytes))()
    oaddr, olen, _k2 = _addr_len(outputBuffer)
    if not own:
        outputBufferBytes = olen
    used = C.c_ulong(0)
    fmt = C.c_uint(0)
    cb = callback if callback is not None else _serial_callback()
    r = lib.HapDecode(iaddr, ilen, index, cb, None, oaddr, outputBufferBytes, C.byref(used), C.byref(fmt))
    if own:
        return r, (C.string_at(outputBuffer, used.value) if r == 0 else None), fmt.value
    return r, used.value, fmt.value


def HapGetFrameTextureCount(frame):
    a, n, _k = _addr_len(frame)
    out = C.c_uint(0)
    return lib.HapGetFrameTextureCount(a, n, C.byref(out)), out.value


def HapGetFrameTextureFormat(frame, index):
    a, n, _k = _addr_len(frame)
    out = C.c_uint(0)
    return lib.HapGetFrameTextureFormat(a, n, index, C.byref(out)), out.value


def HapGetFrameTextureChunkCount(frame, index):
    a, n, _k = _addr_len(frame)
    out = C.c_int(-1)
    return lib.HapGetFrameTextureChunkCount(a, n, index, C.byref(out)), out.value


def HapGpuGetFrameTextureChunkLayout(frame, index):
    """Returns (result, [decoded offset of every chunk ..., decoded size of the texture])."""
    a, n, _k = _addr_len(frame)
    r, count = HapGetFrameTextureChunkCount(frame, index)
    cap = max(1, count) + 1 if r == 0 else 2
    offs = (C.c_ulong * cap)()
    got = C.c_uint(0)
    r = lib.HapGpuGetFrameTextureChunkLayout(a, n, index, cap, offs, C.byref(got))
    return r, (list(offs[: got.value + 1]) if r == 0 else None)


def HapGpuJoinChunkGroups(groupFrames, outputBufferBytes=None):
    """Joins frames holding consecutive chunk groups (host buffers). Returns (result, frame bytes | None)."""
    infos = [_addr_len(f) for f in groupFrames]
    n = len(infos)
    if outputBufferBytes is None:
        outputBufferBytes = sum(i[1] for i in infos) + 64
    out = (C.c_ubyte * max(1, outputBufferBytes))()
    used = C.c_ulong(0)
    r = lib.HapGpuJoinChunkGroups(n, (C.c_void_p * max(1, n))(*[i[0] for i in infos]),
                                  (C.c_ulong * max(1, n))(*[i[1] for i in infos]), out, outputBufferBytes, C.byref(used))
    return r, (C.string_at(out, used.value) if r == 0 else None)


class SequenceWriter:
    """include/hap_sequence.h: append complete Hap frames to a sequence file."""

    def __init__(self, path, width=0, height=0, rate=(60, 1)):
        h = C.c_void_p()
        r = lib.HapSequenceWriterOpen(os.fsencode(path), width, height, rate[0], rate[1], C.byref(h))
        if r != 0:
            raise OSError("HapSequenceWriterOpen(%r) failed with HapResult %d" % (path, r))
        self.handle = h

    def append(self, frame):
        a, n, _k = _addr_len(frame)
        return lib.HapSequenceWriterAppend(self.handle, a, n)

    def close(self):
        r = 0
        if self.handle:
            r = lib.HapSequenceWriterClose(self.handle)
            self.handle = None
        return r

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class SequenceReader:
    """include/hap_sequence.h: random access to the frames of a sequence file."""

    def __init__(self, path):
        h = C.c_void_p()
        r = lib.HapSequenceReaderOpen(os.fsencode(path), C.byref(h))
        if r != 0:
            raise OSError("HapSequenceReaderOpen(%r) failed with HapResult %d" % (path, r))
        self.handle = h
        v = [C.c_uint(0) for _ in range(5)]
        lib.HapSequenceReaderInfo(h, *[C.byref(x) for x in v])
        self.width, self.height, self.rate, self.frame_count = v[0].value, v[1].value, (v[2].value, v[3].value), v[4].value

    def frame_bytes(self, i):
        return lib.HapSequenceReaderFrameBytes(self.handle, i)

    def read(self, first, count=1):
        """Returns (result, [frame bytes, ...])."""
        total = sum(self.frame_bytes(first + i) for i in range(count)) if first + count <= self.frame_count else 0
        buf = (C.c_ubyte * max(1, total))()
        offs = (C.c_ulong * (count + 1))()
        r = lib.HapSequenceReaderRead(self.handle, first, count, buf, total, offs)
        if r != 0:
            return r, None
        raw = C.string_at(buf, total)
        return 0, [raw[offs[i]:offs[i + 1]] for i in range(count)]

    def close(self):
        if self.handle:
            lib.HapSequenceReaderClose(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BufferList:
    """A list of buffers whose addresses and sizes are looked up once: pass it wherever the batched calls take
    a list of buffers that stay in place from call to call (a C client simply keeps its pointer array)."""

    def __init__(self, buffers):
        self.buffers = list(buffers)
        self.infos = [_addr_len(b) for b in self.buffers]
        self.pointers = (C.c_void_p * len(self.buffers))(*[i[0] for i in self.infos])

    def __len__(self):
        return len(self.buffers)

    def __getitem__(self, i):
        return self.buffers[i]


class Context:
    """HapGpuContext: device + stream + scratch (include/hap_gpu.h)."""

    def __init__(self, device=-1):
        h = C.c_void_p()
        r = lib.HapGpuCreate(device, C.byref(h))
        if r != 0:
            raise RuntimeError("HapGpuCreate failed with HapResult %d (no usable HIP device?)" % r)
        self.handle = h

    def close(self):
        if self.handle:
            lib.HapGpuDestroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_fragment_log2(self, v):
        return lib.HapGpuSetFragmentLog2(self.handle, v)

    def synchronize(self):
        return lib.HapGpuSynchronize(self.handle)

    def resolved_blocks(self):
        """64 KiB blocks of other encoders' streams decoded by a workgroup each (HapGpuResolvedBlockCount)"""
        return int(lib.HapGpuResolvedBlockCount(self.handle))

    def placement_retries(self):
        """frames encoded a second time because one of their chunks did not shrink (HapGpuPlacementRetryCount)"""
        return int(lib.HapGpuPlacementRetryCount(self.handle))

    def placement_timeouts(self):
        """... of which because a wavefront gave up waiting for its predecessors' sizes (HapGpuPlacementTimeoutCount)"""
        return int(lib.HapGpuPlacementTimeoutCount(self.handle))

    def table_fallbacks(self):
        """frames decoded a second time because their fragment table did not describe their streams"""
        return int(lib.HapGpuTableFallbackCount(self.handle))

    def compress_rgba(self, rgba, width, height, row_bytes, texture_format, output=None):
        block = 8 if texture_format in (HapTextureFormat.RGB_DXT1, HapTextureFormat.A_RGTC1) else 16
        need = (width // 4) * (height // 4) * block
        a, _n, _k = _addr_len(rgba)
        own = output is None
        if own:
            output = (C.c_ubyte * max(1, need))()
        oa, on, _k2 = _addr_len(output)
        used = C.c_ulong(0)
        r = lib.HapGpuCompressRGBA(self.handle, a, width, height, row_bytes, texture_format, oa, on, C.byref(used))
        if own:
            return r, (C.string_at(output, used.value) if r == 0 else None)
        return r, used.value

    def decompress_rgba(self, texture, texture_format, width, height, rgba=None, alpha=None, row_bytes=None):
        """Texture (+ optional RGTC1 alpha plane) -> RGBA8. Returns (result, bytes | None)."""
        ta, tn, _k = _addr_len(texture)
        aa, an, _k2 = _addr_len(alpha) if alpha is not None else (None, 0, None)
        row_bytes = row_bytes or width * 4
        own = rgba is None
        if own:
            rgba = (C.c_ubyte * (row_bytes * height + 16))()
            base = C.addressof(rgba)
            pad = (-base) % 16
            oa = base + pad
        else:
            oa, _on, _k3 = _addr_len(rgba)
        r = lib.HapGpuDecompressRGBA(self.handle, ta, tn, texture_format, aa, an, width, height, oa, row_bytes)
        if own:
            return r, (C.string_at(oa, row_bytes * height) if r == 0 else None)
        return r, None

    def decode_chunk_group(self, frame, index, first_chunk, chunk_count, output):
        """Decodes chunks [first_chunk, first_chunk + chunk_count) into their place in `output`
        (laid out as the whole texture). Returns (result, texture bytes, format)."""
        ia, il, _k = _addr_len(frame)
        oa, ol, _k2 = _addr_len(output)
        used = C.c_ulong(0)
        fmt = C.c_uint(0)
        r = lib.HapGpuDecodeChunkGroup(self.handle, ia, il, index, first_chunk, chunk_count, oa, ol,
                                       C.byref(used), C.byref(fmt))
        return r, used.value, fmt.value

    @staticmethod
    def _ptr_array(bufs):
        if isinstance(bufs, BufferList):          # addresses resolved once, reused call after call
            return bufs.pointers, bufs.infos
        infos = [_addr_len(b) for b in bufs]
        return (C.c_void_p * len(bufs))(*[i[0] for i in infos]), infos

    def encode_frames(self, textures, formats, compressors, chunk_counts, outputs, flags=0):
        """textures: list (frames) of lists (count) of buffers; outputs: list of buffers.
        Returns (result, used[], results[])."""
        nf, count = len(textures), len(formats)
        flat = [t for fr in textures for t in fr]
        ptrs, infos = self._ptr_array(flat)
        lens = (C.c_ulong * count)(*[infos[i][1] for i in range(count)])
        optrs, oinfos = self._ptr_array(outputs)
        olens = (C.c_ulong * nf)(*[i[1] for i in oinfos])
        used = (C.c_ulong * nf)()
        results = (C.c_uint * nf)()
        r = lib.HapGpuEncodeFrames(self.handle, nf, count, ptrs, lens, (C.c_uint * count)(*formats),
                                   (C.c_uint * count)(*compressors), (C.c_uint * count)(*chunk_counts),
                                   optrs, olens, used, results, flags)
        return r, list(used), list(results)

    def encode_frames_rgba(self, rgba_frames, width, height, row_bytes, formats, compressors, chunk_counts,
                           outputs, flags=0):
        nf, count = len(rgba_frames), len(formats)
        ptrs, _infos = self._ptr_array(rgba_frames)
        optrs, oinfos = self._ptr_array(outputs)
        olens = (C.c_ulong * nf)(*[i[1] for i in oinfos])
        used = (C.c_ulong * nf)()
        results = (C.c_uint * nf)()
        r = lib.HapGpuEncodeFramesRGBA(self.handle, nf, ptrs, width, height, row_bytes, count,
                                       (C.c_uint * count)(*formats), (C.c_uint * count)(*compressors),
                                       (C.c_uint * count)(*chunk_counts), optrs, olens, used, results, flags)
        return r, list(used), list(results)

    def encode_frames_rgba_begin(self, rgba_frames, width, height, row_bytes, formats, compressors, chunk_counts,
                                 outputs, flags=0):
        """First half of encode_frames_rgba (HapGpuEncodeFramesRGBABegin): everything launched, nothing waited for.
        The context takes no other call until encode_finish(), which returns what encode_frames_rgba returns."""
        nf, count = len(rgba_frames), len(formats)
        ptrs, _infos = self._ptr_array(rgba_frames)
        optrs, oinfos = self._ptr_array(outputs)
        olens = (C.c_ulong * nf)(*[i[1] for i in oinfos])
        used = (C.c_ulong * nf)()
        results = (C.c_uint * nf)()
        r = lib.HapGpuEncodeFramesRGBABegin(self.handle, nf, ptrs, width, height, row_bytes, count,
                                            (C.c_uint * count)(*formats), (C.c_uint * count)(*compressors),
                                            (C.c_uint * count)(*chunk_counts), optrs, olens, used, results, flags)
        self._pending = (used, results, ptrs, optrs, olens, rgba_frames, outputs)     # alive until the second half
        return r

    def encode_finish(self):
        r = lib.HapGpuEncodeFramesFinish(self.handle)
        pending, self._pending = getattr(self, "_pending", None), None
        if pending is None:
            return r, [], []
        return r, list(pending[0]), list(pending[1])

    def decode_frames(self, frames, frame_bytes, index, outputs, flags=0):
        nf = len(frames)
        ptrs, infos = self._ptr_array(frames)
        lens = (C.c_ulong * nf)(*[fb if fb is not None else infos[i][1] for i, fb in enumerate(frame_bytes)])
        optrs, oinfos = self._ptr_array(outputs)
        olens = (C.c_ulong * nf)(*[i[1] for i in oinfos])
        used = (C.c_ulong * nf)()
        fmts = (C.c_uint * nf)()
        results = (C.c_uint * nf)()
        r = lib.HapGpuDecodeFrames(self.handle, nf, ptrs, lens, index, optrs, olens, used, fmts, results, flags)
        return r, list(used), list(fmts), list(results)

    def decode_frame_textures(self, frames, frame_bytes, texture_count, outputs, flags=0):
        """All textures of every frame in one batch (HapGpuDecodeFrameTextures).  outputs[f * texture_count + t].
        Returns (result, used[], formats[], results[]), one entry per frame and texture."""
        nf = len(frames)
        n = nf * texture_count
        if len(outputs) != n:
            raise ValueError("outputs must hold frames x textures buffers")
        ptrs, infos = self._ptr_array(frames)
        lens = (C.c_ulong * nf)(*[fb if fb is not None else infos[i][1] for i, fb in enumerate(frame_bytes)])
        optrs, oinfos = self._ptr_array(outputs)
        olens = (C.c_ulong * n)(*[i[1] for i in oinfos])
        used = (C.c_ulong * n)()
        fmts = (C.c_uint * n)()
        results = (C.c_uint * n)()
        r = lib.HapGpuDecodeFrameTextures(self.handle, nf, ptrs, lens, texture_count, optrs, olens, used, fmts, results, flags)
        return r, list(used), list(fmts), list(results)

    def decode_frames_rgba(self, frames, frame_bytes, texture_count, rgba_frames, width, height, row_bytes=None, flags=0):
        """Frames -> RGBA8 pictures in one call (HapGpuDecodeFramesRGBA).  Returns (result, results[])."""
        nf = len(frames)
        if len(rgba_frames) != nf:
            raise ValueError("one picture per frame")
        ptrs, infos = self._ptr_array(frames)
        lens = (C.c_ulong * nf)(*[fb if fb is not None else infos[i][1] for i, fb in enumerate(frame_bytes)])
        optrs, _oinfos = self._ptr_array(rgba_frames)
        results = (C.c_uint * nf)()
        r = lib.HapGpuDecodeFramesRGBA(self.handle, nf, ptrs, lens, texture_count, optrs, width, height,
                                       row_bytes or width * 4, results, flags)
        return r, list(results)

    def decode_sequence(self, reader, first, count, index, outputs, batch=0):
        """Disk -> pinned double buffer -> GPU (HapGpuDecodeSequence). Returns (result, used[], formats[], results[])."""
        optrs, oinfos = self._ptr_array(outputs)
        olens = (C.c_ulong * count)(*[i[1] for i in oinfos])
        used = (C.c_ulong * count)()
        fmts = (C.c_uint * count)()
        results = (C.c_uint * count)()
        r = lib.HapGpuDecodeSequence(self.handle, reader.handle, first, count, index, batch, optrs, olens, used, fmts, results)
        return r, list(used), list(fmts), list(results)

    def join_chunk_groups(self, group_frames, group_bytes, output):
        """HapGpuJoinChunkGroupsDevice: group frames and output in device memory. Returns (result, used)."""
        n = len(group_frames)
        ptrs, _infos = self._ptr_array(group_frames)
        oaddr, olen, _keep = _addr_len(output)
        used = C.c_ulong(0)
        r = lib.HapGpuJoinChunkGroupsDevice(self.handle, n, ptrs, (C.c_ulong * max(1, n))(*group_bytes), oaddr, olen, C.byref(used))
        return r, used.value

    def encode_sequence(self, writer, rgba_frames, width, height, row_bytes, formats, compressors, chunk_counts,
                        flags=0, batch=0):
        """RGBA pictures -> GPU -> pinned double buffer -> file (HapGpuEncodeSequence). Returns (result, frame bytes[], results[])."""
        nf, count = len(rgba_frames), len(formats)
        ptrs, _infos = self._ptr_array(rgba_frames)
        sizes = (C.c_ulong * nf)()
        results = (C.c_uint * nf)()
        r = lib.HapGpuEncodeSequence(self.handle, writer.handle, nf, ptrs, width, height, row_bytes, count,
                                     (C.c_uint * count)(*formats), (C.c_uint * count)(*compressors),
                                     (C.c_uint * count)(*chunk_counts), flags, batch, sizes, results)
        return r, list(sizes), list(results)

    def set_profiling(self, on):
        return lib.HapGpuSetProfiling(self.handle, 1 if on else 0)

    def collect_profile(self):
        n = len(KERNEL_CLASSES)
        launches = (C.c_ulong * n)()
        ms = (C.c_double * n)()
        lib.HapGpuCollectProfileN(self.handle, n, launches, ms)
        return {k: (launches[i], ms[i]) for i, k in enumerate(KERNEL_CLASSES)}

    def timer_start(self):
        return lib.HapGpuTimerStart(self.handle)

    def timer_stop(self):
        ms = C.c_double(0)
        lib.HapGpuTimerStop(self.handle, C.byref(ms))
        return ms.value


def _handles(contexts):
    return (C.c_void_p * len(contexts))(*[c.handle for c in contexts])


def encode_frames_rgba_on_devices(contexts, rgba_frames, width, height, row_bytes, formats, compressors, chunk_counts, outputs, flags=0):
    """HapGpuEncodeFramesRGBAOnDevices: frame f -> contexts[f mod N], one host thread per context, no collective."""
    nf, count = len(rgba_frames), len(formats)
    ptrs, _infos = contexts[0]._ptr_array(rgba_frames)
    optrs, oinfos = contexts[0]._ptr_array(outputs)
    olens = (C.c_ulong * nf)(*[i[1] for i in oinfos])
    used = (C.c_ulong * nf)()
    results = (C.c_uint * nf)()
    r = lib.HapGpuEncodeFramesRGBAOnDevices(_handles(contexts), len(contexts), nf, ptrs, width, height, row_bytes, count,
                                            (C.c_uint * count)(*formats), (C.c_uint * count)(*compressors),
                                            (C.c_uint * count)(*chunk_counts), optrs, olens, used, results, flags)
    return r, list(used), list(results)


def decode_frames_on_devices(contexts, frames, frame_bytes, index, outputs, flags=0):
    """HapGpuDecodeFramesOnDevices: frame f -> contexts[f mod N]."""
    nf = len(frames)
    ptrs, infos = contexts[0]._ptr_array(frames)
    lens = (C.c_ulong * nf)(*[fb if fb is not None else infos[i][1] for i, fb in enumerate(frame_bytes)])
    optrs, oinfos = contexts[0]._ptr_array(outputs)
    olens = (C.c_ulong * nf)(*[i[1] for i in oinfos])
    used = (C.c_ulong * nf)()
    fmts = (C.c_uint * nf)()
    results = (C.c_uint * nf)()
    r = lib.HapGpuDecodeFramesOnDevices(_handles(contexts), len(contexts), nf, ptrs, lens, index, optrs, olens, used, fmts, results, flags)
    return r, list(used), list(fmts), list(results)


def fine_chunk_count(texture_bytes, texture_format):
    """HapGpuFineChunkCount: the chunk count ENCODE_FINE_CHUNKS gives a texture (size buffers with it)."""
    return int(lib.HapGpuFineChunkCount(texture_bytes, texture_format))
