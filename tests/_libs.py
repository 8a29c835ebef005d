"""ctypes loaders for the checker libraries used by the tests.

  oracle()  -> oracle/liboracle.so       (our CPU restatement; always buildable)
  ref()     -> oracle/_ref/libhap_ref.so (unmodified reference hap.c + libsnappy 1.1.8;
                                          built here where /root/reference is mounted,
                                          shipped prebuilt to the GPU box)

TEST INFRASTRUCTURE ONLY: nothing in hap_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

WORK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint)
CALLBACK = C.CFUNCTYPE(None, WORK_FN, C.c_void_p, C.c_uint, C.c_void_p)

FMT_DXT1, FMT_DXT5, FMT_YCOCG, FMT_RGTC1 = 0x83F0, 0x83F3, 0x01, 0x8DBB
FMT_BC7, FMT_BC6U, FMT_BC6S = 0x8E8C, 0x8E8F, 0x8E8E
ALL_FORMATS = [FMT_DXT1, FMT_DXT5, FMT_YCOCG, FMT_RGTC1, FMT_BC7, FMT_BC6U, FMT_BC6S]
COMP_NONE, COMP_SNAPPY = 0, 1
R_OK, R_BAD_ARGS, R_TOO_SMALL, R_BAD_FRAME, R_INTERNAL = 0, 1, 2, 3, 4

_cache = {}


def _make(target=None):
    cmd = ["make", "-s", "-C", ORACLE_DIR] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


def oracle_lib():
    if "oracle" not in _cache:
        _make()
        _cache["oracle"] = C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so"))
    return _cache["oracle"]


def ref_lib():
    """Returns the real reference library or None when it was never built."""
    if "ref" not in _cache:
        path = os.path.join(ORACLE_DIR, "_ref", "libhap_ref.so")
        if os.path.isdir("/root/reference/source"):
            _make("ref")
        _cache["ref"] = C.CDLL(path) if os.path.exists(path) else None
    return _cache["ref"]


def snappy_lib():
    """libsnappy 1.1.8 itself (shipped beside libhap_ref.so)."""
    if "snappy" not in _cache:
        ref_lib()
        path = os.path.join(ORACLE_DIR, "_ref", "libsnappy.so.1")
        _cache["snappy"] = C.CDLL(path) if os.path.exists(path) else None
    return _cache["snappy"]


def serial_callback():
    def cb(fn, p, count, info):
        for i in range(count):
            fn(p, i)
    return CALLBACK(cb)


class HapApi:
    """Uniform python face over any library exporting the six hap.h functions
    (product libhap_amd.so, reference libhap_ref.so) or the ohap_* restatement."""

    def __init__(self, lib, names=None):
        n = names or dict(max="HapMaxEncodedLength", enc="HapEncode", dec="HapDecode",
                          cnt="HapGetFrameTextureCount", fmt="HapGetFrameTextureFormat",
                          chk="HapGetFrameTextureChunkCount")
        self.lib = lib
        self._max = getattr(lib, n["max"]); self._max.restype = C.c_ulong
        self._enc = getattr(lib, n["enc"]); self._enc.restype = C.c_uint
        self._dec = getattr(lib, n["dec"]); self._dec.restype = C.c_uint
        self._cnt = getattr(lib, n["cnt"]); self._cnt.restype = C.c_uint
        self._fmt = getattr(lib, n["fmt"]); self._fmt.restype = C.c_uint
        self._chk = getattr(lib, n["chk"]); self._chk.restype = C.c_uint
        self.callback_calls = 0

    def max_encoded_length(self, lengths, formats, chunks):
        n = len(lengths)
        return self._max(C.c_uint(n), (C.c_ulong * n)(*lengths), (C.c_uint * n)(*formats),
                         (C.c_uint * n)(*chunks))

    def encode(self, textures, formats, compressors, chunks, out_bytes=None):
        """textures: list of bytes-like. Returns (result, frame bytes or None)."""
        n = len(textures)
        bufs = [(C.c_ubyte * max(1, len(t))).from_buffer_copy(bytes(t) or b"\0") for t in textures]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        lens = (C.c_ulong * n)(*[len(t) for t in textures])
        if out_bytes is None:
            out_bytes = self.max_encoded_length([len(t) for t in textures], formats, chunks)
        out = (C.c_ubyte * max(1, out_bytes))()
        used = C.c_ulong(0)
        r = self._enc(C.c_uint(n), ptrs, lens, (C.c_uint * n)(*formats), (C.c_uint * n)(*compressors),
                      (C.c_uint * n)(*chunks), out, C.c_ulong(out_bytes), C.byref(used))
        return r, (bytes(out[: used.value]) if r == 0 else None)

    def decode(self, frame, index=0, out_bytes=1 << 20, callback="serial"):
        """Returns (result, decoded bytes or None, format)."""
        buf = (C.c_ubyte * max(1, len(frame))).from_buffer_copy(bytes(frame) or b"\0")
        out = (C.c_ubyte * max(1, out_bytes))()
        used = C.c_ulong(0)
        fmt = C.c_uint(0)
        calls = [0]

        def cb(fn, p, count, info):
            calls[0] += 1
            for i in range(count):
                fn(p, i)
        cbo = CALLBACK(cb) if callback == "serial" else callback
        r = self._dec(buf, C.c_ulong(len(frame)), C.c_uint(index), cbo, None, out,
                      C.c_ulong(out_bytes), C.byref(used), C.byref(fmt))
        self.callback_calls = calls[0]
        return r, (bytes(out[: used.value]) if r == 0 else None), fmt.value

    # numpy variants for full-size frames (no Python-level byte shuffling)
    def encode_np(self, textures, formats, compressors, chunks):
        """textures: list of C-contiguous uint8 arrays. Returns (result, uint8 array or None)."""
        import numpy as np
        n = len(textures)
        ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in textures])
        lens = (C.c_ulong * n)(*[t.size for t in textures])
        cap = self.max_encoded_length([t.size for t in textures], formats, chunks)
        out = np.empty(cap, dtype=np.uint8)
        used = C.c_ulong(0)
        r = self._enc(C.c_uint(n), ptrs, lens, (C.c_uint * n)(*formats), (C.c_uint * n)(*compressors),
                      (C.c_uint * n)(*chunks), out.ctypes.data_as(C.c_void_p), C.c_ulong(cap), C.byref(used))
        return r, (out[: used.value] if r == 0 else None)

    def decode_np(self, frame, index, out_bytes):
        """frame: uint8 array. Returns (result, uint8 array or None, format); serial callback."""
        import numpy as np
        out = np.empty(max(1, out_bytes), dtype=np.uint8)
        used = C.c_ulong(0)
        fmt = C.c_uint(0)
        calls = [0]

        def cb(fn, p, count, info):
            calls[0] += 1
            for i in range(count):
                fn(p, i)
        cbo = CALLBACK(cb)
        r = self._dec(frame.ctypes.data_as(C.c_void_p), C.c_ulong(frame.size), C.c_uint(index), cbo, None,
                      out.ctypes.data_as(C.c_void_p), C.c_ulong(out_bytes), C.byref(used), C.byref(fmt))
        self.callback_calls = calls[0]
        return r, (out[: used.value] if r == 0 else None), fmt.value

    def texture_count(self, frame):
        buf = (C.c_ubyte * max(1, len(frame))).from_buffer_copy(bytes(frame) or b"\0")
        n = C.c_uint(0)
        r = self._cnt(buf, C.c_ulong(len(frame)), C.byref(n))
        return r, n.value

    def texture_format(self, frame, index):
        buf = (C.c_ubyte * max(1, len(frame))).from_buffer_copy(bytes(frame) or b"\0")
        f = C.c_uint(0)
        r = self._fmt(buf, C.c_ulong(len(frame)), C.c_uint(index), C.byref(f))
        return r, f.value

    def chunk_count(self, frame, index):
        buf = (C.c_ubyte * max(1, len(frame))).from_buffer_copy(bytes(frame) or b"\0")
        n = C.c_int(-1)
        r = self._chk(buf, C.c_ulong(len(frame)), C.c_uint(index), C.byref(n))
        return r, n.value


ORACLE_NAMES = dict(max="ohap_max_encoded_length", enc="ohap_encode", dec="ohap_decode",
                    cnt="ohap_texture_count", fmt="ohap_texture_format", chk="ohap_texture_chunk_count")


def oracle_api():
    return HapApi(oracle_lib(), ORACLE_NAMES)


def ref_api():
    lib = ref_lib()
    return HapApi(lib) if lib is not None else None
