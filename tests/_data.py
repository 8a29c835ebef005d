"""Shared seeded inputs for the parity tests (numpy views of hap_amd.synth)."""
import ctypes as C
import json
import os

import numpy as np

import _libs as L
from hap_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hap_golden.json")


def golden_vectors(kind=None):
    with open(GOLDEN) as f:
        vs = json.load(f)["vectors"]
    return [v for v in vs if kind is None or v["kind"] == kind]


def rgba(width, height, frame=0):
    return synth.rgba_frame(width, height, frame, device="cpu").numpy()


def stream_bytes(n, kind, seed=synth.SEED_BASE):
    return synth.texture_like_bytes(n, kind, seed=seed, device="cpu").numpy().tobytes()


BLOCK_BYTES = {L.FMT_DXT1: 8, L.FMT_RGTC1: 8, L.FMT_DXT5: 16, L.FMT_YCOCG: 16,
               L.FMT_BC7: 16, L.FMT_BC6U: 16, L.FMT_BC6S: 16}

_ORACLE_BC = {L.FMT_DXT1: "obc_encode_dxt1", L.FMT_DXT5: "obc_encode_dxt5",
              L.FMT_YCOCG: "obc_encode_ycocg_dxt5", L.FMT_RGTC1: "obc_encode_rgtc1_alpha"}
_ORACLE_BCDEC = {L.FMT_DXT1: "obc_decode_dxt1", L.FMT_DXT5: "obc_decode_dxt5",
                 L.FMT_YCOCG: "obc_decode_ycocg_dxt5"}


def oracle_bc_encode(img, fmt, row_bytes=None):
    """img: uint8 [h, w, 4] (C-contiguous unless row_bytes given). Returns bytes."""
    h, w = img.shape[:2]
    out = np.zeros((h // 4) * (w // 4) * BLOCK_BYTES[fmt], dtype=np.uint8)
    fn = getattr(L.oracle_lib(), _ORACLE_BC[fmt])
    fn.restype = None
    fn(img.ctypes.data_as(C.c_void_p), C.c_uint(w), C.c_uint(h),
       C.c_size_t(row_bytes if row_bytes else img.strides[0]), out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def oracle_bc_decode(blocks, fmt, w, h):
    buf = np.frombuffer(blocks, dtype=np.uint8)
    if fmt == L.FMT_RGTC1:
        out = np.zeros((h, w), dtype=np.uint8)
        fn = L.oracle_lib().obc_decode_rgtc1
    else:
        out = np.zeros((h, w, 4), dtype=np.uint8)
        fn = getattr(L.oracle_lib(), _ORACLE_BCDEC[fmt])
    fn.restype = None
    fn(buf.ctypes.data_as(C.c_void_p), C.c_uint(w), C.c_uint(h), out.ctypes.data_as(C.c_void_p))
    return out


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)


def osnappy_compress(data):
    o = L.oracle_lib()
    o.osnappy_max_compressed_length.restype = C.c_size_t
    cap = o.osnappy_max_compressed_length(C.c_size_t(len(data)))
    out = (C.c_ubyte * cap)()
    ln = C.c_size_t(cap)
    r = o.osnappy_compress(bytes(data), C.c_size_t(len(data)), out, C.byref(ln))
    assert r == 0
    return bytes(out[: ln.value])


def osnappy_uncompress(stream, cap):
    o = L.oracle_lib()
    out = (C.c_ubyte * max(1, cap))()
    ln = C.c_size_t(cap)
    r = o.osnappy_uncompress(bytes(stream), C.c_size_t(len(stream)), out, C.byref(ln))
    return r, (bytes(out[: ln.value]) if r == 0 else None)


def ref_snappy_compress(data):
    s = L.snappy_lib()
    s.snappy_max_compressed_length.restype = C.c_size_t
    cap = s.snappy_max_compressed_length(C.c_size_t(len(data)))
    out = (C.c_ubyte * cap)()
    ln = C.c_size_t(cap)
    assert s.snappy_compress(bytes(data), C.c_size_t(len(data)), out, C.byref(ln)) == 0
    return bytes(out[: ln.value])


def ref_snappy_uncompress(stream, cap):
    s = L.snappy_lib()
    out = (C.c_ubyte * max(1, cap))()
    ln = C.c_size_t(cap)
    r = s.snappy_uncompress(bytes(stream), C.c_size_t(len(stream)), out, C.byref(ln))
    return r, (bytes(out[: ln.value]) if r == 0 else None)
