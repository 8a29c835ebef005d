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


def quality_images():
    """Three fixed 512 x 256 pictures the block encoder's quality is pinned on: a smooth gradient, the gradient plus
    position-hashed noise of +-16, and hard two-colour edges with thin lines (alpha: ramp / noisy ramp / 0, 128, 255)."""
    y, x = np.mgrid[0:256, 0:512]
    smooth = np.stack([(x * 255) // 511, (y * 255) // 255, ((x + y) * 255) // 766, 255 - (x * 255) // 511], -1).astype(np.uint8)
    h = (x * 0x9E3779B1 + y * 0x85EBCA77) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
    h ^= h >> 12
    noisy = np.stack([np.clip(smooth[..., c].astype(int) + ((h >> (4 * c)) & 31) - 16, 0, 255) for c in range(4)], -1).astype(np.uint8)
    stripe = (((x + 2 * y) >> 3) & 1) == 1
    edge = np.where(stripe[..., None], np.array([230, 40, 20, 255]), np.array([15, 60, 200, 0])).astype(np.uint8)
    edge[(x % 37) == 0] = (255, 255, 255, 128)
    return {"smooth": np.ascontiguousarray(smooth), "noisy": np.ascontiguousarray(noisy), "hard_edge": np.ascontiguousarray(edge)}


def block_quality(blocks, fmt, img):
    """PSNR of the oracle decoder's picture against the source: (colour,), (colour, alpha) for DXT5, (alpha,) for RGTC1"""
    h, w = img.shape[:2]
    dec = oracle_bc_decode(blocks, fmt, w, h)
    if fmt == L.FMT_RGTC1:
        return (psnr(dec, img[..., 3]),)
    if fmt == L.FMT_DXT5:
        return (psnr(dec[..., :3], img[..., :3]), psnr(dec[..., 3], img[..., 3]))
    return (psnr(dec[..., :3], img[..., :3]),)


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


# ---- third-party pin of the block layouts: Pillow's DDS reader (S3TC / RGTC, HapVideoDRAFT.md:22-27) ----
def dds_file(blocks, w, h, fourcc):
    """A minimal legacy DDS file around `blocks` (FourCC pixel format: b"DXT1", b"DXT5", b"ATI1")."""
    import struct
    flags = 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000            # caps, height, width, pixel format, linear size
    head = struct.pack("<4sIIIIIII44x", b"DDS ", 124, flags, h, w, len(blocks), 0, 1)
    pixel_format = struct.pack("<II4sIIIII", 32, 0x4, fourcc, 0, 0, 0, 0, 0)
    caps = struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    return head + pixel_format + caps + bytes(blocks)


PILLOW_FOURCC = {L.FMT_DXT1: b"DXT1", L.FMT_DXT5: b"DXT5", L.FMT_YCOCG: b"DXT5", L.FMT_RGTC1: b"ATI1"}


def pillow_bc_decode(blocks, fmt, w, h):
    """Decodes block data with Pillow (no code of ours involved).  DXT1/DXT5 -> RGBA, RGTC1 -> one plane;
    scaled YCoCg-DXT5 is read as the DXT5 texture it is (R = Co, G = Cg, B = scale code, A = Y)."""
    import io
    from PIL import Image
    im = Image.open(io.BytesIO(dds_file(blocks, w, h, PILLOW_FOURCC[fmt])))
    im.load()
    return np.asarray(im).copy()


def shader_ycocg_to_rgb(tex):
    """The float reconstruction a Hap Q player's fragment shader does on the sampled DXT5 texel
    (van Waveren & Castano, cited by HapVideoDRAFT.md:24): tex = uint8 [h, w, 4] = (Co, Cg, scale, Y)."""
    t = tex.astype(np.float64) / 255.0
    scale = 1.0 / (t[..., 2] * (255.0 / 8.0) + 1.0)
    co = (t[..., 0] - 128.0 / 255.0) * scale
    cg = (t[..., 1] - 128.0 / 255.0) * scale
    y = t[..., 3]
    rgb = np.stack([y + co - cg, y + cg, y - co - cg], axis=-1)
    return np.clip(np.rint(rgb * 255.0), 0, 255).astype(np.uint8)
