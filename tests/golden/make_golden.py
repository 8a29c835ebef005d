#!/usr/bin/env python3
"""Regenerates tests/golden/hap_golden.json by running the UNMODIFIED reference
(/root/reference/source/hap.c + libsnappy 1.1.8, built into oracle/_ref by
oracle/Makefile) on small fixed inputs.  Only runs where /root/reference is
mounted; the JSON it writes is committed and is what travels to the GPU box.

Vectors G-A1..G-A5 are the ones listed in SURVEY.md Appendix A.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _libs as L  # noqa: E402


def c_rand_bytes(n, seed):
    libc = C.CDLL("libc.so.6")
    libc.srand(C.c_uint(seed))
    return bytes(libc.rand() & 0xFF for _ in range(n))


def snappy_compress(data):
    s = L.snappy_lib()
    s.snappy_max_compressed_length.restype = C.c_size_t
    cap = s.snappy_max_compressed_length(C.c_size_t(len(data)))
    out = (C.c_ubyte * cap)()
    ln = C.c_size_t(cap)
    assert s.snappy_compress(bytes(data), C.c_size_t(len(data)), out, C.byref(ln)) == 0
    return bytes(out[: ln.value])


def main():
    ref = L.ref_api()
    assert ref is not None, "reference not built"
    rng = np.random.default_rng(0x48415031)
    vectors = []

    def frame_case(name, textures, formats, comps, chunks):
        r, frame = ref.encode(textures, formats, comps, chunks)
        maxlen = ref.max_encoded_length([len(t) for t in textures], formats, chunks)
        entry = dict(kind="frame", name=name, textures=[t.hex() for t in textures], formats=formats,
                     compressors=comps, chunks=chunks, max_encoded_length=maxlen, result=r,
                     frame=frame.hex() if frame is not None else None)
        if frame is not None:
            entry["texture_count"] = list(ref.texture_count(frame))
            dec = []
            for idx in range(len(textures)):
                dr, out, fmt = ref.decode(frame, idx, out_bytes=max(len(t) for t in textures) + 64)
                dec.append(dict(result=dr, format=fmt, callback_calls=ref.callback_calls,
                                chunk_count=list(ref.chunk_count(frame, idx)),
                                texture_format=list(ref.texture_format(frame, idx)),
                                equals_input=(out == bytes(textures[idx]))))
            entry["decode"] = dec
        vectors.append(entry)

    aa = bytes([0xAA, 0, 0, 0, 0, 0, 0, 0]) * 8
    frame_case("G-A1 dxt1 snappy 2 chunks", [aa], [L.FMT_DXT1], [1], [2])
    frame_case("G-A2 dxt1 snappy 1 chunk", [aa], [L.FMT_DXT1], [1], [1])
    frame_case("G-A3 incompressible -> raw", [c_rand_bytes(64, 1)], [L.FMT_DXT1], [1], [2])
    frame_case("G-A4 dual ycocg+rgtc1", [bytes(32), bytes(16)], [L.FMT_YCOCG, L.FMT_RGTC1], [1, 1], [1, 1])
    frame_case("none compressor dxt5", [bytes(range(64))], [L.FMT_DXT5], [0], [4])
    frame_case("chunk count limited 5->4", [bytes(16) * 8], [L.FMT_DXT5], [1], [5])
    mixed = bytes(512) + rng.integers(0, 256, 512, dtype=np.uint8).tobytes()
    frame_case("mixed chunk compressors", [mixed], [L.FMT_BC7], [1], [2])
    frame_case("bad pair dxt1+dxt5", [bytes(32), bytes(32)], [L.FMT_DXT1, L.FMT_DXT5], [1, 1], [1, 1])
    frame_case("permissive pair rgtc1+dxt1", [bytes(32), bytes(32)], [L.FMT_RGTC1, L.FMT_DXT1], [1, 1], [1, 2])
    pat = (bytes(range(16)) * 3 + rng.integers(0, 8, 16, dtype=np.uint8).tobytes()) * 40
    frame_case("pattern ycocg 8 chunks", [pat], [L.FMT_YCOCG], [1], [8])

    # raw Snappy known answers (libsnappy 1.1.8)
    for name, data in [("G-A5 1000 zeros", bytes(1000)), ("64 x aa-block", aa),
                       ("ramp 300", bytes(i & 255 for i in range(300))),
                       ("pattern", pat[:700]), ("empty", b""), ("one byte", b"Z")]:
        vectors.append(dict(kind="snappy", name=name, input=data.hex(), compressed=snappy_compress(data).hex()))

    # hand-written streams exercising every element type and the error paths
    ref_s = L.snappy_lib()

    def ref_uncompress(stream, cap):
        out = (C.c_ubyte * max(1, cap))()
        ln = C.c_size_t(cap)
        r = ref_s.snappy_uncompress(bytes(stream), C.c_size_t(len(stream)), out, C.byref(ln))
        return r, (bytes(out[: ln.value]).hex() if r == 0 else None)

    streams = {
        "copy4": bytes([10, 0x0C, 1, 2, 3, 4, 0x17, 4, 0, 0, 0]),          # lit4 + copy-4 len 6 off 4
        "literal 2-byte length": bytes([70, 0xF0, 69]) + bytes(range(70)),
        "overlap off1": bytes([9, 0x00, 7, 0x1E, 1, 0]),                    # lit1 + copy-2 len 8 off 1
        "offset zero": bytes([8, 0x00, 7, 0x1A, 0, 0]),
        "offset beyond start": bytes([8, 0x00, 7, 0x1A, 5, 0]),
        "output overrun": bytes([4, 0x00, 7, 0x1E, 1, 0]),
        "short output": bytes([20, 0x00, 7]),
        "truncated literal": bytes([5, 0x10, 1, 2]),
        "truncated copy2": bytes([9, 0x00, 7, 0x1E, 1]),
        "bad varint": bytes([0x80, 0x80, 0x80, 0x80, 0x80, 1]),
        "varint overflow": bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x10]),
        "trailing garbage": bytes([1, 0x00, 7, 0x00, 8]),
        "copy1 max": bytes([15, 0x0C, 1, 2, 3, 4, 0x1D, 4]),                # lit4 + copy-1 len 11 off 4
    }
    for name, st in streams.items():
        r, out = ref_uncompress(st, 256)
        vectors.append(dict(kind="snappy_stream", name=name, stream=st.hex(), capacity=256, result=r, output=out))
    r, out = ref_uncompress(streams["copy4"], 4)
    vectors.append(dict(kind="snappy_stream", name="buffer too small", stream=streams["copy4"].hex(),
                        capacity=4, result=r, output=out))

    path = os.path.join(HERE, "hap_golden.json")
    with open(path, "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py",
                       reference="Vidvox/hap @2024_08_07 source/hap.c + libsnappy 1.1.8",
                       vectors=vectors), f, indent=1)
    print("wrote", path, len(vectors), "vectors")


if __name__ == "__main__":
    main()
