"""CPU-side checks of the product library (no compute calls: there is no GPU here):
the C ABI loads, exports every symbol include/*.h declares, and the host-only functions
(size arithmetic, frame inspectors) agree with the oracle / golden vectors."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _data as D
import _libs as L

ROOT = L.ROOT


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(Hap[A-Za-z0-9]+)\s*\(", text)) - {"HapDecodeWorkFunction", "HapDecodeCallback"})


@pytest.fixture(scope="module")
def hap():
    from hap_amd.build import build
    build()
    import hap_amd
    return hap_amd


def test_every_declared_symbol_is_exported(hap):
    lib = C.CDLL(os.path.join(ROOT, "hap_amd", "libhap_amd.so"))
    names = _declared_functions("hap.h") + _declared_functions("hap_gpu.h")
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n
    for n in ("HapMaxEncodedLength", "HapEncode", "HapDecode", "HapGetFrameTextureCount",
              "HapGetFrameTextureFormat", "HapGetFrameTextureChunkCount"):
        assert n in names


def test_enum_values_match_reference_abi(hap):
    hdr = open(os.path.join(ROOT, "include", "hap.h")).read()
    for name, val in [("HapTextureFormat_RGB_DXT1", 0x83F0), ("HapTextureFormat_RGBA_DXT5", 0x83F3),
                      ("HapTextureFormat_YCoCg_DXT5", 0x01), ("HapTextureFormat_A_RGTC1", 0x8DBB),
                      ("HapTextureFormat_RGBA_BPTC_UNORM", 0x8E8C), ("HapTextureFormat_RGB_BPTC_UNSIGNED_FLOAT", 0x8E8F),
                      ("HapTextureFormat_RGB_BPTC_SIGNED_FLOAT", 0x8E8E), ("HapCompressorNone", 0), ("HapCompressorSnappy", 1),
                      ("HapResult_No_Error", 0), ("HapResult_Bad_Arguments", 1), ("HapResult_Buffer_Too_Small", 2),
                      ("HapResult_Bad_Frame", 3), ("HapResult_Internal_Error", 4)]:
        m = re.search(name + r"\s*=\s*(0x[0-9A-Fa-f]+|\d+)", hdr)
        assert m and int(m.group(1), 0) == val, name


def test_max_encoded_length_matches_oracle(hap):
    ora = L.oracle_api()
    rng = np.random.default_rng(0)
    for _ in range(300):
        count = int(rng.integers(1, 3))
        lengths = [int(rng.integers(1, 1 << 22)) * 8 for _ in range(count)]
        fmts = [int(rng.choice(L.ALL_FORMATS)) for _ in range(count)]
        chunks = [int(rng.integers(1, 200)) for _ in range(count)]
        assert hap.HapMaxEncodedLength(lengths, fmts, chunks) == ora.max_encoded_length(lengths, fmts, chunks)
    assert hap.HapMaxEncodedLength([64], [L.FMT_DXT1], [0]) == 0
    assert hap.HapMaxEncodedLength([], [], []) == 0
    # BASELINE configs
    assert hap.HapMaxEncodedLength([33177600], [L.FMT_YCOCG], [24]) == ora.max_encoded_length([33177600], [L.FMT_YCOCG], [24])
    assert hap.HapMaxEncodedLength([268435456, 134217728], [L.FMT_YCOCG, L.FMT_RGTC1], [64, 64]) == \
        ora.max_encoded_length([268435456, 134217728], [L.FMT_YCOCG, L.FMT_RGTC1], [64, 64])


def test_inspectors_on_golden_and_malformed_frames(hap):
    ora = L.oracle_api()
    frames = [bytes.fromhex(v["frame"]) for v in D.golden_vectors("frame") if v["frame"]]
    rng = np.random.default_rng(1)
    for frame in frames:
        variants = [frame] + [frame[:k] for k in range(1, min(len(frame), 40))]
        for _ in range(40):
            f = bytearray(frame)
            f[int(rng.integers(0, min(len(f), 48)))] = int(rng.integers(0, 256))
            variants.append(bytes(f))
        for f in variants:
            assert hap.HapGetFrameTextureCount(f) == ora.texture_count(f)
            for idx in (0, 1, 2):
                assert hap.HapGetFrameTextureFormat(f, idx) == ora.texture_format(f, idx)
                assert hap.HapGetFrameTextureChunkCount(f, idx) == ora.chunk_count(f, idx)


def test_no_gpu_means_loud_failure_not_fallback(hap):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r, frame = hap.HapEncode([bytes(64)], [L.FMT_DXT1], [1], [1])
    assert r == hap.HapResult.Internal_Error and frame is None
    f = bytes.fromhex(D.golden_vectors("frame")[0]["frame"])
    assert hap.HapDecode(f, 0, outputBufferBytes=64)[0] == hap.HapResult.Internal_Error
    with pytest.raises(RuntimeError):
        hap.Context(0)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under hap_amd/ may include, link, load or import it
    (comments that cite oracle/bc_oracle.c as the definition of the block algorithm are fine)."""
    import re
    forbidden = re.compile(r'#\s*include\s*[<"][^>"]*oracle|liboracle|-loracle|-L\S*oracle|dlopen|'
                           r'import\s+_libs|from\s+_libs|import\s+oracle|oracle_lib|osnappy_|ohap_|obc_encode|'
                           r'CDLL\([^)]*oracle')
    for base, _dirs, files in os.walk(os.path.join(ROOT, "hap_amd")):
        if os.path.basename(base) == "build":
            continue
        for name in files:
            if name.endswith((".c", ".h", ".hip", ".py", "Makefile", ".map")):
                text = open(os.path.join(base, name), errors="ignore").read()
                assert not forbidden.search(text), (base, name, forbidden.search(text).group(0))
