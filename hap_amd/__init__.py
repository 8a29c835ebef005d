"""hap_amd -- MI355X-native Hap frame encode/decode hot path.

The product is the C-ABI shared library hap_amd/libhap_amd.so (headers in
include/): HIP kernels for gfx950 behind the reference's hap.h API.  This
Python package is a thin ctypes mirror of that API for tests and bench.py.

The shared library is loaded on first use of an API name (so that
`hap_amd.build` and `hap_amd.synth` can be imported before it exists); there is
no Python or CPU implementation behind it -- a missing library raises ImportError.
"""
_API_NAMES = (
    "HapCompressorNone", "HapCompressorSnappy", "HapResult", "HapTextureFormat",
    "HapDecode", "HapEncode", "HapGetFrameTextureChunkCount", "HapGetFrameTextureCount",
    "HapGetFrameTextureFormat", "HapMaxEncodedLength", "Context", "ENCODE_FRAGMENT_INDEX", "ENCODE_COARSE_MATCHES", "ENCODE_SMALLER_FILES",
    "DECODE_IGNORE_FRAGMENT_INDEX", "DECODE_IGNORE_HALF_TILES", "DECODE_NO_BLOCK_SCAN", "KERNEL_CLASSES", "HapGpuGetFrameTextureChunkLayout", "HapGpuJoinChunkGroups", "SequenceWriter", "SequenceReader", "BufferList",
    "encode_frames_rgba_on_devices", "decode_frames_on_devices", "ENCODE_FINE_CHUNKS", "fine_chunk_count", "DECODE_NO_FIELD_GUESS", "DECODE_GUESS_FIELDS",
)


def __getattr__(name):
    if name in _API_NAMES or name in ("api", "_lib"):
        import importlib
        module = importlib.import_module("." + ("_lib" if name == "_lib" else "api"), __name__)
        if name in ("api", "_lib"):
            return module
        return getattr(module, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
