"""hap_amd -- MI355X-native Hap frame encode/decode hot path.

The product is the C-ABI shared library hap_amd/libhap_amd.so (headers in
include/): HIP kernels for gfx950 behind the reference's hap.h API.  This
Python package is a thin ctypes mirror of that API for tests and bench.py.
"""
from .api import (  # noqa: F401
    HapCompressorNone, HapCompressorSnappy, HapResult, HapTextureFormat,
    HapDecode, HapEncode, HapGetFrameTextureChunkCount, HapGetFrameTextureCount,
    HapGetFrameTextureFormat, HapMaxEncodedLength, Context, ENCODE_FRAGMENT_INDEX,
    DECODE_IGNORE_FRAGMENT_INDEX, KERNEL_CLASSES,
)
