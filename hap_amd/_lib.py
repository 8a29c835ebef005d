"""ctypes binding of the C ABI declared in include/hap.h and include/hap_gpu.h.

There is no Python or CPU implementation behind this module: if
hap_amd/libhap_amd.so (the HIP build) is missing, importing fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# HAP_AMD_LIBRARY: development override to load an experimental build of the same library
LIB_PATH = os.environ.get("HAP_AMD_LIBRARY") or os.path.join(HERE, "libhap_amd.so")

WORK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint)
CALLBACK = C.CFUNCTYPE(None, WORK_FN, C.c_void_p, C.c_uint, C.c_void_p)


def _load():
    # PyTorch wheels bundle their own libamdhip64/libhsa-runtime64 (same SONAMEs as /opt/rocm's).
    # Two HIP runtimes in one process fight over the device, so when torch is installed let it
    # load its runtime first; libhap_amd.so then binds to that already-loaded copy.  Plain C
    # clients link /opt/rocm directly and are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "hap_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). hap_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u, ul = C.c_void_p, C.c_uint, C.c_ulong
    P = C.POINTER
    sig = {
        "HapMaxEncodedLength": (ul, [u, P(ul), P(u), P(u)]),
        "HapEncode": (u, [u, P(vp), P(ul), P(u), P(u), P(u), vp, ul, P(ul)]),
        "HapDecode": (u, [vp, ul, u, CALLBACK, vp, vp, ul, P(ul), P(u)]),
        "HapGetFrameTextureCount": (u, [vp, ul, P(u)]),
        "HapGetFrameTextureFormat": (u, [vp, ul, u, P(u)]),
        "HapGetFrameTextureChunkCount": (u, [vp, ul, u, P(C.c_int)]),
        "HapGpuCreate": (u, [C.c_int, P(vp)]),
        "HapGpuDestroy": (None, [vp]),
        "HapGpuDefaultContext": (vp, []),
        "HapGpuSetFragmentLog2": (u, [vp, u]),
        "HapGpuSynchronize": (u, [vp]),
        "HapGpuFineChunkCount": (u, [ul, u]),
        "HapGpuTableFallbackCount": (ul, [vp]),
        "HapGpuPlacementRetryCount": (ul, [vp]),
        "HapGpuResolvedBlockCount": (ul, [vp]),
        "HapGpuPlacementTimeoutCount": (ul, [vp]),
        "HapGpuCompressRGBA": (u, [vp, vp, u, u, ul, u, vp, ul, P(ul)]),
        "HapGpuDecompressRGBA": (u, [vp, vp, ul, u, vp, ul, u, u, vp, ul]),
        "HapGpuEncodeFrames": (u, [vp, u, u, P(vp), P(ul), P(u), P(u), P(u), P(vp), P(ul), P(ul), P(u), u]),
        "HapGpuEncodeFramesRGBA": (u, [vp, u, P(vp), u, u, ul, u, P(u), P(u), P(u), P(vp), P(ul), P(ul), P(u), u]),
        "HapGpuEncodeFramesRGBABegin": (u, [vp, u, P(vp), u, u, ul, u, P(u), P(u), P(u), P(vp), P(ul), P(ul), P(u), u]),
        "HapGpuEncodeFramesBegin": (u, [vp, u, u, P(vp), P(ul), P(u), P(u), P(u), P(vp), P(ul), P(ul), P(u), u]),
        "HapGpuEncodeFramesFinish": (u, [vp]),
        "HapGpuEncodeFramesRGBAOnDevices": (u, [P(vp), u, u, P(vp), u, u, ul, u, P(u), P(u), P(u), P(vp), P(ul), P(ul), P(u), u]),
        "HapGpuEncodeFramesOnDevices": (u, [P(vp), u, u, u, P(vp), P(ul), P(u), P(u), P(u), P(vp), P(ul), P(ul), P(u), u]),
        "HapGpuDecodeFramesOnDevices": (u, [P(vp), u, u, P(vp), P(ul), u, P(vp), P(ul), P(ul), P(u), P(u), u]),
        "HapGpuDecodeFrames": (u, [vp, u, P(vp), P(ul), u, P(vp), P(ul), P(ul), P(u), P(u), u]),
        "HapGpuDecodeFrameTextures": (u, [vp, u, P(vp), P(ul), u, P(vp), P(ul), P(ul), P(u), P(u), u]),
        "HapGpuDecodeFramesRGBA": (u, [vp, u, P(vp), P(ul), u, P(vp), u, u, ul, P(u), u]),
        "HapGpuDecodeChunkGroup": (u, [vp, vp, ul, u, u, u, vp, ul, P(ul), P(u)]),
        "HapGpuGetFrameTextureChunkLayout": (u, [vp, ul, u, u, P(ul), P(u)]),
        "HapGpuJoinChunkGroups": (u, [u, P(vp), P(ul), vp, ul, P(ul)]),
        "HapGpuJoinChunkGroupsDevice": (u, [vp, u, P(vp), P(ul), vp, ul, P(ul)]),
        "HapSequenceWriterOpen": (u, [C.c_char_p, u, u, u, u, P(vp)]),
        "HapSequenceWriterAppend": (u, [vp, vp, ul]),
        "HapSequenceWriterClose": (u, [vp]),
        "HapSequenceReaderOpen": (u, [C.c_char_p, P(vp)]),
        "HapSequenceReaderClose": (None, [vp]),
        "HapSequenceReaderInfo": (u, [vp, P(u), P(u), P(u), P(u), P(u)]),
        "HapSequenceReaderFrameBytes": (ul, [vp, u]),
        "HapSequenceReaderRead": (u, [vp, u, u, vp, ul, P(ul)]),
        "HapGpuDecodeSequence": (u, [vp, vp, u, u, u, u, P(vp), P(ul), P(ul), P(u), P(u)]),
        "HapGpuEncodeSequence": (u, [vp, vp, u, P(vp), u, u, ul, u, P(u), P(u), P(u), u, u, P(ul), P(u)]),
        "HapGpuSetProfiling": (u, [vp, u]),
        "HapGpuCollectProfile": (u, [vp, P(ul), P(C.c_double)]),
        "HapGpuCollectProfileN": (u, [vp, u, P(ul), P(C.c_double)]),
        "HapGpuTimerStart": (u, [vp]),
        "HapGpuTimerStop": (u, [vp, P(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
