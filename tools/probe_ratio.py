"""Dev probe: compression ratio of the GPU compressor vs libsnappy (oracle) on synthetic textures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _libs as L, hap_amd
ORA = L.oracle_api()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 512)
img = D.rgba(w, h, 0)
ctx = hap_amd.Context(0)
for lg in (12, 14, 16):
    os.environ["HAP_AMD_FRAGMENT_LOG2"] = str(lg)
    import ctypes
    hap_amd._lib.lib.HapGpuSetFragmentLog2(hap_amd._lib.lib.HapGpuDefaultContext(), lg)
    for fmt in [L.FMT_DXT1, L.FMT_DXT5, L.FMT_YCOCG, L.FMT_RGTC1]:
        tex = D.oracle_bc_encode(img, fmt)
        ours = len(hap_amd.HapEncode([tex], [fmt], [1], [8])[1])
        theirs = len(ORA.encode([tex], [fmt], [1], [8])[1])
        print("frag 2^%d fmt %#06x bytes %d ours %.4f libsnappy %.4f" % (lg, fmt, len(tex), ours / len(tex), theirs / len(tex)))
