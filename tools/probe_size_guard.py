"""GPU probe (r06): the size of this library's frames against the reference encoder's on the picture of
tests/test_gpu_parity.py::test_encode_compression_ratio_close_to_libsnappy, per format.   python tools/probe_size_guard.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _libs as L, hap_amd
ORA = L.oracle_api()
img = D.rgba(2048, 512, frame=0)
for fmt in (L.FMT_DXT1, L.FMT_DXT5, L.FMT_YCOCG, L.FMT_RGTC1):
    tex = D.oracle_bc_encode(img, fmt)
    ours = len(hap_amd.HapEncode([tex], [fmt], [1], [8])[1])
    theirs = len(ORA.encode([tex], [fmt], [1], [8])[1])
    print("%#x ours %d theirs %d ratio %.3f" % (fmt, ours, theirs, ours / theirs))
