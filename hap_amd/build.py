"""In-tree build of hap_amd/libhap_amd.so (hipcc, gfx950). No JIT cache: the
built library sits next to the sources so it travels with the repo snapshot."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhap_amd.so")


def build(verbose=False, jobs=8):
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hap_amd: building libhap_amd.so failed (hipcc --offload-arch=gfx950)")
    return LIB
