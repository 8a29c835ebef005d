"""The C ABI as a C compiler sees it: include/*.h compile as strict C99 and a small client links against
libhap_amd.so and runs its host-only checks (on a GPU box it also decodes one frame through hap.h)."""
import os
import subprocess
import sys

import pytest

import _libs as L

ROOT = L.ROOT


def _build(tmp_path, make_library=True):
    if make_library or not os.path.exists(os.path.join(ROOT, "hap_amd", "libhap_amd.so")):
        from hap_amd.build import build
        build()
    exe = str(tmp_path / "abi_client")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_client.c"), "-o", exe,
           "-L", os.path.join(ROOT, "hap_amd"), "-lhap_amd", "-Wl,-rpath," + os.path.join(ROOT, "hap_amd")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def _run(exe, tmp_path):
    env = dict(os.environ)
    # the library's own HIP runtime (plain C clients link /opt/rocm; PyTorch's bundled copy is not involved here)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return subprocess.run([exe, str(tmp_path / "client.hapseq")], capture_output=True, text=True, env=env, timeout=120)


def test_c99_client_compiles_links_and_runs_host_checks(tmp_path):
    exe = _build(tmp_path)
    done = _run(exe, tmp_path)
    assert done.returncode == 0 and done.stdout.strip() == "ok", (done.returncode, done.stdout, done.stderr[-500:])


@pytest.mark.gpu
def test_c99_client_decodes_on_the_gpu(tmp_path):
    """On a GPU box the same client also goes HapEncode -> HapDecode over an 8-chunk Snappy frame with a callback that
    runs the items backwards and decodes a second frame from inside the callback (re-entrancy of hap.h).
    The library is used as shipped (no rebuild on the GPU box)."""
    exe = _build(tmp_path, make_library=False)
    done = _run(exe, tmp_path)
    assert done.returncode == 0 and done.stdout.strip() == "ok", (done.returncode, done.stdout, done.stderr[-500:])
