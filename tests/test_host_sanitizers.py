"""Host-side container code under AddressSanitizer + UBSan: hap_frame.c (section parser, planner), hap_join.c and
the file part of hap_sequence.c are compiled with -fsanitize=address,undefined into a small C harness
(tests/c/host_fuzz.c) that mutates checker-made frames thousands of times and runs every parser over them."""
import os
import subprocess

import _data as D
import _libs as L

ROOT = L.ROOT


def test_container_code_is_clean_under_asan_and_ubsan(tmp_path):
    ora = L.oracle_api()
    frames = []
    tex = D.stream_bytes(16 * 64 * 40, "mixed", seed=3)
    frames.append(ora.encode([tex], [L.FMT_DXT5], [1], [6])[1])                                   # chunked, compressed
    frames.append(ora.encode([tex[:4096]], [L.FMT_DXT1], [0], [1])[1])                             # stored as-is
    frames.append(ora.encode([tex, tex[: len(tex) // 2]], [L.FMT_YCOCG, L.FMT_RGTC1], [1, 1], [3, 2])[1])   # two textures
    # a frame of ours would carry the private fragment table: add one by hand to a checker-made frame
    f = bytearray(ora.encode([tex], [L.FMT_YCOCG], [1], [2])[1])
    hdr = 4 if f[0:3] != b"\\0\\0\\0" else 8
    ilen = int.from_bytes(f[hdr:hdr + 3], "little")
    table = bytes([12, 0, 0, 0x46, 1, 13, 1, 12]) + (1000).to_bytes(4, "little") + (2000).to_bytes(4, "little")
    body = bytes(f[hdr + 4: hdr + 4 + ilen]) + table
    inner = len(body).to_bytes(3, "little") + bytes([1]) + body + bytes(f[hdr + 4 + ilen:])
    frames.append(len(inner).to_bytes(3, "little") + bytes([f[3]]) + inner)
    # section lengths that wrap the reference's 32-bit bound check (hap.c:160-181): once a hang / a 4 GiB plan
    huge = (0xFFFFFFF8).to_bytes(4, "little")
    frames.append((16).to_bytes(3, "little") + b"\x0d" + b"\0\0\0\xcf" + huge + bytes(8))
    frames.append(b"\0\0\0\xcb" + (0xFFFFFFF9).to_bytes(4, "little") + bytes(8))
    paths = []
    for i, fr in enumerate(frames):
        p = tmp_path / ("frame%d.bin" % i)
        p.write_bytes(fr)
        paths.append(str(p))
    exe = str(tmp_path / "host_fuzz")
    csrc = os.path.join(ROOT, "hap_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-g", "-O1", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-I", csrc, "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "host_fuzz.c"), os.path.join(csrc, "hap_frame.c"),
           os.path.join(csrc, "hap_join.c"), os.path.join(csrc, "hap_sequence.c"), "-o", exe, "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    done = subprocess.run([exe, str(tmp_path / "scratch.hapseq")] + paths, capture_output=True, text=True, env=env, timeout=600)
    assert done.returncode == 0 and done.stdout.strip().endswith("ok"), (done.returncode, done.stderr[-3000:])
