"""Dev fuzz campaign for the block encoders / decoders: adversarial pixel classes, random geometry and
row strides; GPU output must equal the CPU checker's bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _libs as L, hap_amd
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
c = hap_amd.Context(0)
FORMATS = [L.FMT_DXT1, L.FMT_DXT5, L.FMT_YCOCG, L.FMT_RGTC1]


def picture(w, h, kind):
    if kind == 0:
        return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    if kind == 1:      # near-flat: tiny variance around a random colour per block
        base = np.repeat(np.repeat(rng.integers(0, 256, (h // 4, w // 4, 4)), 4, 0), 4, 1)
        return np.clip(base + rng.integers(-2, 3, (h, w, 4)), 0, 255).astype(np.uint8)
    if kind == 2:      # two colours per block
        a = np.repeat(np.repeat(rng.integers(0, 256, (h // 4, w // 4, 4)), 4, 0), 4, 1)
        b = np.repeat(np.repeat(rng.integers(0, 256, (h // 4, w // 4, 4)), 4, 0), 4, 1)
        m = rng.integers(0, 2, (h, w, 1))
        return np.where(m, a, b).astype(np.uint8)
    if kind == 3:      # saturated / extreme values
        return rng.choice(np.array([0, 1, 127, 128, 254, 255], dtype=np.uint8), (h, w, 4))
    if kind == 4:      # smooth gradients with random slopes per channel
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([(x * rng.integers(-3, 4) + y * rng.integers(-3, 4) + rng.integers(0, 256)) for _ in range(4)], -1)
        return (img & 255).astype(np.uint8)
    # one channel varies, the others are constant (degenerate covariance axes)
    img = np.zeros((h, w, 4), dtype=np.int64) + rng.integers(0, 256, 4)
    ch = int(rng.integers(0, 4))
    img[..., ch] = rng.integers(0, 256, (h, w))
    return img.astype(np.uint8)


fails = 0
t0 = time.time()
for it in range(N):
    w = 4 * int(rng.integers(1, 80)); h = 4 * int(rng.integers(1, 40))
    img = picture(w, h, it % 6)
    pad = int(rng.choice([0, 4, 16, 20, 64]))
    stride = w * 4 + pad
    buf = np.zeros((h, stride), dtype=np.uint8)
    buf[:, : w * 4] = img.reshape(h, w * 4)
    if pad:
        buf[:, w * 4:] = rng.integers(0, 256, (h, pad), dtype=np.uint8)
    for fmt in FORMATS:
        want = D.oracle_bc_encode(img, fmt)
        r, got = c.compress_rgba(buf, w, h, stride, fmt)
        if r != 0 or got != want:
            fails += 1
            print("ENCODE FAIL it", it, w, h, stride, hex(fmt), "kind", it % 6)
    # decoders on random blocks + optional alpha plane
    for fmt in (L.FMT_DXT1, L.FMT_DXT5, L.FMT_YCOCG):
        bpb = 8 if fmt == L.FMT_DXT1 else 16
        blocks = rng.integers(0, 256, (w // 4) * (h // 4) * bpb, dtype=np.uint8).tobytes()
        alpha = rng.integers(0, 256, (w // 4) * (h // 4) * 8, dtype=np.uint8).tobytes() if it % 2 else None
        want = D.oracle_bc_decode(blocks, fmt, w, h).copy()
        if alpha is not None:
            want[..., 3] = D.oracle_bc_decode(alpha, L.FMT_RGTC1, w, h)
        r, got = c.decompress_rgba(blocks, fmt, w, h, alpha=alpha)
        if r != 0 or np.frombuffer(got, np.uint8).reshape(h, w, 4).tobytes() != want.tobytes():
            fails += 1
            print("DECODE FAIL it", it, w, h, hex(fmt), alpha is not None)
print("block fuzz: %d cases in %.1fs, failures %d" % (N, time.time() - t0, fails))
