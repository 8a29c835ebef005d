"""Deterministic synthetic frames for tests and bench.py.

Pure integer, position-hashed content so that the same (width, height, seed)
gives identical bytes on CPU and GPU tensors:

  * smooth 2-D gradients                         (compressible, short matches)
  * band-limited noise, amplitude +-8            (mostly literals)
  * flat 64x64 tiles, ~25 % of the frame         (long Snappy copies)
  * hard edges / stripes                         (endpoint-fit stress)
  * alpha = gradient with 0 / 255 plateaus

torch is used only as an array library (works on "cpu" and "cuda").
"""
import torch

SEED_BASE = 0x48415031  # "HAP1"

_M32 = 0xFFFFFFFF


def _mix(h):
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & _M32
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & _M32
    h = h ^ (h >> 15)
    return h


def _hash2(x, y, seed):
    return _mix((x * 0x9E3779B1 + y * 0x85EBCA77 + (seed & _M32) * 0x1B873593) & _M32)


def rgba_frame(width, height, frame_index=0, device="cpu", seed=SEED_BASE):
    """uint8 tensor [height, width, 4] (R,G,B,A)."""
    s = (seed + frame_index) & _M32
    y = torch.arange(height, dtype=torch.int64, device=device).view(-1, 1).expand(height, width)
    x = torch.arange(width, dtype=torch.int64, device=device).view(1, -1).expand(height, width)
    shift = (frame_index * 7) % max(1, width)
    xs = (x + shift) % width

    # gradients
    r = (xs * 255) // max(1, width - 1)
    g = (y * 255) // max(1, height - 1)
    b = ((xs + y) * 255) // max(1, width + height - 2)

    tile = _hash2(x >> 6, y >> 6, s)
    kind = tile & 7                      # per-64x64-tile content class

    # band-limited noise: hash at half resolution, +-8
    n = _hash2(x >> 1, y >> 1, s ^ 0x5bd1e995)
    nr = (n & 15) - 8
    ng = ((n >> 4) & 15) - 8
    nb = ((n >> 8) & 15) - 8
    noisy = (kind == 2) | (kind == 3)
    r = torch.where(noisy, r + nr, r)
    g = torch.where(noisy, g + ng, g)
    b = torch.where(noisy, b + nb, b)

    # flat tiles (kind 0,1 => 25 %)
    flat = kind < 2
    r = torch.where(flat, (tile >> 8) & 255, r)
    g = torch.where(flat, (tile >> 16) & 255, g)
    b = torch.where(flat, (tile >> 24) & 255, b)

    # hard edges: 8-pixel stripes with a diagonal cut
    edge = kind == 4
    stripe = (((x + y) >> 3) & 1) == 1
    r = torch.where(edge & stripe, 255 - r, r)
    g = torch.where(edge & stripe, 255 - g, g)
    b = torch.where(edge & ~stripe, b // 4, b)

    a = (x * 255) // max(1, width - 1)
    a = torch.where(kind == 5, torch.zeros_like(a), a)
    a = torch.where(kind == 6, torch.full_like(a, 255), a)

    out = torch.stack([r, g, b, a], dim=-1).clamp_(0, 255).to(torch.uint8)
    return out.contiguous()


def texture_like_bytes(nbytes, kind, seed=SEED_BASE, device="cpu"):
    """Adversarial byte streams for the Snappy stage alone:
    'zero', 'random' (incompressible -> store-raw path, hap.c:460-466),
    'mixed' (alternating compressible / incompressible 64 KiB runs),
    'runs' (short repeated 8/16-byte blocks, DXT-like)."""
    i = torch.arange(nbytes, dtype=torch.int64, device=device)
    if kind == "zero":
        return torch.zeros(nbytes, dtype=torch.uint8, device=device)
    h = _hash2(i, i >> 7, seed)
    if kind == "random":
        return (h & 255).to(torch.uint8)
    if kind == "mixed":
        rnd = h & 255
        blk = (i >> 16) & 1
        pat = ((i >> 3) * 37 + (i & 7) * ((i >> 12) & 3)) & 255
        return torch.where(blk == 1, rnd, pat).to(torch.uint8)
    if kind == "runs":
        blockid = i >> 4
        src = blockid - (_hash2(blockid, blockid >> 3, seed) & 3)     # repeat one of the last 4 blocks
        v = _hash2(src >> 2, i & 15, seed ^ 0x1234567)
        return (v & 255).to(torch.uint8)
    raise ValueError(kind)
