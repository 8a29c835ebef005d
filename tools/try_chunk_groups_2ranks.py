import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hap_amd
from hap_amd import shard, synth
dev = torch.device("cuda", 0)
ctx = hap_amd.Context(0)
w = h = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
world = 2; nchunks = 64; fmts = [0x01, 0x8DBB]
parts = []
bands = []
for rank in range(world):
    lo, hi, band_chunks = shard.band_for_rank(h // 4, nchunks, rank, world)
    rows = (hi - lo) * 4
    band = synth.rgba_frame(w, rows, 1000 + rank, device=dev)
    tex_bytes = [(w // 4) * (rows // 4) * b for b in (16, 8)]
    cap = hap_amd.HapMaxEncodedLength(tex_bytes, fmts, [band_chunks] * 2)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    r, used, res = ctx.encode_frames_rgba([band], w, rows, w * 4, fmts, [1, 1], [band_chunks] * 2, [out], flags=hap_amd.ENCODE_FRAGMENT_INDEX)
    assert r == 0, (r, res)
    parts.append(out[:used[0]].clone()); bands.append(band)
dframe = torch.empty(sum(int(p.numel()) for p in parts) + 64, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
r, joined = ctx.join_chunk_groups(parts, [int(p.numel()) for p in parts], dframe)
print("join", r, joined)
dframe = dframe[:joined]
for copy in (False, True):
    fr = dframe.clone() if copy else dframe
    for idx in (0, 1):
        r, layout = hap_amd.HapGpuGetFrameTextureChunkLayout(fr, idx)
        print("layout", r, len(layout), layout[-1])
        whole = torch.zeros(layout[-1], dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for rank in range(world):
            g = shard.chunk_group_for_rank(len(layout) - 1, rank, world)
            r, used, fmt = ctx.decode_chunk_group(fr, idx, g.start, len(g), whole)
            print("copy", copy, "tex", idx, "rank", rank, "group", g.start, len(g), "->", r, used, hex(fmt))
