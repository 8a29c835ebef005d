"""Dev tool: bench.py's c5_chunk_groups on ONE rank (RCCL world of 1): exercises the gather / device join / broadcast /
group decode code path without an 8-GPU node."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import bench, hap_amd
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
ctx = hap_amd.Context(0)
def fence():
    torch.cuda.synchronize(); ctx.synchronize(); dist.barrier(); torch.cuda.synchronize()
print(json.dumps(bench.c5_chunk_groups(hap_amd, ctx, dist, dev, 0, 1, fence)))
dist.destroy_process_group()
