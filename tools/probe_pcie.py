"""Dev probe: raw host<->device copy rates on the GPU box (torch, pinned vs pageable)."""
import time, torch
n = 132710400
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, h in (("pageable", torch.empty(n, dtype=torch.uint8)), ("pinned", torch.empty(n, dtype=torch.uint8).pin_memory())):
    h.fill_(7)
    for direction in ("h2d", "d2h"):
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            if direction == "h2d":
                d.copy_(h, non_blocking=True)
            else:
                h.copy_(d, non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(name, direction, "%.1f GB/s" % (n / dt / 1e9))
