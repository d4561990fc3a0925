"""D2H of ~99 MB into pinned memory: one copy, or split over n streams (several SDMA engines)?  (dev tool, GPU box)"""
import time
import torch
n = 99_389_745
src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
for parts in (1, 2, 3, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cuts = [n * i // parts for i in range(parts + 1)]
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[cuts[i]:cuts[i + 1]].copy_(src[cuts[i]:cuts[i + 1]], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%d stream(s): %.3f ms = %.1f GB/s" % (parts, best * 1e3, n / best / 1e9))
assert torch.equal(dst, src.cpu())
