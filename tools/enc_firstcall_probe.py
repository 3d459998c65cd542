"""First vs second call of netG.filter by batch size, with allocator statistics (what a Coalesced filter stage
pays the first time it meets a batch size)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
netG, _ = bench.build_netg(dev)
for b in (1, 8, 7, 8, 3, 8, 1):
    x = torch.zeros((b, 3, 512, 512), device=dev)
    for rep in range(2):
        torch.cuda.synchronize()
        s0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        with torch.no_grad():
            netG.filter(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        s1 = torch.cuda.memory_stats()
        print("batch %d call %d: host %.1f ms, total %.1f ms; device allocs +%d frees +%d; reserved %.2f GB"
              % (b, rep, 1e3 * (t1 - t0), 1e3 * (t2 - t0), s1["num_device_alloc"] - s0["num_device_alloc"],
                 s1["num_device_free"] - s0["num_device_free"], torch.cuda.memory_reserved() / 2 ** 30), flush=True)
