"""pinned host -> device copy rates on this box: what bounds e2e_with_h2d"""
import time
import torch

dev = torch.device("cuda", 0)
for mib in (32, 128, 512):
    for nstreams in (1, 2, 3, 4):
        n = mib << 20
        hosts = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(nstreams)]
        devs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        reps = max(2, (2 << 30) // (n * nstreams))
        for _ in range(2):
            for s, h, d in zip(streams, hosts, devs):
                with torch.cuda.stream(s):
                    d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            for s, h, d in zip(streams, hosts, devs):
                with torch.cuda.stream(s):
                    d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("%4d MiB x %d streams: %6.1f GB/s" % (mib, nstreams, reps * nstreams * n / dt / 1e9), flush=True)
