"""Pinned host -> device copy rate on this box: one stream, two streams, chunk sizes (run with and without HSA_ENABLE_SDMA=0)."""
import os, time, torch
n = 196_608_000
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
def run(n_streams, chunk):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    def once():
        k = 0
        for lo in range(0, n, chunk):
            with torch.cuda.stream(streams[k % n_streams]):
                d[lo:lo + chunk].copy_(h[lo:lo + chunk], non_blocking=True)
            k += 1
        torch.cuda.synchronize()
    once()
    t = time.perf_counter()
    for _ in range(5): once()
    return n * 5 / (time.perf_counter() - t) / 1e9
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
for ns, chunk in ((1, n), (1, n // 5), (2, n // 4), (2, n // 16), (4, n // 16)):
    print(f"  streams {ns} chunk {chunk / 1e6:.0f} MB: {run(ns, chunk):.1f} GB/s")
