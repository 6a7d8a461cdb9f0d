import time, torch
n = 196_608_000
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
cs = torch.cuda.Stream()
for chunk in (n, n // 5, n // 20):
    for rep in range(2):
        torch.cuda.synchronize()
        ts = []
        t0 = time.perf_counter()
        with torch.cuda.stream(cs):
            for lo in range(0, n, chunk):
                t = time.perf_counter()
                d[lo:lo + chunk].copy_(h[lo:lo + chunk], non_blocking=True)
                ts.append((time.perf_counter() - t) * 1e3)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"chunk {chunk/1e6:.0f} MB: enqueue total {(t1-t0)*1e3:.2f} ms, then sync {(t2-t1)*1e3:.2f} ms; per call {['%.2f' % x for x in ts[:6]]}")
