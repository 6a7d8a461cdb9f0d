#!/usr/bin/env python3
"""Measured HBM ceilings of this GPU with plain torch kernels: write-only (fill), read-only (sum), copy (1:1)."""
import torch

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

GIB = 1 << 30
x = torch.empty(2 * GIB // 4, dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
x.fill_(1)
print("| pattern | bytes moved | ms | TB/s |")
print("|---|---|---|---|")
for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(3), 2 * GIB), ("sum (read only)", lambda: x.sum(), 2 * GIB),
                         ("copy (read + write)", lambda: y.copy_(x), 4 * GIB)):
    t = timeit(fn)
    print(f"| {name} | {nbytes / 1e9:.2f} GB | {t * 1e3:.3f} | {nbytes / t / 1e12:.2f} |")
