#!/usr/bin/env python3
"""Cost model of mspa_inflate_blocks_device by kind of stream: time per output byte / per compressed bit for literal-only,
match-only and mixed payloads, one wave per SIMD or several.
    python tools/device_inflate_kinds.py [--streams 640]"""
import argparse
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=640)
    ap.add_argument("--n", type=int, default=614880)
    a = ap.parse_args()
    import torch
    from mspa import engine
    rng = np.random.default_rng(1)
    N = a.n
    depth_rows = (np.cumsum(rng.integers(-3, 4, N // 2)) % 4000 + 500).astype(">u2").tobytes()
    kinds = {
        "literals_8bit (random bytes, Z_HUFFMAN_ONLY)": (rng.integers(0, 256, N, dtype=np.uint8).tobytes(), zlib.Z_HUFFMAN_ONLY, 6),
        "literals_4bit (16 symbols, Z_HUFFMAN_ONLY)": (rng.integers(0, 16, N, dtype=np.uint8).tobytes(), zlib.Z_HUFFMAN_ONLY, 6),
        "runs (zeros: length-258 matches at distance 1)": (bytes(N), zlib.Z_DEFAULT_STRATEGY, 6),
        "period 100 (length-258 matches at distance 100)": ((bytes(range(100)) * (N // 100 + 1))[:N], zlib.Z_DEFAULT_STRATEGY, 6),
        "short matches (4 random bytes, then the same 4 again)": (
            b"".join(bytes(x) * 2 for x in rng.integers(0, 256, (N // 8 + 1, 4), dtype=np.uint8))[:N], zlib.Z_DEFAULT_STRATEGY, 6),
        "depth-like rows": (depth_rows, zlib.Z_DEFAULT_STRATEGY, 6),
    }
    for name, (data, strat, level) in kinds.items():
        c = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strat)
        comp = c.compress(data) + c.flush()
        slot = (len(comp) + 15) // 16 * 16 + 16
        one = np.zeros(slot, dtype=np.uint8)
        one[:len(comp)] = np.frombuffer(comp, dtype=np.uint8)
        src = torch.from_numpy(one).cuda().repeat(a.streams)
        off = torch.arange(a.streams, dtype=torch.int64, device="cuda") * slot
        nb = torch.full((a.streams,), len(comp), dtype=torch.int64, device="cuda")
        raw, st = engine.inflate_blocks_device(src, off, nb, N)
        torch.cuda.synchronize()
        ok = bool((st == 0).all()) and raw[0, :N].cpu().numpy().tobytes() == data
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            engine.inflate_blocks_device(src, off, nb, N, raw, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(name, json.dumps({"ok": ok, "compressed_bytes": len(comp), "ms": round(ms, 2),
                                "ns_per_output_byte_per_wave": round(ms * 1e6 / N, 1),
                                "ns_per_compressed_bit_per_wave": round(ms * 1e6 / (len(comp) * 8), 1)}), flush=True)


if __name__ == "__main__":
    main()
