#!/usr/bin/env python3
"""Out-of-suite fuzz of mspa_inflate_blocks_device (csrc/device_ingest.hip) against zlib: per round one output size N and a few
hundred streams -- every payload kind of tests/test_gpu_device_ingest.py spliced at random, random level / strategy / window /
memLevel -- plus damaged copies (bit flips, truncations, garbage tails, swapped halves).  Required: a valid stream is accepted and
equals zlib's output bit for bit; a damaged one is either refused or still equals the original (the damage hit nothing that
matters); nothing hangs.
    python tools/fuzz_device_inflate.py [--rounds 24] [--streams 256] [--seed 1]"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), os.path.join(ROOT, "tests"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=24)
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    from mspa import engine
    from test_gpu_device_ingest import _payloads, _upload
    rng = np.random.default_rng(a.seed)
    tot = {"valid": 0, "valid_accepted": 0, "damaged": 0, "damaged_refused": 0, "damaged_harmless": 0, "mismatch": 0, "valid_refused": 0}
    t0 = time.time()
    sizes = []
    for rnd in range(a.rounds):
        N = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 70001, 614880, int(rng.integers(1, 300000))]))
        sizes.append(N)
        pay = list(_payloads(max(N, 64), rng).values())
        streams, want, valid = [], [], []
        for k in range(a.streams):
            if rng.random() < 0.5:
                data = pay[int(rng.integers(len(pay)))][:N]
            else:                                                   # spliced: pieces of several kinds
                parts, left = [], N
                while left > 0:
                    n = int(min(left, rng.integers(1, max(2, N // 2 + 1))))
                    p = pay[int(rng.integers(len(pay)))]
                    o = int(rng.integers(0, max(1, len(p) - n + 1)))
                    parts.append(p[o:o + n])
                    left -= len(parts[-1])
                data = b"".join(parts)[:N]
            data = (data + bytes(N))[:N]
            c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, int(rng.integers(9, 16)), int(rng.integers(1, 10)),
                                 int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
            good = c.compress(data) + c.flush()
            kind = rng.random()
            if kind < 0.6:
                s, ok = good, True
            elif kind < 0.75:
                b = bytearray(good)
                for _ in range(int(rng.integers(1, 4))):
                    pos = int(rng.integers(0, len(b) * 8))
                    b[pos >> 3] ^= 1 << (pos & 7)
                s, ok = bytes(b), False
            elif kind < 0.85:
                s, ok = good[:int(rng.integers(0, len(good)))], False
            elif kind < 0.95:
                s, ok = good[:len(good) // 2] + rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8).tobytes(), False
            else:
                h = len(good) // 2
                s, ok = good[:2] + good[h:] + good[2:h], False
            streams.append(s)
            want.append(data)
            valid.append(ok)
        src, off, nb = _upload(streams)
        out, status = engine.inflate_blocks_device(src, off, nb, N)
        torch.cuda.synchronize()
        st, got = status.cpu().numpy(), out.cpu().numpy()
        for k in range(len(streams)):
            same = got[k, :N].tobytes() == want[k]
            if valid[k]:
                tot["valid"] += 1
                if st[k] == 0 and same:
                    tot["valid_accepted"] += 1
                elif st[k] != 0:
                    tot["valid_refused"] += 1
                else:
                    tot["mismatch"] += 1
            else:
                tot["damaged"] += 1
                if st[k] != 0:
                    tot["damaged_refused"] += 1
                elif same:
                    tot["damaged_harmless"] += 1
                else:
                    tot["mismatch"] += 1
    tot.update(rounds=a.rounds, streams_per_round=a.streams, seed=a.seed, seconds=round(time.time() - t0, 1), output_sizes=sizes)
    print(json.dumps(tot))
    sys.exit(1 if tot["mismatch"] or tot["valid_refused"] else 0)


if __name__ == "__main__":
    main()
