#!/usr/bin/env python3
"""Where do a row group's depth-2 gathers land?  CPU-only analysis of the benchmark's own pairs (the vc workload: the
reference's overlap-binned sampler over the 64-frame sweep scene), used in round 3 to price "stage frame 2's footprint in
LDS" before building it (DESIGN.md section 4, K3 round 3, item iv).

For 40 of the 1 000 pairs: per 4 x 64-pixel row group of the tight kernel -- the share culled, the lanes in view, the distinct
dwords / 64-byte lines its gathers touch, the bounding box of the projected pixels, and how many groups a fixed box around
the middle lane would cover.
    python tools/footprint_stats.py > profiles/r03_k3_footprint_stats.md
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]
import bench  # noqa: E402
from mspa import engine, workload  # noqa: E402
from oracle import np_oracle as O  # noqa: E402

H, W = 480, 640
UNPROJ, REPROJ = 5, 6
BOXES = [(128, 16), (128, 12), (128, 8), (192, 16)]


def main():
    sys.argv = [sys.argv[0]]
    args = bench.parse_args()
    sc = bench.make_base_scene(args, 0)
    ids = sc.valid_image_ids
    masks = O.scene_visibility_masks(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, (H, W))
    n = len(ids)
    ov = np.array([O.calculate_camera_overlap(masks[ids[i]], masks[ids[j]]) for i in range(n) for j in range(i + 1, n)])
    pairs, info = workload.select_pairs(ov, n, 1000, "vc", seed=77)
    mats = engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])
    rng = np.random.default_rng(0)
    sel = rng.choice(len(pairs), 40, replace=False)
    yy, xx = np.mgrid[0:H, 0:W]
    groups = culled = rows = dwords = lines_row = inview = 0
    fit = {b: 0 for b in BOXES}
    ws, hs, lines_grp, rots = [], [], [], []
    vis_px = 0
    for p in sel:
        a, b = pairs[p]
        M = (mats[b, REPROJ].reshape(4, 4) @ mats[a, UNPROJ].reshape(4, 4))[:3]
        d = sc.depth[ids[a]].astype(np.float64)
        q = np.stack([xx * d, yy * d, d, np.full_like(d, 1000.0)], -1) @ M.T
        with np.errstate(all="ignore"):
            u, v = q[..., 0] / q[..., 2], q[..., 1] / q[..., 2]
        inv = (d > 0) & (u >= 0) & (u < W) & (v >= 0) & (v < H) & (q[..., 2] > 0)
        xi = np.clip(np.rint(np.nan_to_num(u, nan=0, posinf=1e9, neginf=-1e9)), 0, W - 1).astype(np.int64)
        yi = np.clip(np.rint(np.nan_to_num(v, nan=0, posinf=1e9, neginf=-1e9)), 0, H - 1).astype(np.int64)
        vis_px += int((inv & (q[..., 2] < sc.depth[ids[b]].astype(np.float64)[yi, xi])).sum())
        R = sc.E[ids[a]][:3, :3].T @ sc.E[ids[b]][:3, :3]
        rots.append(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))
        for r0 in range(0, H, 4):
            for c0 in range(0, W, 64):
                m = inv[r0:r0 + 4, c0:c0 + 64]
                groups += 1
                if not m.any():
                    culled += 1
                    continue
                x, y = xi[r0:r0 + 4, c0:c0 + 64], yi[r0:r0 + 4, c0:c0 + 64]
                for j in range(4):
                    dwords += len(np.unique(y[j] * 1024 + (x[j] >> 1)))
                    lines_row += len(np.unique(y[j] * 64 + (x[j] >> 5)))
                    rows += 1
                inview += int(m.sum())
                ws.append(int(x[m].max() - x[m].min() + 1))
                hs.append(int(y[m].max() - y[m].min() + 1))
                lines_grp.append(len(np.unique(y[m] * 64 + (x[m] >> 5))))
                jr = next(j for j in (1, 2, 0, 3) if m[j].any())
                lanes = np.nonzero(m[jr])[0]
                hi = lanes[lanes >= 32]
                L = hi[0] if len(hi) else lanes[lanes < 32][-1]
                xr, yr = x[jr, L], y[jr, L]
                for bw, bh in BOXES:
                    xlo = min(max((xr - bw // 2) & ~7, 0), W - bw)
                    ylo = min(max(yr - bh // 2, 0), H - bh)
                    fit[(bw, bh)] += bool(((x[m] >= xlo) & (x[m] < xlo + bw) & (y[m] >= ylo) & (y[m] < ylo + bh)).all())
    # the same pixels, other lane <-> pixel shapes per gather instruction (64 pixels each): distinct 64-byte lines touched
    shapes = {"64 x 1 (as built)": (1, 64), "32 x 2": (2, 32), "16 x 4": (4, 16), "8 x 8": (8, 8)}
    shape_lines = {k: [0, 0] for k in shapes}
    for p in sel[:20]:
        a, b = pairs[p]
        M = (mats[b, REPROJ].reshape(4, 4) @ mats[a, UNPROJ].reshape(4, 4))[:3]
        d = sc.depth[ids[a]].astype(np.float64)
        q = np.stack([xx * d, yy * d, d, np.full_like(d, 1000.0)], -1) @ M.T
        with np.errstate(all="ignore"):
            u, v = q[..., 0] / q[..., 2], q[..., 1] / q[..., 2]
        inv = (d > 0) & (u >= 0) & (u < W) & (v >= 0) & (v < H) & (q[..., 2] > 0)
        xi = np.clip(np.rint(np.nan_to_num(u, nan=0, posinf=1e9, neginf=-1e9)), 0, W - 1).astype(np.int64)
        yi = np.clip(np.rint(np.nan_to_num(v, nan=0, posinf=1e9, neginf=-1e9)), 0, H - 1).astype(np.int64)
        for r0 in range(0, H, 8):
            for c0 in range(0, W, 64):
                if not inv[r0:r0 + 8, c0:c0 + 64].any():
                    continue
                for name, (bh, bw) in shapes.items():
                    for rb in range(r0, r0 + 8, bh):
                        for cb in range(c0, c0 + 64, bw):
                            x, y = xi[rb:rb + bh, cb:cb + bw].reshape(-1), yi[rb:rb + bh, cb:cb + bw].reshape(-1)
                            shape_lines[name][0] += len(np.unique(y * 64 + (x >> 5)))
                            shape_lines[name][1] += 1
    ng = groups - culled
    pct = lambda a, q: np.percentile(a, q).round(1).tolist()
    print("# Depth-2 gather footprints of the tight kernel's row groups (tools/footprint_stats.py, CPU analysis)\n")
    print(f"Workload: {info['rule']}; 40 of the 1 000 pairs; relative rotation of the two views (degrees) 10 / 50 / 90 th percentile: {pct(rots, [10, 50, 90])}\n")
    print("| quantity | value |\n|---|---|")
    print(f"| visible fraction of the pixels | {vis_px / (len(sel) * H * W):.3f} |")
    print(f"| 4 x 64 groups with no lane in view (culled by the early-out) | {culled / groups:.3f} |")
    print(f"| lanes in view per surviving group | {inview / ng / 256:.3f} |")
    print(f"| distinct dwords per 64-lane gather (what the address unit coalesces to) | {dwords / rows:.1f} |")
    print(f"| distinct 64-byte lines per 64-lane gather | {lines_row / rows:.1f} |")
    print(f"| distinct 64-byte lines per group (4 gathers) | mean {np.mean(lines_grp):.1f}, 50 / 90 th pct {pct(lines_grp, [50, 90])} |")
    print(f"| bounding box of a group's projections: width, 50 / 75 / 90 / 95 th pct | {pct(ws, [50, 75, 90, 95])} |")
    print(f"| bounding box of a group's projections: height, 50 / 75 / 90 / 95 th pct | {pct(hs, [50, 75, 90, 95])} |")
    for (bw, bh), c in fit.items():
        print(f"| groups covered by a {bw} x {bh} box around the middle in-view lane ({bw * bh * 2} B of LDS, {bw * bh * 2 // 64} line requests) | {c / ng:.3f} |")
    print("\nDistinct 64-byte lines per 64-lane gather for other lane <-> pixel shapes (8 x 64 blocks with a lane in view; the front end's")
    print("cost of a gather is ~14 + 1.2 cycles per distinct line when everything hits: tools/ubench/gather_rates.hip, profiles/r03_gather_rates.txt):\n")
    print("| pixels of one gather instruction | distinct lines | modelled front-end cycles |\n|---|---|---|")
    for name, (l, n_) in shape_lines.items():
        print(f"| {name} | {l / n_:.1f} | {14 + 1.2 * l / n_:.0f} |")


if __name__ == "__main__":
    main()
