#!/usr/bin/env python3
"""CPU emulation of the fast kernels' guard AS A BOUND ("guard") against the NumPy oracle.

What the kernels do since round 4, restated in NumPy: per tile, from slot MSPA_MAT_BOUNDS of the two frame records and the
tile's largest depth sample, B_k = C * 2^-53 * (nr_k * wmax + nt_k) bounds the difference between ANY two float64
evaluation orders of the homogeneous image coordinate k (pixel * millimetre); lanes whose camera-2 depth is below
zmin = 2 (B_0 + B_1 + (max(W, H) + 1) B_2) / 1e-6 or whose depth test is closer than gz = 2 B_2 + 1e-6 mm, and
lanes within 1e-6 px of a rounding tie or an integer, take the reference chain.  This tool checks, on adversarial pairs
(the recipe of tests/test_gpu_tight.py), on pairs whose second camera is centred delta in {1e-9 .. 1e-3} m behind a
back-projected frame-1 point, and on a scene translated by 1e4 m:
  (a) |q_fast - q_ref| <= B_k for every valid lane (the bound holds), and by what factor;
  (b) every decision (in view, xi, yi, visible) of the guarded fast path equals the oracle's;
  (c) the same with round 3's constant guard (1e-6 px, 1e-6 mm) -- where it fails.
    python tools/guard_bound_emulation.py > profiles/r04_guard_bound_emulation.md
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT, os.path.join(ROOT, "tests")]
from mspa import engine, synth, _lib  # noqa: E402
from oracle import np_oracle as O  # noqa: E402

EPS = 2.0 ** -53
C_BOUND = 256.0
G_PX = 1e-6
G_ZMM = 1e-6
TH, TW = 48, 64


import adversarial as ADV  # noqa: E402  (tests/adversarial.py: the generators the GPU tests use)


def ref_homogeneous(depth1, K, E1, E2, A, hw):
    """The reference chain's homogeneous image coordinates (before IH:69's division), pixel * metre, per colour pixel."""
    H, W = hw
    mask = np.ones((H, W), dtype=bool)
    pts, my, mx = O.project_mask_to_3d(depth1, K, E1, mask, A, None, return_index=True)
    E2a = O.aligned_extrinsic(A, E2)
    cam = np.linalg.inv(E2a) @ O.homogeneous(pts[:, :3]).T
    img = K @ cam
    q = np.full((H * W, 3), np.nan)
    q[my * W + mx] = img[:3].T
    return q.reshape(H, W, 3)


def emulate_pair(depth1, depth2, K, E1, E2, A, hw, m1, m2, stats, constant_guard=False):
    H, W = hw
    yy, xx = np.mgrid[0:H, 0:W]
    ref = O.frame_pair(depth1, depth2, K, E1, E2, A, hw)
    qref = ref_homogeneous(depth1, K, E1, E2, A, hw) * 1000.0             # pixel * millimetre
    M = (m2[_lib.MAT_REPROJ].reshape(4, 4) @ m1[_lib.MAT_UNPROJ].reshape(4, 4))[:3].copy()
    M[:, 3] *= 1000.0
    d = depth1.astype(np.float64)
    valid = d > 0
    # the tight kernel's form since round 4: rows 0 / 1 centred on the image and scaled by its half extent (+ guard); "in view"
    # is |xn| < z on the homogeneous coordinates, the division happens only for candidates
    khw, khh = W / 2 + G_PX, H / 2 + G_PX
    Mc = M.copy()
    Mc[0] = (M[0] - (W / 2) * M[2]) * (1.0 / khw)
    Mc[1] = (M[1] - (H / 2) * M[2]) * (1.0 / khh)
    t = Mc[:, 0] * xx[..., None] + Mc[:, 1] * yy[..., None] + Mc[:, 2]
    qc = t * d[..., None] + Mc[:, 3]
    z = qc[..., 2]
    with np.errstate(all="ignore"):
        rz = 1.0 / z
        u = qc[..., 0] * (rz * khw) + W / 2          # un-centred only for the comparison with the oracle below
        v = qc[..., 1] * (rz * khh) + H / 2
        homog_in = (np.abs(qc[..., 0]) < z) & (np.abs(qc[..., 1]) < z)
    q = np.stack([qc[..., 0] * khw + (W / 2) * z, qc[..., 1] * khh + (H / 2) * z, z], -1)     # for the bound check
    b1, b2 = m1[_lib.MAT_BOUNDS], m2[_lib.MAT_BOUNDS]
    dv_of = depth2.astype(np.float64)
    guard = np.zeros((H, W), dtype=bool)
    for R0 in range(0, H, TH):
        for c0 in range(0, W, TW):
            sl = (slice(R0, R0 + TH), slice(c0, c0 + TW))
            dhi = d[sl].max()
            if dhi == 0:
                continue
            wmax = dhi * (b1[0] * (c0 + TW - 1) + b1[1] * (R0 + TH - 1) + b1[2]) + b1[3]               # millimetres
            B = b2[4:7] * wmax + b2[8:11]
            if constant_guard:
                zmin, gz = G_ZMM, G_ZMM
            else:
                zmin = 2.0 * (B[0] + B[1] + (max(W, H) + 1.0) * B[2]) / G_PX      # mspa_common.h guard_from_bounds
                gz = 2.0 * B[2] + G_ZMM
                stats["zmin_max"] = max(stats["zmin_max"], zmin)
                stats["zmin_sum"] += zmin
                stats["tiles"] += 1
                with np.errstate(all="ignore"):
                    err = np.abs(q[sl] - qref[sl])
                    ratio = np.where(valid[sl][..., None], err / B, 0.0)
                stats["ratio_max"] = max(stats["ratio_max"], float(np.nanmax(ratio)))
                stats["bound_viol"] += int((ratio > 1.0).sum())
            with np.errstate(all="ignore"):
                us, vs, zs = u[sl], v[sl], z[sl]
                fu, fv = np.abs(us - np.rint(us)), np.abs(vs - np.rint(vs))
                tie = ~((fu > G_PX) & (fu < 0.5 - G_PX) & (fv > G_PX) & (fv < 0.5 - G_PX))
                nearz = ~(np.abs(zs) > zmin)
                cand = (homog_in[sl] & (zs > zmin)) | nearz
                xi = np.clip(np.rint(np.nan_to_num(us, nan=0, posinf=1e9, neginf=-1e9)), 0, W - 1).astype(np.int64)
                yi = np.clip(np.rint(np.nan_to_num(vs, nan=0, posinf=1e9, neginf=-1e9)), 0, H - 1).astype(np.int64)
                dtie = ~(np.abs(zs - dv_of[yi, xi]) > gz)
            guard[sl] = valid[sl] & cand & (tie | nearz | dtie)
    with np.errstate(all="ignore"):
        inview = valid & (u >= 0) & (u < W) & (v >= 0) & (v < H) & (z > 0)
        xi = np.clip(np.rint(np.nan_to_num(u, nan=0, posinf=1e9, neginf=-1e9)), 0, W - 1).astype(np.int64)
        yi = np.clip(np.rint(np.nan_to_num(v, nan=0, posinf=1e9, neginf=-1e9)), 0, H - 1).astype(np.int64)
        vis = inview & (z < dv_of[yi, xi])
        rin = ref["valid"].reshape(H, W) & O.check_point_in_image_boundary(ref["uv2"], hw).reshape(H, W) & (ref["depth2"].reshape(H, W) > 0)
    rxi, ryi, rvis = ref["xi"].reshape(H, W), ref["yi"].reshape(H, W), ref["vis"].reshape(H, W)
    mism = valid & ~guard & ((inview != rin) | (vis != rvis) | (rin & ((xi != rxi) | (yi != ryi))))
    key = "mismatch_const" if constant_guard else "mismatch"
    stats[key] += int(mism.sum())
    if not constant_guard:
        stats["guarded"] += int(guard.sum())
        stats["valid"] += int(valid.sum())
        with np.errstate(all="ignore"):
            stats["du_max"] = max(stats["du_max"], float(np.nanmax(np.where(valid & rin & ~guard, np.abs(u - ref["uv2"][:, 0].reshape(H, W)), 0.0))))
    return int(mism.sum())


def new_stats():
    return dict(zmin_max=0.0, zmin_sum=0.0, tiles=0, ratio_max=0.0, bound_viol=0, mismatch=0, mismatch_const=0, guarded=0,
                valid=0, du_max=0.0, pairs=0)


def run_case(name, K, A, E, depth, hw, pair_list, out):
    mats = engine.frame_matrices(K, A, E)
    st = new_stats()
    for ia, ib in pair_list:
        for const in (False, True):
            emulate_pair(depth[ia], depth[ib], K, E[ia], E[ib], A, hw, mats[ia], mats[ib], st, constant_guard=const)
        st["pairs"] += 1
    out.append((name, st))
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    out = []
    rng = np.random.default_rng(2024)
    hw_s, hw_b = (96, 128), (480, 640)

    def adv_case(hw, n):
        K, A, E = ADV.adversarial_pairs(rng, n, hw)
        boxes = synth._make_boxes(rng)
        depth = [ADV.render_mm(A @ e, K, hw, boxes, rng) for e in E]
        return K, A, E, depth, [(i, j) for i in range(n) for j in range(n)]

    K, A, E, depth, pairs = adv_case(hw_s, 12)
    run_case("adversarial poses, 96x128, all ordered pairs", K, A, E, depth, hw_s, pairs, out)
    for t in (1e4, 1e6):
        A2, E2 = ADV.translated(A, E, [t, -t, t])
        run_case(f"the same scene translated by {t:g} m (A' = A inv(T), E' = T E)", K, A2, E2, depth, hw_s,
                 pairs[:: (5 if (a.quick or t > 1e4) else 1)], out)
    for delta in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 3e-8, 1e-9):
        K, A, E, depth, pairs = ADV.near_plane_case(rng, hw_s, [delta], 10)
        run_case(f"camera 2 centred {delta:g} m behind a frame-1 point, 96x128", K, A, E, depth, hw_s, pairs, out)
    if not a.quick:
        for delta in (1e-4, 1e-6, 1e-7, 1e-9):
            K, A, E, depth, pairs = ADV.near_plane_case(rng, hw_b, [delta], 5)
            run_case(f"camera 2 centred {delta:g} m behind a frame-1 point, 640x480", K, A, E, depth, hw_b, pairs, out)
        K, A, E, depth, pairs = adv_case(hw_b, 6)
        run_case("adversarial poses, 640x480, all ordered pairs", K, A, E, depth, hw_b, pairs, out)
    print("# The fast kernels' guard as a bound, emulated on the CPU against the NumPy oracle (tools/guard_bound_emulation.py)\n")
    print(f"C = {C_BOUND:g}; per tile B_k = C 2^-53 (nr_k wmax + nt_k), zmin = 2 (B_0 + B_1 + (max(W, H) + 1) B_2) / 1e-6, "
          "gz = 2 B_2 + 1e-6 mm (mspa_common.h guard_from_bounds).  `err / B`: largest |q_fast - q_ref| over its bound; "
          "`du`: largest |u_fast - u_ref| among lanes the guard does NOT send to the reference chain.\n")
    print("| case | pairs | valid lanes | worst err / B | lanes over B | zmin mean / max (mm) | worst unguarded du (px) | guarded lanes | "
          "mismatches, bound guard | mismatches, round-3 constant guard |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    bad = 0
    for name, st in out:
        print(f"| {name} | {st['pairs']} | {st['valid']} | {st['ratio_max']:.3g} | {st['bound_viol']} | "
              f"{st['zmin_sum'] / max(st['tiles'], 1):.3g} / {st['zmin_max']:.3g} | {st['du_max']:.3g} | "
              f"{st['guarded']} ({100.0 * st['guarded'] / max(st['valid'], 1):.4f} %) | {st['mismatch']} | {st['mismatch_const']} |")
        bad += st["mismatch"] + st["bound_viol"]
    if bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
