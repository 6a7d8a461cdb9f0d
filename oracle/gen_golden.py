"""TEST INFRASTRUCTURE (build container only): freeze reference outputs into tests/golden/*.npz.

Runs the *imported, unmodified* reference (oracle/ref_harness.py) on small seeded synthetic inputs
and stores inputs + the reference's outputs side by side, so that the oracle and the HIP path can
be checked against the reference on machines where /root/reference does not exist (the GPU box).
The files hold data only -- arrays the reference consumed and produced -- never reference source.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Recorded versions (numpy / OpenBLAS kernel) are stored in each file under ``meta``.
"""
from __future__ import annotations

import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]

from mspa import synth  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def _meta():
    import platform
    return json.dumps({"numpy": np.__version__, "python": platform.python_version(),
                       "machine": platform.machine(),
                       "reference": "facebookresearch/Multi-SpatialMLLM @ /root/reference (2025-10-31)",
                       "generator": "oracle/gen_golden.py"})


def scene_arrays(sc):
    ids = sc.image_ids
    out = {
        "K": sc.K, "A": sc.A, "points": sc.points,
        "image_ids": np.array(ids), "E": np.stack([sc.E[i] for i in ids]),
        "depth": np.stack([sc.depth[i] for i in ids]),
        "color_hw": np.array(sc.color_hw), "depth_hw": np.array(sc.depth_hw),
    }
    if sc.color:
        out["color"] = np.stack([sc.color[i] for i in ids])
    return out


def golden_scene(ns, name, seed, color_hw, depth_hw, n_points, n_frames, with_color):
    sc = synth.make_scene(seed, n_points=n_points, n_frames=n_frames, color_hw=color_hw, depth_hw=depth_hw,
                          invalid_pose_frac=0.2, with_color=with_color)
    h = RH.make_handler(ns, [sc])
    sid = sc.scene_id
    g = scene_arrays(sc)
    valid_ids = h.get_all_extrinsic_valid_image_ids(sid)
    g["valid_image_ids"] = np.array(valid_ids)
    pts = h.get_scene_points_align(sid)[:, :3]
    uv, dep, vis = [], [], []
    for image_id in valid_ids:                                   # HOT LOOP 1 through the reference
        u, d = h.project_3d_point_to_image(sid, image_id, pts)
        uv.append(u)
        dep.append(d)
        vis.append(h.check_point_visibility(sid, image_id, u, d))
    g["ref_uv"], g["ref_depth"], g["ref_vis"] = np.stack(uv), np.stack(dep), np.stack(vis)

    # a7: masked back-projection of the second valid frame
    rng = np.random.default_rng(seed + 7)
    H, W = color_hw
    mask = rng.random((H, W)) < 0.3
    fid = valid_ids[1]
    color = sc.color.get(fid)
    g["a7_mask"] = mask
    g["a7_frame"] = np.array(fid)
    g["ref_a7"] = ns.OPS.project_mask_to_3d(sc.depth[fid], sc.K, sc.E[fid], mask, sc.A, color)
    g["ref_a7_noalign"] = ns.OPS.project_mask_to_3d(sc.depth[fid], sc.K, sc.E[fid], mask)

    # composite pairs a7 -> a2 -> a5 (all pixels), through the reference's own entry points
    pair_list = [(valid_ids[0], valid_ids[2]), (valid_ids[2], valid_ids[0]), (valid_ids[1], valid_ids[1])]
    g["pair_ids"] = np.array(pair_list)
    for n, (id1, id2) in enumerate(pair_list):
        col = sc.color.get(id1)
        if col is None:
            col = np.zeros((H, W, 3), np.uint8)
        full = np.ones((H, W), dtype=bool)
        p3, = (ns.OPS.project_mask_to_3d(sc.depth[id1], sc.K, sc.E[id1], full, sc.A, col),)
        u, d = h.project_3d_point_to_image(sid, id2, p3[:, :3])
        v = h.check_point_visibility(sid, id2, u, d)
        g[f"pair{n}_xyzrgb"], g[f"pair{n}_uv"], g[f"pair{n}_depth"], g[f"pair{n}_vis"] = p3, u, d, v

    # CFR / MVI scene products
    _, table = ns.CFR.process_scene(sid, h, os.path.join("/tmp", f"mspa_golden_warn_{name}.txt"))
    keys = list(table.keys())
    g["cfr_pairs"] = np.array(keys)
    g["cfr_values"] = np.array([[table[k][f] for f in ("overlap", "distance", "yaw", "pitch")] for k in keys])
    _, vis_info = ns.MVI.process_scene(sid, h, os.path.join("/tmp", f"mspa_golden_warn2_{name}.txt"))
    g["mvi_json"] = np.array(json.dumps({"image_to_points": vis_info["image_to_points"],
                                         "point_to_images": {str(k): v for k, v in
                                                             vis_info["point_to_images"].items()}}))

    # CME relative pose: both swap branches and the yaw wrap
    rows, answers, swaps = [], [], []
    for n, (k, vals) in enumerate(zip(keys, g["cfr_values"])):
        row = {"scene_id": sid, "image_id1": k[0], "image_id2": k[1], "overlap": 20.0,
               "distance": float(vals[1]), "yaw": float(vals[2]) + (290.0 if n % 3 == 0 else 0.0) -
               (310.0 if n % 5 == 0 else 0.0), "pitch": float(vals[3])}
        random.seed(1000 + n)
        swap = random.random() < 0.5
        random.seed(1000 + n)
        sample = ns.CME.build_training_sample(h, row, n, "total_distance")
        av = sample["answer_values"]
        rows.append([row["yaw"], row["pitch"]])
        swaps.append(swap)
        answers.append(json.dumps(av))
    g["cme_yaw_pitch"] = np.array(rows)
    g["cme_swap"] = np.array(swaps)
    g["cme_answers_json"] = np.array(answers)
    g["meta"] = np.array(_meta())
    path = os.path.join(GOLDEN_DIR, f"{name}.npz")
    np.savez_compressed(path, **g)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


def golden_ties(ns):
    """Engineered rounding ties and depth equalities, every operation exact in float64."""
    H, W = 48, 64
    K = np.eye(4)
    K[0, 0] = K[1, 1] = 64.0
    K[0, 2], K[1, 2] = 32.0, 24.0
    E = np.eye(4)
    E[:3, 3] = [0.5, -0.25, 0.125]
    A = np.eye(4)
    A[:3, 3] = [1.0, 2.0, -0.5]
    Ea = A @ E
    depth = np.full((H, W), 2000, dtype=np.uint16)      # 2000 * 0.001 == 2.0 exactly in float64
    depth[10, 10] = 0
    cam_pts = []
    for kx in (-1, 0, 5, 6, 30, 62, 63, 64):            # u = kx + 0.5 exactly (half-integers), and integers
        for ky in (-1, 0, 7, 8, 46, 47, 48):
            for half in (0.0, 0.5):
                for z in (2.0, np.nextafter(2.0, 0.0), np.nextafter(2.0, 3.0), 1.0, -2.0, 0.0):
                    u, v = kx + half, ky + half
                    x = (u - 32.0) * z / 64.0
                    y = (v - 24.0) * z / 64.0
                    cam_pts.append([x, y, z])
    cam_pts = np.array(cam_pts)
    world = (Ea @ np.hstack([cam_pts, np.ones((len(cam_pts), 1))]).T).T[:, :3]   # exact: dyadic values
    points = np.hstack([world, np.zeros((len(world), 3))])
    sc = synth.SynthScene("ties0000_00", K, A, {"00000": E}, points, {"00000": depth},
                          {"00000": np.zeros((H, W, 3), np.uint8)}, (H, W), (H, W), np.zeros((0, 2, 3)))
    h = RH.make_handler(ns, [sc])
    with np.errstate(all="ignore"):
        uv, d = h.project_3d_point_to_image(sc.scene_id, "00000", world)
        vis = h.check_point_visibility(sc.scene_id, "00000", uv, d)
        inb = h.check_point_in_image_boundary(sc.scene_id, uv)
    g = scene_arrays(sc)
    g.update(ref_uv=uv, ref_depth=d, ref_vis=vis, ref_inb=inb, cam_pts=cam_pts, meta=np.array(_meta()))
    path = os.path.join(GOLDEN_DIR, "ties.npz")
    np.savez_compressed(path, **g)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB; visible {int(vis.sum())}/{len(vis)}")


def golden_tracks(ns):
    tr = synth.make_tracks(11, T=24, P=48)
    eng = ns.OM_C.TwoFrameVideoQAEngine("tapvid3d_displacement_vector", "adt")
    c2w = np.linalg.inv(tr.extrinsics_w2c)                                 # OM_C:446-454 verbatim steps
    hom = np.concatenate([tr.tracks_XYZ, np.ones(tr.tracks_XYZ.shape[:2] + (1,))], axis=2)
    world = np.einsum("nij,nkj->nki", c2w, hom)[..., :3]
    rng = np.random.default_rng(5)
    pairs = [(int(a), int(b), int(p)) for a, b, p in
             zip(rng.integers(0, 24, 64), rng.integers(0, 24, 64), rng.integers(0, 48, 64))]
    H, W = tr.image_hw
    recs_json, kept = [], []
    for (f1, f2, p) in pairs:
        random.seed(0)
        r = eng.format_training_samples([{"frame1": f1, "frame2": f2, "point_index": p}], tr.fx_fy_cx_cy,
                                        tr.scene_id, world, tr.tracks_XYZ, H, W, tr.extrinsics_w2c)
        kept.append(len(r) == 1)
        if r:
            r = r[0]
            recs_json.append(json.dumps({k: r[k] for k in ("gt_value", "point_moving", "cam_moving", "p1", "p2")}))
        else:
            recs_json.append("null")
    groups = ns.OM_C.rigid_body_segmentation(tr.tracks_XYZ)
    g = {"tracks_XYZ": tr.tracks_XYZ, "visibility": tr.visibility, "extrinsics_w2c": tr.extrinsics_w2c,
         "fx_fy_cx_cy": tr.fx_fy_cx_cy, "image_hw": np.array(tr.image_hw), "ref_world": world,
         "pairs": np.array(pairs), "kept": np.array(kept), "records_json": np.array(recs_json),
         "groups_json": np.array(json.dumps(groups)), "meta": np.array(_meta())}
    path = os.path.join(GOLDEN_DIR, "tracks.npz")
    np.savez_compressed(path, **g)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB; kept {sum(kept)}/{len(kept)}")


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    ns = RH.import_reference()
    golden_scene(ns, "scene_ident", 2001, (48, 64), (48, 64), n_points=700, n_frames=6, with_color=True)
    golden_scene(ns, "scene_scaled", 2002, (73, 98), (48, 64), n_points=700, n_frames=6, with_color=False)
    golden_ties(ns)
    golden_tracks(ns)


if __name__ == "__main__":
    main()
