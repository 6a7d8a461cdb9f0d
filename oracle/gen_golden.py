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


def stable_color_image(hw):
    """A colour image that needs no random generator (so that tests can rebuild it anywhere, bit for bit)."""
    H, W = hw
    return ((np.arange(H * W * 3, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)).astype(np.uint8).reshape(H, W, 3)


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden_scannet_shape(ns):
    """The reference at ScanNet's own shapes (colour 1296x968 over depth 640x480): vertex projections and
    masks as arrays, the full-frame back-projection and one composite pair as SHA-256 of the float64 bytes."""
    color_hw, depth_hw = (968, 1296), (480, 640)
    sc = synth.make_scene(6006, n_points=4096, n_frames=3, color_hw=color_hw, depth_hw=depth_hw, invalid_pose_frac=0.34,
                          with_color=False)
    ids = sc.image_ids
    bad = [i for i in ids if not np.isfinite(sc.E[i]).all()]
    assert len(bad) == 1
    sc.depth[bad[0]] = np.zeros(depth_hw, dtype=np.uint16)          # never read (the frame is dropped); compresses to nothing
    h = RH.make_handler(ns, [sc])
    sid = sc.scene_id
    g = scene_arrays(sc)
    valid = h.get_all_extrinsic_valid_image_ids(sid)
    g["valid_image_ids"] = np.array(valid)
    pts = h.get_scene_points_align(sid)[:, :3]
    uv, dep, vis = [], [], []
    for image_id in valid:
        u, d = h.project_3d_point_to_image(sid, image_id, pts)
        uv.append(u)
        dep.append(d)
        vis.append(h.check_point_visibility(sid, image_id, u, d))
    g["ref_uv"], g["ref_depth"], g["ref_vis"] = np.stack(uv), np.stack(dep), np.stack(vis)
    color = stable_color_image(color_hw)
    f0, f1 = valid[0], valid[1]
    a7 = ns.OPS.project_mask_to_3d(sc.depth[f0], sc.K, sc.E[f0], None, sc.A, color)       # colour grid: 1,254,528 pixels
    g["a7_rows"] = np.array(a7.shape[0])
    g["a7_sha_xyz"], g["a7_sha_rgb"], g["a7_sha_all"] = np.array(sha(a7[:, :3])), np.array(sha(a7[:, 3:])), np.array(sha(a7))
    g["a7_head"] = a7[:64].copy()
    u, d = h.project_3d_point_to_image(sid, f1, a7[:, :3])
    v = h.check_point_visibility(sid, f1, u, d)
    g["pair_ids"] = np.array([[f0, f1]])
    g["pair_n_vis"] = np.array(int(v.sum()))
    g["pair_sha_vis"], g["pair_sha_uv"], g["pair_sha_depth"] = np.array(sha(v)), np.array(sha(u)), np.array(sha(d))
    _, table = ns.CFR.process_scene(sid, h, "/tmp/mspa_golden_warn_scannet_shape.txt")
    keys = list(table.keys())
    g["cfr_pairs"] = np.array(keys)
    g["cfr_values"] = np.array([[table[k][f] for f in ("overlap", "distance", "yaw", "pitch")] for k in keys])
    g["meta"] = np.array(_meta())
    path = os.path.join(GOLDEN_DIR, "scannet_shape.npz")
    np.savez_compressed(path, **g)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB; a7 rows {a7.shape[0]}, pair visible {int(v.sum())}")


def golden_k3_640x480(ns):
    """The reference at the BASELINE shape itself (colour = depth = 640x480, BASELINE.json configs[1]): two frames of a seeded
    scene, both orders.  Per pair: project_mask_to_3d over the full frame (OPS:235-329), project_3d_point_to_image into the
    other frame (IH:313-335), check_point_visibility (IH:375-386).  Frozen: SHA-256 of the float64 bytes, the visibility mask as
    a bitset in frame-1 pixel order, the counters, and SHA-256 of the [P, 2] int16 pixel-index table in the layout K3 writes it
    ((xi, yi) = IH:362-366's clipped half-to-even indices where the point lies inside frame 2 in front of the camera,
    (-1, -1) elsewhere)."""
    hw = (480, 640)
    sc = synth.make_scene(6007, n_points=64, n_frames=2, color_hw=hw, depth_hw=hw, invalid_pose_frac=0.0, with_color=False,
                          walk_step=0.1, target_jitter=0.3)
    h = RH.make_handler(ns, [sc])
    sid = sc.scene_id
    g = scene_arrays(sc)
    ids = h.get_all_extrinsic_valid_image_ids(sid)
    assert len(ids) == 2
    color = stable_color_image(hw)
    P = hw[0] * hw[1]
    g["pair_ids"] = np.array([[ids[0], ids[1]], [ids[1], ids[0]]])
    for n, (f0, f1) in enumerate(g["pair_ids"]):
        f0, f1 = str(f0), str(f1)
        a7 = ns.OPS.project_mask_to_3d(sc.depth[f0], sc.K, sc.E[f0], None, sc.A, color)
        valid = (sc.depth[f0].reshape(-1) > 0)
        assert a7.shape[0] == int(valid.sum())
        u, d = h.project_3d_point_to_image(sid, f1, a7[:, :3])
        v = h.check_point_visibility(sid, f1, u, d)
        inb = h.check_point_in_image_boundary(sid, u)
        with np.errstate(invalid="ignore"):
            inview = inb & (d > 0)
            xi = np.clip(np.round(u[:, 0]).astype(int), 0, hw[1] - 1)          # IH:362-366 (sx = sy = 1)
            yi = np.clip(np.round(u[:, 1]).astype(int), 0, hw[0] - 1)
        pix = np.full((P, 2), -1, dtype=np.int16)
        rows = np.nonzero(valid)[0]
        pix[rows[inview], 0] = xi[inview]
        pix[rows[inview], 1] = yi[inview]
        vis_full = np.zeros(P, dtype=bool)
        vis_full[rows] = v
        g[f"pair{n}_rows"] = np.array(a7.shape[0])
        g[f"pair{n}_n_vis"] = np.array(int(v.sum()))
        g[f"pair{n}_n_inview"] = np.array(int(inview.sum()))
        g[f"pair{n}_sha_xyz"], g[f"pair{n}_sha_rgb"] = np.array(sha(a7[:, :3])), np.array(sha(a7[:, 3:]))
        g[f"pair{n}_sha_uv"], g[f"pair{n}_sha_depth"] = np.array(sha(u)), np.array(sha(d))
        g[f"pair{n}_vis_bits"] = np.packbits(vis_full, bitorder="little")
        g[f"pair{n}_sha_pix"] = np.array(sha(pix))
        g[f"pair{n}_xyz_head"] = a7[:64, :3].copy()
        print(f"k3_640x480 pair {n}: rows {a7.shape[0]}, in view {int(inview.sum())}, visible {int(v.sum())}")
    g["meta"] = np.array(_meta())
    path = os.path.join(GOLDEN_DIR, "k3_640x480.npz")
    np.savez_compressed(path, **g)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


def golden_k3_near_plane(ns):
    """The reference in the regime round 4's guard band is about: camera 2 centred 1e-4 / 1e-7 / 1e-9 m
    behind a back-projected frame-1 point, the point engineered onto rounding ties and image bounds +- 2e-6 px
    (tests/adversarial.py near_plane_case), and adversarial poses with the world shifted by 1e4 m (E' = T E, A' = A inv(T)).
    96x128, colour = depth.  Per pair, through the reference's own functions (OPS:235-329 -> IH:46-72, then IH:337-371's
    expressions): the visibility mask as a bitset, counters, SHA-256 of the float64 projections."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import adversarial as ADV
    hw = (96, 128)
    H, W = hw
    rng = np.random.default_rng(4242)
    cases = []
    K, A, E, depth, pairs = ADV.near_plane_case(rng, hw, [1e-4, 1e-7, 1e-9], 3)
    cases.append(("near", K, A, E, depth, pairs))
    K2, A2, E2 = ADV.adversarial_pairs(rng, 4, hw)
    boxes = synth._make_boxes(rng)
    depth2 = [ADV.render_mm(A2 @ e, K2, hw, boxes, rng) for e in E2]
    A3, E3 = ADV.translated(A2, E2, [1e4, -1e4, 1e4])
    cases.append(("far", K2, A3, E3, depth2, [(0, 1), (1, 0), (2, 3), (3, 1)]))
    g = {"hw": np.array(hw), "cases": np.array([c[0] for c in cases])}
    for name, K, A, E, depth, pairs in cases:
        g[f"{name}_K"], g[f"{name}_A"], g[f"{name}_E"] = K, A, np.stack(E)
        g[f"{name}_depth"], g[f"{name}_pairs"] = np.stack(depth), np.array(pairs)
        for n, (ia, ib) in enumerate(pairs):
            mask = np.ones(hw, dtype=bool)
            a7 = ns.OPS.project_mask_to_3d(depth[ia], K, E[ia], mask, A, None)
            uv, d = ns.IH.project_points(np.hstack([a7[:, :3], np.ones((a7.shape[0], 1))]), K, A @ E[ib])
            with np.errstate(invalid="ignore"):
                inb = (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)             # IH:337-344
                xi = np.clip(np.round(uv[:, 0]).astype(int), 0, W - 1)                                     # IH:362-366
                yi = np.clip(np.round(uv[:, 1]).astype(int), 0, H - 1)
                vis = inb & (d > 0) & (d < depth[ib][yi, xi] * 0.001)                                      # IH:368-371
                inview = inb & (d > 0)
            valid = depth[ia].reshape(-1) > 0
            rows = np.nonzero(valid)[0]
            full = np.zeros(H * W, dtype=bool)
            full[rows] = vis
            pix = np.full((H * W, 2), -1, dtype=np.int16)
            pix[rows[inview], 0], pix[rows[inview], 1] = xi[inview], yi[inview]
            g[f"{name}{n}_rows"], g[f"{name}{n}_n_vis"] = np.array(a7.shape[0]), np.array(int(vis.sum()))
            g[f"{name}{n}_vis_bits"] = np.packbits(full, bitorder="little")
            g[f"{name}{n}_sha_uv"], g[f"{name}{n}_sha_depth"] = np.array(sha(uv)), np.array(sha(d))
            g[f"{name}{n}_sha_pix"] = np.array(sha(pix))
            g[f"{name}{n}_min_abs_depth2"] = np.array(float(np.nanmin(np.abs(d))))
    g["meta"] = np.array(_meta())
    path = os.path.join(GOLDEN_DIR, "k3_near_plane.npz")
    np.savez_compressed(path, **g)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


def golden_k1_near_vertex(ns):
    """K1's half of the same regime: aligned camera poses centred 1e-3 / 1e-6 / 1e-9 m behind scene vertices (the vertex on a
    rounding tie / an image bound +- 2e-6 px, tests/adversarial.py near_vertex_cameras), and the same cameras and vertices 1e5 m
    from the origin.  Per camera, the reference's project_points (IH:46-72) and IH:337-371's expressions: visibility mask as a
    bitset, SHA-256 of the float64 projections.  96x128, colour = depth."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import adversarial as ADV
    hw = (96, 128)
    H, W = hw
    rng = np.random.default_rng(4343)
    sc = synth.make_scene(778, n_points=4096 + 37, n_frames=2, color_hw=hw, depth_hw=hw, invalid_pose_frac=0.0, with_color=False)
    pts = np.ascontiguousarray(sc.points[:, :3])
    boxes = synth._make_boxes(rng)
    E_al = ADV.near_vertex_cameras(rng, pts, sc.K, hw, (1e-3, 1e-6, 1e-9), per_delta=3)
    depth = [ADV.render_mm(e, sc.K, hw, boxes, rng) for e in E_al]
    shift = np.array([1e5, -1e5, 1e5])
    g = {"hw": np.array(hw), "K": sc.K, "points": pts, "E": np.stack(E_al), "depth": np.stack(depth), "shift": shift}
    for tag, P3, Es in (("near", pts, E_al), ("far", pts + shift, [np.block([[e[:3, :3], (e[:3, 3] + shift)[:, None]], [e[3:, :]]]) for e in E_al])):
        for k, e in enumerate(Es):
            uv, d = ns.IH.project_points(np.hstack([P3, np.ones((P3.shape[0], 1))]), sc.K, e)
            with np.errstate(invalid="ignore"):
                inb = (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)
                xi = np.clip(np.round(uv[:, 0]).astype(int), 0, W - 1)
                yi = np.clip(np.round(uv[:, 1]).astype(int), 0, H - 1)
                vis = inb & (d > 0) & (d < depth[k][yi, xi] * 0.001)
            g[f"{tag}{k}_vis_bits"] = np.packbits(vis, bitorder="little")
            g[f"{tag}{k}_n_vis"] = np.array(int(vis.sum()))
            g[f"{tag}{k}_sha_uv"], g[f"{tag}{k}_sha_depth"] = np.array(sha(uv)), np.array(sha(d))
    g["meta"] = np.array(_meta())
    path = os.path.join(GOLDEN_DIR, "k1_near_vertex.npz")
    np.savez_compressed(path, **g)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


def golden_cme256(ns):
    """G6 of SURVEY.md 8c: CME.build_training_sample answer_values for 256 pairs of a 24-frame walk -- both swap branches,
    yaw differences pushed beyond +-180 (wrap), a static pair, mirrored pairs."""
    sc = synth.make_scene(8118, n_points=64, n_frames=24, color_hw=(48, 64), depth_hw=(48, 64), invalid_pose_frac=0.0,
                          with_color=False, walk_step=0.4, target_jitter=1.5)
    h = RH.make_handler(ns, [sc])
    sid, ids = sc.scene_id, sc.image_ids
    rng = np.random.default_rng(3)
    pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, 24, 256), rng.integers(0, 24, 256))]
    pairs[0] = (5, 5)                                                     # no motion at all
    pairs[1], pairs[2] = (3, 17), (17, 3)
    yaw_pitch = {}
    for i in ids:
        yaw_pitch[i] = ns.CFR.extract_yaw_pitch(h.get_extrinsic_matrix_align(sid, i))
    rows, answers, swaps = [], [], []
    for n, (a, b) in enumerate(pairs):
        ia, ib = ids[a], ids[b]
        ta = h.get_extrinsic_matrix_align(sid, ia)[:3, 3]
        tb = h.get_extrinsic_matrix_align(sid, ib)[:3, 3]
        yaw = yaw_pitch[ib][0] - yaw_pitch[ia][0] + (290.0 if n % 7 == 0 else 0.0) - (310.0 if n % 11 == 0 else 0.0)
        row = {"scene_id": sid, "image_id1": ia, "image_id2": ib, "overlap": 20.0, "distance": float(np.linalg.norm(tb - ta)),
               "yaw": float(yaw), "pitch": float(yaw_pitch[ib][1] - yaw_pitch[ia][1])}
        random.seed(4000 + n)
        swaps.append(random.random() < 0.5)
        random.seed(4000 + n)
        sample = ns.CME.build_training_sample(h, row, n, "displacement_vector")
        rows.append([a, b, row["yaw"], row["pitch"], row["distance"]])
        answers.append(json.dumps(sample["answer_values"]))
    out = {"A": sc.A, "E": np.stack([sc.E[i] for i in ids]), "rows": np.array(rows), "swap": np.array(swaps),
           "answers_json": np.array(answers), "meta": np.array(_meta())}
    np.savez_compressed(os.path.join(GOLDEN_DIR, "cme256.npz"), **out)
    print("cme256.npz: swaps", int(np.sum(swaps)), "of", len(swaps), "wrapped rows", int(np.sum(np.abs(np.array(rows)[:, 2]) > 180)))


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


def golden_coverage(ns, only=None):
    """Object coverage search (COV) + object visibility (COVIS) + perception records (OPE) of the reference on a
    70-frame scene whose eight furniture boxes are the objects; run once on float64 and once on float32 points."""
    import pandas  # noqa: F401  (COVIS/COV import it)
    RH.import_object_perception(ns)
    sc = synth.make_scene(4242, n_points=6000, n_frames=70, color_hw=(48, 64), depth_hw=(48, 64),
                          invalid_pose_frac=0.05, with_color=False, walk_step=0.6, target_jitter=1.6)
    xyz = sc.points[:, :3]
    inst = np.zeros(len(xyz), dtype=np.int64)
    bboxes, cats = [], []
    for o, (lo, hi) in enumerate(sc.boxes):
        on = ((xyz >= lo - 1e-9) & (xyz <= hi + 1e-9)).all(axis=1) & (inst == 0)
        inst[on] = o + 1
        p = xyz[on]
        c, d = (p.max(0) + p.min(0)) / 2, p.max(0) - p.min(0)
        bboxes.append(np.concatenate([c, d, [o]]))
        cats.append("wall" if o == 6 else f"cabinet{o}")
    out = {"points": sc.points, "instance_mask": inst, "bboxes": np.stack(bboxes), "categories": np.array(cats),
           "image_ids": np.array(sc.image_ids), "color_hw": np.array(sc.color_hw)}
    for tag, dtype in (("f64", np.float64), ("f32", np.float32)):
        sc_t = synth.SynthScene(sc.scene_id, sc.K, sc.A, sc.E, sc.points.astype(dtype), sc.depth, sc.color,
                                sc.color_hw, sc.depth_hw, sc.boxes)
        h = RH.make_handler(ns, [sc_t])
        sid = sc.scene_id
        np.save(os.path.join(h.instance_data_root, sid, "instance_mask.npy"), inst)
        h.infos[sid]["num_objects"] = len(bboxes)
        for o in range(len(bboxes)):
            h.infos[sid][o] = {"raw_category": cats[o], "aligned_bbox": bboxes[o].astype(dtype), "unaligned_bbox": bboxes[o]}
        _, vis = ns.MVI.process_scene(sid, h, os.path.join("/tmp", f"mspa_golden_warn_cov_{tag}.txt"))
        vis_dict = {f"{sid}:image_to_points:{k}": json.dumps(v) for k, v in vis["image_to_points"].items()}
        _, objvis, _ = ns.COVIS.process_scene(sid, h, vis_dict)
        random.seed(0)
        _, cov = ns.COV.process_scene_for_coverage(sid, h, vis_dict, {sid: objvis})
        if tag == "f64":
            valid = h.get_all_extrinsic_valid_image_ids(sid)
            bits = np.zeros((len(valid), (len(xyz) + 63) // 64 * 64), dtype=bool)
            for r, img in enumerate(valid):
                bits[r, vis["image_to_points"][img]] = True
            out["valid_image_ids"] = np.array(valid)
            out["vis_bits"] = np.packbits(bits, axis=1, bitorder="little").view(np.int64)
            out["object_visibility_json"] = json.dumps(objvis)
        out[f"coverage_{tag}_json"] = json.dumps({str(o): {dim: {str(k): [list(c) for c in combos] for k, combos in t.items()}
                                                           for dim, t in res.items()} for o, res in cov.items()})
        if tag == "f64":
            # perception records of the reference from this table (its script needs two repairs to run at all:
            # TASK_DESCRIPTION is the list it defines as ASK_DESCRIPTION, and the handler gets image_height/width)
            import pickle
            import tempfile
            ns.OPE.TASK_DESCRIPTION = ns.OPE.ASK_DESCRIPTION
            h.image_height, h.image_width = sc.color_hw
            tmp = tempfile.mkdtemp(prefix="mspa_ope_")
            table = {sid: {o: res["height"] for o, res in cov.items()}}
            with open(os.path.join(tmp, "height.pkl"), "wb") as f:
                pickle.dump(table, f)
            random.seed(1)
            ns.OPE.build_lwh_qa_samples(h, os.path.join(tmp, "height.pkl"), "height", "val", tmp, max_k=6, max_samples=40)
            recs = {}
            for fname in sorted(os.listdir(tmp)):
                if fname.endswith(".jsonl"):
                    recs[fname] = [json.loads(line) for line in open(os.path.join(tmp, fname))]
            out["ope_records_json"] = json.dumps(recs)
    # direct cases for the search itself: images that each see an interval of a 1-D object, so that several
    # are needed, both random caps (25 first-layer images, 5000 nodes per level) trigger and levels reach 5
    cases = []
    rng = np.random.default_rng(77)
    for c, (n_img, frac, dtype) in enumerate([(12, 0.6, np.float64), (34, 0.45, np.float64), (40, 0.3, np.float32),
                                              (30, 0.22, np.float64), (26, 0.5, np.float32), (3, 0.2, np.float64)]):
        n_pts = 300
        pts = np.zeros((n_pts, 3), dtype=dtype)
        axis = c % 3
        pts[:, axis] = rng.uniform(-1.0, 2.0, n_pts).astype(dtype)
        obj = np.sort(rng.choice(n_pts, size=220, replace=False))
        coords = pts[obj, axis]
        target = float(coords.max() - coords.min()) * (1.0 if c % 2 == 0 else 0.97)
        lists = {}
        imgs = [f"{5 * k:05d}" for k in range(n_img)]
        for k, img in enumerate(imgs):
            a = rng.uniform(coords.min() - 0.2, coords.max() - frac * 3.0 + 0.2)
            seen = (pts[:, axis] >= a) & (pts[:, axis] <= a + frac * 3.0) & (rng.random(n_pts) < 0.9)
            if k == 1:
                seen[:] = False                                           # an image that sees nothing
            lists[f"s:image_to_points:{img}"] = json.dumps(np.where(seen)[0].tolist())
        visible = list(imgs)
        if c == 1:
            del lists[f"s:image_to_points:{imgs[4]}"]                      # missing from the index (COV:106-108)
        random.seed(100 + c)
        want = ns.COV.find_minimal_combinations("s", pts, obj, visible, lists, axis, target, 0.1)
        cases.append({"points": pts.tolist(), "dtype": np.dtype(dtype).name, "object": obj.tolist(), "axis": axis,
                      "target": target, "images": visible, "lists": lists, "seed": 100 + c,
                      "want": {str(k): [list(x) for x in v] for k, v in want.items()}})
        print("bfs case", c, {k: len(v) for k, v in want.items()})
    out["bfs_cases_json"] = json.dumps(cases)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "coverage.npz"), meta=_meta(), **out)
    cov64 = json.loads(out["coverage_f64_json"])
    print("images per object:", {o: len(v) for o, v in json.loads(out["object_visibility_json"])["object_to_images"].items()})
    print("coverage.npz: objects", list(cov64), "solutions per k (height):",
          {o: {k: len(v) for k, v in r["height"].items()} for o, r in cov64.items()},
          "f32 == f64:", out["coverage_f32_json"] == out["coverage_f64_json"])


def golden_sens(ns):
    """A synthetic .sens stream parsed by the reference's SensorData: poses, intrinsics, depth frames, the text it
    exports and what update_info_file_with_images.py parses back from that text."""
    import tempfile
    import warnings
    from mspa import sens as S
    RH.import_sens(ns)
    sc = synth.make_scene(5151, n_points=64, n_frames=11, color_hw=(48, 64), depth_hw=(24, 32), invalid_pose_frac=0.2,
                          with_color=False)
    ids = sc.image_ids
    poses = [sc.E[i].astype(np.float32) * np.float32(1.0000001) for i in ids]      # not representable in 6 decimals
    K = sc.K.astype(np.float32)
    tmp = tempfile.mkdtemp(prefix="mspa_sens_")
    path = os.path.join(tmp, "scene5151_00.sens")
    payloads = [bytes([k, 255 - k, 7]) * (5 + k) for k in range(len(ids))]
    S.write_sens(path, K, poses, [sc.depth[i] for i in ids], color_hw=sc.color_hw, color_payloads=payloads)
    out = {"sens_bytes": np.frombuffer(open(path, "rb").read(), dtype=np.uint8)}
    for skip in (1, 2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            data = ns.SENS.SensorData(path, skip)
            folder = os.path.join(tmp, f"posed_{skip}")
            data.export_intrinsics(folder)
            data.export_poses(folder)
            RH.STORE.written.clear()
            data.export_depth_images(folder)
        out[f"skip{skip}_c2w"] = np.stack([f.camera_to_world for f in data.frames])
        out[f"skip{skip}_depth"] = np.stack([RH.STORE.written[os.path.join(folder, data.index_to_str(k) + ".png")]
                                             for k in range(len(data.frames))])
        out[f"skip{skip}_color"] = np.array([f.color_data for f in data.frames], dtype=object).astype("S")
        texts = {name: open(os.path.join(folder, name)).read() for name in sorted(os.listdir(folder))}
        out[f"skip{skip}_text_json"] = json.dumps(texts)
        # update_info_file_with_images.py:30-35,48-53 -- the parse of those files, every 5th exported frame
        parse = lambda t: np.array([list(map(float, line.split())) for line in t.splitlines()])   # noqa: E731
        out[f"skip{skip}_info_K"] = parse(texts["intrinsic.txt"])
        kept = [k for k in range(len(data.frames)) if k % 5 == 0]
        out[f"skip{skip}_info_E"] = np.stack([parse(texts[data.index_to_str(k) + ".txt"]) for k in kept])
        out[f"skip{skip}_header"] = np.array([data.color_width, data.color_height, data.depth_width, data.depth_height,
                                              len(data.frames)])
        assert data.depth_compression_type == "zlib_ushort" and data.color_compression_type == "jpeg"
    np.savez_compressed(os.path.join(GOLDEN_DIR, "sens.npz"), meta=_meta(), **out)
    print("sens.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    ns = RH.import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "coverage":
        return golden_coverage(ns)
    if len(sys.argv) > 1 and sys.argv[1] == "sens":
        return golden_sens(ns)
    if len(sys.argv) > 1 and sys.argv[1] == "scannet_shape":
        return golden_scannet_shape(ns)
    if len(sys.argv) > 1 and sys.argv[1] == "cme256":
        return golden_cme256(ns)
    if len(sys.argv) > 1 and sys.argv[1] == "k3_640x480":
        return golden_k3_640x480(ns)
    if len(sys.argv) > 1 and sys.argv[1] == "k3_near_plane":
        return golden_k3_near_plane(ns)
    if len(sys.argv) > 1 and sys.argv[1] == "k1_near_vertex":
        return golden_k1_near_vertex(ns)
    golden_scene(ns, "scene_ident", 2001, (48, 64), (48, 64), n_points=700, n_frames=6, with_color=True)
    golden_scene(ns, "scene_scaled", 2002, (73, 98), (48, 64), n_points=700, n_frames=6, with_color=False)
    golden_ties(ns)
    golden_tracks(ns)
    golden_coverage(ns)
    golden_sens(ns)
    golden_scannet_shape(ns)
    golden_cme256(ns)
    golden_k3_640x480(ns)
    golden_k3_near_plane(ns)
    golden_k1_near_vertex(ns)


if __name__ == "__main__":
    main()
