"""TEST INFRASTRUCTURE -- NumPy CPU restatement of the reference's per-frame(-pair) geometry path.

This file is the parity oracle for the HIP kernels.  It restates, function by function and in
the reference's own operation order (float64 everywhere, BLAS mat-mat products, ``np.round``
half-to-even, strict ``<`` depth test), what facebookresearch/Multi-SpatialMLLM computes on the
hot path of SURVEY.md §8(a).  Every function names the reference lines it follows; file
abbreviations as in SURVEY.md (IH = spatial_engine/utils/scannet_utils/handler/info_handler.py,
OPS = .../handler/ops.py, CFR = camera_movement/calculate_frames_relations.py,
MVI = utils/scannet_utils/make_visibility_info.py, CME = camera_movement/
camera_movement_engine_train_val.py, VC_C = visual_correspondence/...coor_2_coor.py,
OM_C = object_movement/single_object_movement_engine_coord.py).

Pinning: the reference ships no tests or golden vectors, so the oracle is pinned
against the reference itself, imported unmodified in the build container
(oracle/ref_harness.py): tests/test_oracle_vs_reference.py compares live, and
oracle/gen_golden.py froze reference outputs into tests/golden/*.npz which
tests/test_oracle_golden.py replays anywhere (the GPU box has no /root/reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It works on plain arrays: no file I/O, no image decoding, no scene handler.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

DEPTH_VALUE_SCALE = 0.001   # IH:76 depth_value_scale, OPS:292-294


# --------------------------------------------------------------------------------------
# a1 / a2: world -> image projection
# --------------------------------------------------------------------------------------
def project_points(points: np.ndarray, K: np.ndarray, E: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """IH:46-72.  points [N,4] homogeneous, K 4x4 intrinsic, E 4x4 camera->world.

    Returns un-rounded pixel coordinates [N,2] and the signed camera-space depth [N].
    No z>0 guard: z == 0 gives inf/nan, z < 0 a mirrored coordinate (filtered later).
    """
    E_inv = np.linalg.inv(E)                 # IH:57
    cam = E_inv @ points.T                   # IH:60   4xN
    depth = cam[2, :]                        # IH:63
    img = K @ cam                            # IH:66
    with np.errstate(divide="ignore", invalid="ignore"):
        img = img / img[2, :]                # IH:69 (in place there; a fresh array here, same values)
    return img.T[:, :2], depth               # IH:72


def homogeneous(points_3d: np.ndarray) -> np.ndarray:
    """IH:328-330: promote (3,) to (1,3) and append a column of ones."""
    p = np.expand_dims(points_3d, axis=0) if points_3d.ndim == 1 else points_3d
    return np.hstack([p, np.ones((p.shape[0], 1))])


def aligned_extrinsic(A: np.ndarray, E: np.ndarray) -> np.ndarray:
    """IH:113-124: axis-align matrix times camera->world."""
    return A @ E


def project_3d_point_to_image(points_3d, K, E_c2w) -> Tuple[np.ndarray, np.ndarray]:
    """IH:313-335 with the extrinsic already chosen by the caller (A@E when align=True)."""
    return project_points(homogeneous(np.asarray(points_3d, dtype=np.float64)), K, E_c2w)


# --------------------------------------------------------------------------------------
# a3 / a4 / a5: bounds + depth-buffer visibility
# --------------------------------------------------------------------------------------
def check_point_in_image_boundary(uv: np.ndarray, image_hw: Tuple[int, int]) -> np.ndarray:
    """IH:337-344: bounds test on the *float* coordinates against the colour size (H, W)."""
    H, W = image_hw
    return (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)


def depth_pixel_index(uv: np.ndarray, image_hw, depth_hw) -> Tuple[np.ndarray, np.ndarray]:
    """IH:359-366: colour-pixel coordinate -> clipped depth-pixel index (int64, half-to-even)."""
    H, W = image_hw
    DH, DW = depth_hw
    scale_x = DW / W
    scale_y = DH / H
    with np.errstate(invalid="ignore"):
        xi = np.round(uv[:, 0] * scale_x).astype(int)
        yi = np.round(uv[:, 1] * scale_y).astype(int)
    xi = np.clip(xi, 0, DW - 1)
    yi = np.clip(yi, 0, DH - 1)
    return xi, yi


def check_point_visibility_by_depth(uv, depth, depth_image, image_hw,
                                    scale: float = DEPTH_VALUE_SCALE) -> np.ndarray:
    """IH:346-373: strict ``0 < depth < depth_image[yi, xi] * scale``; zero pixels never pass."""
    xi, yi = depth_pixel_index(uv, image_hw, depth_image.shape[:2])
    dv = depth_image[yi, xi] * scale
    with np.errstate(invalid="ignore"):
        return (depth > 0) & (depth < dv)


def check_point_visibility(uv, depth, depth_image, image_hw, scale: float = DEPTH_VALUE_SCALE) -> np.ndarray:
    """IH:375-386 (``scale`` = the handler's depth_value_scale, IH:76)."""
    return check_point_in_image_boundary(uv, image_hw) & \
        check_point_visibility_by_depth(uv, depth, depth_image, image_hw, scale)


def vertex_visibility(points_xyz, K, E_aligned, depth_image, image_hw, scale: float = DEPTH_VALUE_SCALE):
    """HOT LOOP 1 body (CFR:152-157 == MVI:93-100): mask of vertices visible in one image."""
    uv, d = project_3d_point_to_image(points_xyz, K, E_aligned)
    return check_point_visibility(uv, d, depth_image, image_hw, scale), uv, d


# --------------------------------------------------------------------------------------
# a7: depth image -> 3D points
# --------------------------------------------------------------------------------------
def project_mask_to_3d(depth_image, K, E, mask=None, A=None, color_image=None,
                       return_index: bool = False):
    """OPS:235-329.  ``mask`` spans the colour grid; None needs ``color_image`` (OPS:268-269).

    Returns [M,3] or [M,6] float64 in row-major mask order, zero-depth pixels dropped.
    ``return_index`` additionally returns the kept (my, mx) -- an oracle-side convenience.
    """
    if mask is None:
        mask = np.ones(color_image.shape[:2], dtype=bool)     # AttributeError if both None, as in OPS
    scale_y = depth_image.shape[0] / mask.shape[0]            # OPS:272-273
    scale_x = depth_image.shape[1] / mask.shape[1]
    my, mx = np.where(mask)                                   # OPS:276-278
    dy = np.clip(np.round(my * scale_y).astype(int), 0, depth_image.shape[0] - 1)   # OPS:285-290
    dx = np.clip(np.round(mx * scale_x).astype(int), 0, depth_image.shape[1] - 1)
    d = depth_image[dy, dx] * 0.001                           # OPS:292-294
    valid = d > 0                                             # OPS:297-300
    d, mx, my = d[valid], mx[valid], my[valid]
    pix = np.vstack((mx * d, my * d, d, np.ones_like(d)))     # OPS:303-310
    cam = np.dot(np.linalg.inv(K), pix)                       # OPS:313
    world = np.dot(E, cam)                                    # OPS:316
    if A is not None:
        world = np.dot(A, world)                              # OPS:319-320
    out = world[:3].T
    if color_image is not None:
        out = np.hstack((out, color_image[my, mx]))           # OPS:323-327
    if return_index:
        return out, my, mx
    return out


# --------------------------------------------------------------------------------------
# composite frame pair: a7 -> a2 -> a5  (BASELINE.json north_star)
# --------------------------------------------------------------------------------------
def frame_pair(depth1, depth2, K, E1, E2, A, image_hw, color1=None) -> Dict[str, np.ndarray]:
    """Back-project every colour pixel of frame 1, move it into frame 2, test it there.

    Dense per-pixel outputs on the colour grid (row-major):
      valid  bool   depth-1 sample > 0                      (OPS:297)
      xyz    f64x3  aligned world point (nan where !valid)
      uv2    f64x2  projection into frame 2 (IH:69)         depth2 f64 camera-2 z (IH:63)
      xi,yi  int64  clipped depth-2 pixel index (IH:362-366)
      vis    bool   valid & in-bounds & depth test (IH:375-386)
    """
    H, W = image_hw
    mask = np.ones((H, W), dtype=bool)
    pts, my, mx = project_mask_to_3d(depth1, K, E1, mask, A, color1, return_index=True)
    E2a = aligned_extrinsic(A, E2)
    uv, d2 = project_3d_point_to_image(pts[:, :3], K, E2a)
    inb = check_point_in_image_boundary(uv, image_hw)
    xi, yi = depth_pixel_index(uv, image_hw, depth2.shape[:2])
    vis_c = inb & check_point_visibility_by_depth(uv, d2, depth2, image_hw)
    flat = my * W + mx
    P = H * W
    out = {
        "valid": np.zeros(P, dtype=bool),
        "xyz": np.full((P, 3), np.nan),
        "uv2": np.full((P, 2), np.nan),
        "depth2": np.full(P, np.nan),
        "xi": np.zeros(P, dtype=np.int64),
        "yi": np.zeros(P, dtype=np.int64),
        "vis": np.zeros(P, dtype=bool),
    }
    out["valid"][flat] = True
    out["xyz"][flat] = pts[:, :3]
    out["uv2"][flat] = uv
    out["depth2"][flat] = d2
    out["xi"][flat] = xi
    out["yi"][flat] = yi
    out["vis"][flat] = vis_c
    if color1 is not None:
        out["rgb"] = np.zeros((P, 3), dtype=np.uint8)
        out["rgb"][flat] = pts[:, 3:6].astype(np.uint8)
    out["n_valid"] = int(out["valid"].sum())
    out["n_vis"] = int(out["vis"].sum())
    return out


# --------------------------------------------------------------------------------------
# a9 / a10 / a11: overlap + per-frame angles + pair table
# --------------------------------------------------------------------------------------
def calculate_camera_overlap(m1: np.ndarray, m2: np.ndarray):
    """CFR:102-137 (CPU branch): |a & b| / |a | b| * 100, nan when the union is empty."""
    union = np.logical_or(m1, m2)
    inter = np.logical_and(m1, m2)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.sum(inter) / np.sum(union) * 100


def extract_yaw_pitch(R: np.ndarray):
    """CFR:86-100: angles (degrees) of the camera's forward (z) axis."""
    R3 = R[:3, :3] if R.shape == (4, 4) else R
    z = R3[:, 2]
    yaw = np.degrees(np.arctan2(z[1], z[0]))
    pitch = np.degrees(np.arcsin(z[2] / np.linalg.norm(z)))
    return yaw, pitch


def valid_image_ids(E: Dict[str, np.ndarray]) -> List[str]:
    """IH:184-189, 409-418: frames whose pose holds inf/nan are dropped."""
    return [k for k, e in E.items() if not (np.any(np.isinf(e)) or np.any(np.isnan(e)))]


def scene_visibility_masks(points_xyz, K, A, E: Dict[str, np.ndarray], depth: Dict[str, np.ndarray],
                           image_hw) -> Dict[str, np.ndarray]:
    """HOT LOOP 1 (CFR:152-164 == MVI:93-100) for every valid frame of a scene."""
    masks = {}
    for image_id in valid_image_ids(E):
        m, _, _ = vertex_visibility(points_xyz, K, aligned_extrinsic(A, E[image_id]), depth[image_id], image_hw)
        masks[image_id] = m
    return masks


def frames_relations_scene(points_xyz, K, A, E, depth, image_hw):
    """CFR.process_scene (CFR:139-197): {(id1,id2): {overlap,distance,yaw,pitch}}, i<j in key order."""
    ids = valid_image_ids(E)
    masks = scene_visibility_masks(points_xyz, K, A, E, depth, image_hw)
    yaw, pitch, pos = {}, {}, {}
    for image_id in ids:
        Ea = aligned_extrinsic(A, E[image_id])
        yaw[image_id], pitch[image_id] = extract_yaw_pitch(Ea)     # CFR:166
        pos[image_id] = Ea[:3, 3]                                   # CFR:172
    table = {}
    for i, id1 in enumerate(ids):
        for j in range(i + 1, len(ids)):
            id2 = ids[j]
            table[(id1, id2)] = {
                "overlap": calculate_camera_overlap(masks[id1], masks[id2]),
                "distance": np.linalg.norm(pos[id2] - pos[id1]),    # CFR:183
                "yaw": yaw[id2] - yaw[id1],                         # CFR:181 (raw difference)
                "pitch": pitch[id2] - pitch[id1],                   # CFR:182
            }
    return table


def visibility_index_scene(points_xyz, K, A, E, depth, image_hw):
    """MVI.process_scene (MVI:75-125): index lists both ways."""
    masks = scene_visibility_masks(points_xyz, K, A, E, depth, image_hw)
    n = points_xyz.shape[0]
    image_to_points = {}
    sets = [set() for _ in range(n)]
    for image_id, m in masks.items():
        idx = np.where(m)[0]
        image_to_points[image_id] = idx.tolist()                    # MVI:103-104
        for v in idx:
            sets[v].add(image_id)                                   # MVI:107-108
    point_to_images = {i: sorted(list(s)) for i, s in enumerate(sets)}   # MVI:116-118
    return {"image_to_points": image_to_points, "point_to_images": point_to_images}


# --------------------------------------------------------------------------------------
# a13: correspondence extraction (geometry part; RNG draws stay with the caller)
# --------------------------------------------------------------------------------------
def common_visible_points(points1: Sequence[int], points2: Sequence[int]) -> np.ndarray:
    """VC_C:303: sorted unique intersection of the two frames' visible-vertex lists."""
    return np.intersect1d(points1, points2)


def point_2d_in_image(point_xyz, K, E_aligned, depth_image, image_hw, check_visible=True):
    """IH:291-305 (a6) for one vertex: ([0 or 1, 2] uv, [0 or 1] depth)."""
    uv, d = project_3d_point_to_image(np.asarray(point_xyz, dtype=np.float64)[:3], K, E_aligned)
    if check_visible:
        m = check_point_visibility(uv, d, depth_image, image_hw)
        uv, d = uv[m], d[m]
    return uv, d


def normalised_coordinate(uv_row, image_hw) -> Tuple[int, int]:
    """VC_C:341-344 / DE_D:216-217: builtin round (half-to-even) of u/W*1000, v/H*1000."""
    H, W = image_hw
    return round((uv_row[0] / W) * 1000), round((uv_row[1] / H) * 1000)


# --------------------------------------------------------------------------------------
# a14: relative camera pose (CME:153-225), RNG-free: ``swap`` is the 50 % coin of CME:163
# --------------------------------------------------------------------------------------
def relative_pose_answer_values(E1_aligned, E2_aligned, yaw, pitch, swap: bool) -> dict:
    yaw_angle, pitch_angle = float(yaw), float(pitch)
    if swap:                                                        # CME:163-166
        yaw_angle, pitch_angle = -yaw_angle, -pitch_angle
        E1_aligned, E2_aligned = E2_aligned, E1_aligned
    if abs(yaw_angle) > 180:                                        # CME:168-172
        yaw_angle = yaw_angle - 360 if yaw_angle > 0 else yaw_angle + 360
    rel = np.linalg.inv(E1_aligned) @ E2_aligned                    # CME:185-186
    d = rel[:3, 3]                                                  # CME:189
    return {
        "x_movement": "right" if d[0] > 0 else "left",              # CME:209-214
        "y_movement": "down" if d[1] > 0 else "up",
        "z_movement": "forward" if d[2] > 0 else "backward",
        "yaw_movement": "left" if yaw_angle > 0 else "right",
        "pitch_movement": "up" if pitch_angle > 0 else "down",
        "x_distance": int(abs(d[0]) * 1000),                        # CME:215-223 (truncation)
        "y_distance": int(abs(d[1]) * 1000),
        "z_distance": int(abs(d[2]) * 1000),
        "yaw_angle": int(abs(yaw_angle)),
        "pitch_angle": int(abs(pitch_angle)),
        "x_value": int(d[0] * 1000),
        "y_value": int(d[1] * 1000),
        "z_value": int(d[2] * 1000),
        "total_distance": int(np.linalg.norm(d) * 1000),
        "displacement_vector": d.tolist(),
    }


# --------------------------------------------------------------------------------------
# a15 - a18: TAPVid-3D track geometry
# --------------------------------------------------------------------------------------
def project_point(point_3d, fx_fy_cx_cy, image_height, image_width):
    """OM_C:293-315: normalised pinhole projection of a camera-space point, None when outside."""
    fx, fy, cx, cy = fx_fy_cx_cy
    x, y, z = point_3d
    u = (fx * x / (z + 1e-8)) + cx
    v = (fy * y / (z + 1e-8)) + cy
    un = u / image_width
    vn = v / image_height
    if not (0 <= un < 1 and 0 <= vn < 1 and z > 0):
        return None
    return [un, vn]


def tracks_cam_to_world(tracks_xyz, extrinsics_w2c):
    """OM_C:446-454: batched inverse + einsum over homogeneous tracks -> [T,P,3] world."""
    T, P, _ = tracks_xyz.shape
    c2w = np.linalg.inv(extrinsics_w2c)
    hom = np.concatenate([tracks_xyz, np.ones((T, P, 1))], axis=2)
    return np.einsum("nij,nkj->nki", c2w, hom)[..., :3]


def point_pair_distances(tracks_world, visibility, point_idx):
    """OM_C:476-498: all i<j pairs of the frames where the point is visible, with ||p_j - p_i||."""
    vf = np.where(visibility[:, point_idx])[0]
    if len(vf) < 2:
        return np.zeros(0), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    ii, jj = np.triu_indices(len(vf), k=1)            # same (i, j) order as the nested comprehension
    f1, f2 = vf[ii], vf[jj]
    d = np.linalg.norm(tracks_world[f2, point_idx] - tracks_world[f1, point_idx], axis=1)
    return d, f1, f2


def object_displacement(tracks_world, tracks_cam, extrinsics_w2c, fx_fy_cx_cy, image_hw,
                        frame1, frame2, point_index, not_moving_threshold=0.01,
                        camera_not_moving_threshold=0.01) -> Optional[dict]:
    """OM_C:324-376 + the numeric fields of OM_C:387-399 (text templating left out)."""
    H, W = image_hw
    dw = tracks_world[frame2, point_index] - tracks_world[frame1, point_index]
    dist = np.linalg.norm(dw)
    if dist < not_moving_threshold:                                  # OM_C:334-339
        moving, dist, dw = False, 0, np.zeros(3)
    else:
        moving = True
    c1 = np.linalg.inv(extrinsics_w2c[frame1])                       # OM_C:343-344
    c2 = np.linalg.inv(extrinsics_w2c[frame2])
    cam_moving = not (np.linalg.norm(c2[:3, 3] - c1[:3, 3]) < camera_not_moving_threshold)
    d_cam1 = (extrinsics_w2c[frame1] @ np.concatenate([dw, [0]]))[:3]   # OM_C:354-356
    p1 = project_point(tracks_cam[frame1, point_index], fx_fy_cx_cy, H, W)
    p2 = project_point(tracks_cam[frame2, point_index], fx_fy_cx_cy, H, W)
    if p1 is None or p2 is None:                                     # OM_C:360-362
        return None
    return {
        "p1": (round(p1[0] * 1000), round(p1[1] * 1000)),            # OM_C:364-365
        "p2": (round(p2[0] * 1000), round(p2[1] * 1000)),
        "total_distance_text": round(dist * 1000),                   # OM_C:372
        "vector_text": tuple(round(v * 1000) for v in d_cam1),       # OM_C:373-375
        "gt_total_distance": int(dist * 1000),                       # OM_C:393
        "gt_vector": d_cam1.tolist(),
        "point_moving": int(moving),
        "cam_moving": int(cam_moving),
    }


# --------------------------------------------------------------------------------------
# object perception: object visibility (COVIS), coverage search on boolean masks (COV)
#   COVIS = object_perception/compute_object_visibility.py, COV = object_perception/single_object_coverage_finder.py
# --------------------------------------------------------------------------------------
def object_visibility(object_points: Dict[int, Sequence[int]], image_to_points: Dict[str, Sequence[int]],
                      image_ids: Sequence[str]) -> dict:
    """COVIS:103-150 with Python sets: per (object, image) intersection size against max(1, int(0.05 * n))."""
    result = {"object_to_images": {}, "image_to_objects": {}}
    for obj, pts in object_points.items():
        pts = set(int(p) for p in pts)
        if not pts:
            continue
        threshold = max(1, int(0.05 * len(pts)))                           # COVIS:112
        for image_id in image_ids:
            if image_id not in image_to_points:
                continue
            count = len(set(image_to_points[image_id]) & pts)              # COVIS:119-121
            if count >= threshold:
                vis = (count / len(pts)) * 100.0
                result["object_to_images"].setdefault(obj, []).append(
                    {"image_id": image_id, "intersection_count": count, "visibility": vis})
                result["image_to_objects"].setdefault(image_id, []).append(
                    {"object_id": obj, "intersection_count": count, "visibility": vis})
    return result


def compute_coverage(scene_pts, mask, axis):
    """COV:56-65."""
    if not mask.any():
        return None
    coords = scene_pts[mask][:, axis]
    return max(coords) - min(coords)


def coverage_minimal_combinations(scene_pts, object_points, images: Sequence[str], image_to_points, axis, target, tolerance=0.1,
                                  max_images=5, rng=None):
    """COV:76-220 on boolean masks over all scene vertices, as upstream carries them: level-by-level search with the
    superset pruning, the cumulative-union early prune, the 25-image cap and the 5000-node cap."""
    import random as _random
    rng = rng or _random
    n_pts = len(scene_pts)
    obj_mask = np.zeros(n_pts, dtype=bool)
    obj_mask[np.asarray(object_points, dtype=np.int64)] = True                # COV:99-100
    masks = {}
    for img in images:                                                       # COV:103-112
        if img not in image_to_points:
            continue
        m = np.zeros(n_pts, dtype=bool)
        m[np.asarray(image_to_points[img], dtype=np.int64)] = True
        masks[img] = np.logical_and(m, obj_mask)
    valid = list(masks)
    if len(valid) > 25:                                                      # COV:118-119
        valid = rng.sample(valid, 25)
    n = len(valid)

    def covers(mask):                                                        # COV:67-73, 144-146
        cov = compute_coverage(scene_pts, mask, axis)
        return cov is not None and abs(cov - target) <= tolerance * target

    tail = [None] * n                                                        # COV:122-127
    for i in range(n - 1, -1, -1):
        tail[i] = masks[valid[i]].copy() if i == n - 1 else np.logical_or(masks[valid[i]], tail[i + 1])
    found = []
    level = []
    for i, img in enumerate(valid):                                          # COV:151-155
        members = np.zeros(n, dtype=bool)
        members[i] = True
        level.append(([img], masks[img], i, members))
    first_layer, solutions, k = [], {}, 1
    while k <= max_images and level:                                         # COV:163
        expand, fresh = [], []
        for comb, union, last, members in level:
            if any(np.array_equal(np.logical_and(m, members), m) for m in found):   # COV:132-142
                continue
            if covers(union):                                                # COV:174-179
                fresh.append(members)
                solutions.setdefault(k, []).append(tuple(comb))
            else:
                if last < n - 1 and not covers(np.logical_or(tail[last], union)):   # COV:181-184
                    continue
                expand.append((comb, union, last, members))
                if k == 1:
                    first_layer.append((comb, union, last, members))
        found.extend(fresh)
        nxt = []
        if k < max_images:                                                   # COV:195-205
            for comb, union, last, members in expand:
                for c1, u1, last1, m1 in first_layer:
                    if last1 > last:
                        nxt.append((comb + c1, np.logical_or(union, u1), last1, np.logical_or(m1, members)))
        if len(nxt) > 5000:                                                  # COV:207-209
            nxt = rng.sample(nxt, 5000)
        level = nxt
        k += 1
    return solutions


# --------------------------------------------------------------------------------------
# object movement: rigid grouping loss (OM_C:66-78), frame-pair mining (OM_C:465-567)
# --------------------------------------------------------------------------------------
def rigidity_loss(points: np.ndarray, smoothing_factor: float = 0.01) -> np.ndarray:
    """Cumulative thresholded change of all pairwise distances, frame after frame (OM_C:66-78)."""
    T, P, _ = points.shape
    loss = np.zeros((P, P))

    def dist_matrix(x):
        d = x[:, None, :] - x[None, :, :]
        return np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
    prev = dist_matrix(points[0])
    for t in range(1, T):
        cur = dist_matrix(points[t])
        change = np.abs(cur - prev)
        loss += np.where(change > smoothing_factor, change, 0)                # OM_C:46-47
        prev = cur
    return loss


def mine_frame_pairs(tracks_world, visibility, groups, npoints_per_group=5, npairs_per_bin=1e8, augment=True,
                     augment_ratio=1.0, rng=None, object_not_moving_threshold=0.01, future_frame_windows=1e8):
    """OM_C:465-567 with the reference's Python lists and loops (its frame-window test, which reads the distance where
    a frame index is meant, included)."""
    import random as _random
    rng = rng or _random
    sample_pairs = []
    for group in groups:
        rng.shuffle(group)
        for point_idx in group[:npoints_per_group]:
            frames = np.where(visibility[:, point_idx])[0]
            if len(frames) < 2:
                continue
            pairs = np.array([(i, j) for i in range(len(frames)) for j in range(i + 1, len(frames))])
            f1, f2 = frames[pairs[:, 0]], frames[pairs[:, 1]]
            dists = np.linalg.norm(tracks_world[f2, point_idx] - tracks_world[f1, point_idx], axis=1)
            static, moving = [], []
            for disp in zip(dists, f1, f2):
                if disp[1] > disp[0] + future_frame_windows:                 # OM_C:505-507 (sic)
                    continue
                (static if disp[0] < object_not_moving_threshold else moving).append(disp)
            selected = []
            if static:
                selected.append(rng.choice(static))
            if moving:
                moving.sort(key=lambda x: x[0])
                edges = np.histogram_bin_edges([d[0] for d in moving], bins=10)
                bins = [[] for _ in range(10)]
                for disp in moving:
                    bins[min(np.digitize(disp[0], edges) - 1, 9)].append(disp)
                npairs_per_bin = max(min(len(bins[4]), npairs_per_bin), 1)
                for b in bins:
                    selected.extend(rng.sample(b, npairs_per_bin) if len(b) > npairs_per_bin else b)
            for _, a, b in selected:
                sample_pairs.append({"point_index": point_idx, "frame1": a, "frame2": b})
    if augment:
        for s in rng.sample(sample_pairs, int(len(sample_pairs) * augment_ratio)):
            sample_pairs.append({"point_index": s["point_index"], "frame1": s["frame2"], "frame2": s["frame1"]})
    return sample_pairs
