"""Seeded synthetic RGB-D scenes and TAPVid-like track sets (inputs only, no geometry under test).

Follows the recipe of SURVEY.md §8(d): an axis-aligned 6x6x3 m room holding 8 boxes, N surface
vertices stored like ``aligned_points.npy`` (N x 6 float64, xyz + rgb), F cameras on a random walk
whose camera->world matrices are round-tripped through ``"%f"`` text like the reference's pose
files (extract_posed_images.py:139-142), a share of frames carrying ``-inf`` poses (the reference
drops them, info_handler.py:409-418), 16-bit millimetre depth rendered by analytic ray/box
intersection with noise and invalid (zero) pixels, and uint8 colour from a PCG64 stream.

Pure NumPy on purpose: the very same arrays feed the HIP path, the oracle and the imported
reference (golden generation), so no second generator can drift.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

ROOM = np.array([6.0, 6.0, 3.0])


@dataclasses.dataclass
class SynthScene:
    scene_id: str
    K: np.ndarray                 # 4x4 colour intrinsic (info_handler.py:97-98)
    A: np.ndarray                 # 4x4 world -> axis-aligned (info_handler.py:175-176)
    E: Dict[str, np.ndarray]      # image_id -> 4x4 camera->world, insertion ordered, may hold -inf
    points: np.ndarray            # [N, 6] float64 aligned xyz + rgb  (aligned_points.npy)
    depth: Dict[str, np.ndarray]  # image_id -> [DH, DW] uint16 millimetres
    color: Dict[str, np.ndarray]  # image_id -> [H, W, 3] uint8 (RGB)
    color_hw: Tuple[int, int]
    depth_hw: Tuple[int, int]
    boxes: np.ndarray             # [8, 2, 3] min/max corners in aligned space

    @property
    def image_ids(self) -> List[str]:
        return list(self.E.keys())

    @property
    def valid_image_ids(self) -> List[str]:
        return [k for k, e in self.E.items() if np.all(np.isfinite(e))]

    def objects(self):
        """The furniture boxes as labelled objects: ({obj: vertex indices}, {obj: aligned bbox (cx,cy,cz,dx,dy,dz)},
        {obj: category}); bbox dimensions are the extents of the object's own vertices, as ScanNet's export takes them."""
        xyz = self.points[:, :3]
        taken = np.zeros(len(xyz), dtype=bool)
        idx, bbox, cat = {}, {}, {}
        for o, (lo, hi) in enumerate(self.boxes):
            on = ((xyz >= lo - 1e-9) & (xyz <= hi + 1e-9)).all(axis=1) & ~taken
            taken |= on
            if not on.any():
                continue
            p = xyz[on]
            idx[o] = np.where(on)[0]
            bbox[o] = np.concatenate([(p.max(0) + p.min(0)) / 2, p.max(0) - p.min(0)])
            cat[o] = f"cabinet{o}"
        return idx, bbox, cat

    def info_dict(self) -> dict:
        """Scene record in the reference's scene-info layout (info_handler.py:7-30)."""
        images_info = {k: {"extrinsic_matrix": e} for k, e in self.E.items()}
        return {
            "num_posed_images": len(self.E),
            "intrinsic_matrix": self.K,
            "images_info": images_info,
            "axis_align_matrix": self.A,
            "num_objects": 0,
        }


def intrinsics_for(color_hw: Tuple[int, int]) -> np.ndarray:
    """ScanNet-like pinhole intrinsics scaled to the requested colour size."""
    H, W = color_hw
    if (H, W) == (968, 1296):
        fx = fy = 1170.19
        cx, cy = 647.75, 483.75
    else:
        fx = fy = 577.87 * W / 640.0
        cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    K = np.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    return K


def _roundtrip_f(m: np.ndarray) -> np.ndarray:
    """Mimic pose text files: ``%f`` keeps six decimals."""
    return np.array([[float("%f" % v) for v in row] for row in m], dtype=np.float64)


def _make_boxes(rng: np.random.Generator) -> np.ndarray:
    boxes = []
    for _ in range(8):
        size = rng.uniform([0.4, 0.4, 0.3], [1.4, 1.4, 1.8])
        lo_xy = rng.uniform([0.2, 0.2], ROOM[:2] - size[:2] - 0.2)
        lo = np.array([lo_xy[0], lo_xy[1], 0.0])
        boxes.append(np.stack([lo, lo + size]))
    return np.stack(boxes)


def _sample_box_surface(rng, lo, hi, n):
    """Uniform points on the six faces of an axis-aligned box."""
    d = hi - lo
    areas = np.array([d[1] * d[2], d[1] * d[2], d[0] * d[2], d[0] * d[2], d[0] * d[1], d[0] * d[1]])
    face = rng.choice(6, size=n, p=areas / areas.sum())
    p = lo + rng.random((n, 3)) * d
    axis = face // 2
    side = face % 2
    p[np.arange(n), axis] = np.where(side == 0, lo[axis], hi[axis])
    return p


def _sample_surface_points(rng, boxes, n):
    surf = [(np.zeros(3), ROOM)] + [(b[0], b[1]) for b in boxes]
    areas = []
    for lo, hi in surf:
        d = hi - lo
        areas.append(2 * (d[0] * d[1] + d[1] * d[2] + d[0] * d[2]))
    areas = np.array(areas)
    counts = rng.multinomial(n, areas / areas.sum())
    pts = [_sample_box_surface(rng, lo, hi, c) for (lo, hi), c in zip(surf, counts)]
    pts = np.concatenate(pts, axis=0)
    rng.shuffle(pts, axis=0)
    return pts


def _look_at(eye, target):
    """Camera->world with +z forward, +x right, +y down (ScanNet / OpenCV convention)."""
    fwd = target - eye
    fwd = fwd / np.linalg.norm(fwd)
    down = np.array([0.0, 0.0, -1.0])
    right = np.cross(down, fwd)
    right = right / np.linalg.norm(right)
    down = np.cross(fwd, right)
    E = np.eye(4)
    E[:3, 0], E[:3, 1], E[:3, 2], E[:3, 3] = right, down, fwd, eye
    return E


def render_depth(E_aligned, Kd, depth_hw, boxes):
    """z-depth (metres, float64) of the first surface seen through every depth pixel."""
    DH, DW = depth_hw
    xs, ys = np.meshgrid(np.arange(DW, dtype=np.float64), np.arange(DH, dtype=np.float64))
    dirs_cam = np.stack([(xs - Kd[0, 2]) / Kd[0, 0], (ys - Kd[1, 2]) / Kd[1, 1], np.ones_like(xs)], -1)
    R, o = E_aligned[:3, :3], E_aligned[:3, 3]
    d = dirs_cam.reshape(-1, 3) @ R.T                      # ray directions with cam-z == 1
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        # room: the camera sits inside, the hit is the exit of the slab intersection
        t1 = (0.0 - o) * inv
        t2 = (ROOM - o) * inv
        t_room = np.min(np.maximum(t1, t2), axis=1)
        t_best = t_room
        for b in boxes:
            ta = (b[0] - o) * inv
            tb = (b[1] - o) * inv
            tn = np.max(np.minimum(ta, tb), axis=1)
            tf = np.min(np.maximum(ta, tb), axis=1)
            hit = (tn <= tf) & (tn > 1e-6)
            t_best = np.where(hit & (tn < t_best), tn, t_best)
    return t_best.reshape(DH, DW)


def make_scene(seed: int, n_points: int = 131072, n_frames: int = 64,
               color_hw: Tuple[int, int] = (480, 640), depth_hw: Tuple[int, int] = (480, 640),
               invalid_pose_frac: float = 0.02, zero_frac: float = 0.07, noise_mm: float = 5.0,
               frame_step: int = 5, with_color: bool = True, scene_id: Optional[str] = None,
               walk_step: float = 0.15, target_jitter: float = 0.8, trajectory: str = "jitter",
               target_step: float = 0.25) -> SynthScene:
    """``trajectory="jitter"`` (default): every frame looks at the room centre +- an independent jitter.
    ``trajectory="sweep"``: the look-at target itself performs a slow random walk (``target_step`` metres per frame), like
    a hand-held scan -- neighbouring frames overlap strongly, distant ones little, so one scene populates every overlap
    bin the reference samples from (VC_C:485-505: 6..35 %)."""
    if trajectory not in ("jitter", "sweep"):
        raise ValueError(f"unknown trajectory {trajectory!r}")
    rng = np.random.default_rng(np.random.PCG64(seed))
    boxes = _make_boxes(rng)
    K = intrinsics_for(color_hw)
    H, W = color_hw
    DH, DW = depth_hw
    Kd = K.copy()
    Kd[0] *= DW / W
    Kd[1] *= DH / H

    yaw = rng.uniform(-np.pi, np.pi)
    A = np.eye(4)
    A[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
    A[:3, 3] = rng.uniform(-2.0, 2.0, 3)
    A = _roundtrip_f(A)
    A_inv = np.linalg.inv(A)

    xyz = _sample_surface_points(rng, boxes, n_points)
    rgb = rng.integers(0, 256, size=(n_points, 3)).astype(np.float64)
    points = np.concatenate([xyz, rgb], axis=1)

    eye = np.array([rng.uniform(1.0, 5.0), rng.uniform(1.0, 5.0), 1.5])
    E, depth, color = {}, {}, {}
    n_bad = int(round(invalid_pose_frac * n_frames))
    bad = set(rng.choice(np.arange(1, n_frames), size=n_bad, replace=False).tolist()) if n_bad else set()
    for f in range(n_frames):
        image_id = f"{f * frame_step:05d}"
        eye = np.clip(eye + rng.normal(0, walk_step, 3) * [1, 1, 0.2], [0.6, 0.6, 1.2], [5.4, 5.4, 1.9])
        for _ in range(8):   # keep the camera out of the boxes
            inside = [(eye > b[0] - 0.1).all() and (eye < b[1] + 0.1).all() for b in boxes]
            if not any(inside):
                break
            eye = np.array([rng.uniform(0.6, 5.4), rng.uniform(0.6, 5.4), 1.9])
        if trajectory == "sweep":
            if f == 0:
                target = ROOM / 2 + rng.normal(0, target_jitter, 3) * [1, 1, 0.4]
            else:
                target = np.clip(target + rng.normal(0, target_step, 3) * [1, 1, 0.4], [0.3, 0.3, 0.2], ROOM - [0.3, 0.3, 0.2])
        else:
            target = ROOM / 2 + rng.normal(0, target_jitter, 3) * [1, 1, 0.4]
        E_al = _look_at(eye, target)
        E_f = _roundtrip_f(A_inv @ E_al)
        E_al = A @ E_f
        z = render_depth(E_al, Kd, depth_hw, boxes)
        mm = z * 1000.0 + rng.normal(0, noise_mm, z.shape)
        mm = np.clip(np.rint(mm), 0, 65535).astype(np.uint16)
        mm[rng.random(mm.shape) < zero_frac] = 0
        depth[image_id] = mm
        if with_color:
            color[image_id] = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        if f in bad:
            E_f = np.full((4, 4), -np.inf)
        E[image_id] = E_f
    return SynthScene(scene_id or f"scene{seed:04d}_00", K, A, E, points, depth, color,
                      color_hw, depth_hw, boxes)


@dataclasses.dataclass
class SynthTracks:
    """TAPVid-3D shaped sample (keys as read at single_object_movement_engine_coord.py:441-444)."""
    scene_id: str
    tracks_XYZ: np.ndarray       # [T, P, 3] camera-space
    visibility: np.ndarray       # [T, P] bool
    extrinsics_w2c: np.ndarray   # [T, 4, 4]
    fx_fy_cx_cy: np.ndarray      # (4,)
    image_hw: Tuple[int, int]


def make_tracks(seed: int, T: int = 300, P: int = 256, n_groups: int = 8,
                image_hw: Tuple[int, int] = (512, 512)) -> SynthTracks:
    rng = np.random.default_rng(np.random.PCG64(seed))
    fx_fy_cx_cy = np.array([500.0, 500.0, 256.0, 256.0])
    # world-space points: half static background, half on rigid groups that translate + rotate
    n_static = P // 2
    base = rng.uniform([-1.5, -1.0, 2.0], [1.5, 1.0, 5.0], size=(P, 3))
    group = np.full(P, -1)
    group[n_static:] = rng.integers(0, n_groups, size=P - n_static)
    world = np.repeat(base[None], T, axis=0)
    t = np.arange(T)[:, None]
    for g in range(n_groups):
        idx = np.where(group == g)[0]
        if len(idx) == 0:
            continue
        vel = rng.normal(0, 0.004, 3)
        amp = rng.uniform(0.0, 0.3)
        centre = base[idx].mean(0)
        ang = amp * np.sin(t[:, 0] * rng.uniform(0.01, 0.05))
        c, s = np.cos(ang), np.sin(ang)
        rel = base[idx] - centre
        rot = np.stack([c[:, None] * rel[None, :, 0] - s[:, None] * rel[None, :, 1],
                        s[:, None] * rel[None, :, 0] + c[:, None] * rel[None, :, 1],
                        np.repeat(rel[None, :, 2], T, 0)], -1)
        world[:, idx] = centre + rot + t[:, :, None] * vel
    w2c = np.repeat(np.eye(4)[None], T, axis=0)
    cam_pos = np.cumsum(rng.normal(0, 0.003, (T, 3)), axis=0)
    still = rng.random() < 0.3
    for k in range(T):
        a = 0.0 if still else 0.1 * np.sin(k * 0.02)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        c2w = np.eye(4)
        c2w[:3, :3] = R
        c2w[:3, 3] = 0.0 if still else cam_pos[k]
        w2c[k] = np.linalg.inv(c2w)
    hom = np.concatenate([world, np.ones((T, P, 1))], -1)
    cam = np.einsum("nij,nkj->nki", w2c, hom)[..., :3]
    vis = np.ones((T, P), dtype=bool)
    for p in range(P):   # visibility comes in runs
        k = 0
        state = rng.random() < 0.8
        while k < T:
            run = int(rng.geometric(0.05))
            vis[k:k + run, p] = state
            k += run
            state = rng.random() < 0.8
    return SynthTracks(f"synth_tracks_{seed:04d}", cam, vis, w2c, fx_fy_cx_cy, image_hw)


def write_scannet_layout(scenes: Sequence["SynthScene"], root: str, info_name: str = "scenes_info.pkl",
                         compress_level: int = 1, jpeg_for_every_image: bool = False,
                         link_identical: bool = False) -> Dict[str, str]:
    """Write synthetic scenes to disk the way the reference's pipeline finds ScanNet:
    ``<root>/posed_images/<scene>/<image>.png`` (16-bit depth, extract_posed_images.py:118-123) and ``<image>.jpg``,
    ``<root>/scannet_instance_data/<scene>/aligned_points.npy`` (N x 6 float64, batch_load_scannet_data.py:201) and the
    scene-info pickle (info_handler.py:7-30).  A scene without colour frames gets a flat grey JPEG of the right size for its
    first image (only its header is ever read: IH:133-139 takes the image size from it; ``jpeg_for_every_image``: for all
    of them -- the dataset builders look the size up per row, CME:238-239).  Returns the paths to hand to
    ``SceneInfoHandler(info_path, posed_images_root=..., instance_data_root=...)``.
    ``link_identical``: a depth array that appears under several image ids (or scenes) -- the SAME ndarray object -- is
    encoded once and hard-linked under the other names (ScanNet-sized inputs for the sweep benchmarks in seconds instead of
    minutes; every name is still opened, read and inflated on its own)."""
    import os
    import pickle
    from PIL import Image
    from concurrent.futures import ThreadPoolExecutor
    posed, inst = os.path.join(root, "posed_images"), os.path.join(root, "scannet_instance_data")
    infos, jobs = {}, []
    first_png: Dict[int, str] = {}
    links = []
    for sc in scenes:
        os.makedirs(os.path.join(posed, sc.scene_id), exist_ok=True)
        os.makedirs(os.path.join(inst, sc.scene_id), exist_ok=True)
        np.save(os.path.join(inst, sc.scene_id, "aligned_points.npy"), sc.points)
        H, W = sc.color_hw
        for n, (image_id, d) in enumerate(sc.depth.items()):
            png = os.path.join(posed, sc.scene_id, f"{image_id}.png")
            if link_identical and id(d) in first_png:
                links.append((first_png[id(d)], png))
            else:
                first_png[id(d)] = png
                jobs.append((Image.fromarray(np.ascontiguousarray(d, dtype=np.uint16)), png, {"compress_level": compress_level}))
            col = sc.color.get(image_id) if sc.color else None
            if col is not None:
                jobs.append((Image.fromarray(col), os.path.join(posed, sc.scene_id, f"{image_id}.jpg"), {"quality": 90}))
            elif n == 0 or jpeg_for_every_image:
                jobs.append((Image.new("RGB", (W, H), (128, 128, 128)), os.path.join(posed, sc.scene_id, f"{image_id}.jpg"),
                             {"quality": 50}))
        infos[sc.scene_id] = sc.info_dict()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:      # the encoders release the interpreter lock
        list(ex.map(lambda j: j[0].save(j[1], **j[2]), jobs))
    for src, dst in links:
        if os.path.exists(dst):
            os.remove(dst)
        os.link(src, dst)
    info_path = os.path.join(inst, info_name)
    with open(info_path, "wb") as f:
        pickle.dump(infos, f)
    return {"info_path": info_path, "posed_images_root": posed, "instance_data_root": inst}
