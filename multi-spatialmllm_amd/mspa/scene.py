"""Scene-level products on top of the kernels: the reference's per-scene scripts as device pipelines.

``SceneOnDevice`` keeps one scene resident (vertices, depth frames, camera tables) and produces
what ``CFR.process_scene`` (CFR:139-197) and ``MVI.process_scene`` (MVI:75-125) return -- the pair
table and the visibility index -- from K1 (vertex visibility), K2 (pair overlap) and K4 (pair pose),
plus the correspondence primitives of the visual-correspondence head (VC_C:280-344).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import engine


def valid_image_ids(E: Dict[str, np.ndarray]) -> List[str]:
    """Frames whose pose holds inf/nan are dropped before any projection (IH:184-189, 409-418)."""
    keys = list(E)
    if not keys:
        return []
    try:
        ok = np.isfinite(np.stack([E[k] for k in keys]).astype(np.float64, copy=False)).all(axis=(1, 2))   # one pass, not one per frame
    except ValueError:                        # ragged pose shapes: let the per-frame form decide
        return [k for k, e in E.items() if np.all(np.isfinite(np.asarray(e, dtype=np.float64)))]
    return [k for k, o in zip(keys, ok.tolist()) if o]


class SceneOnDevice:
    def __init__(self, K: np.ndarray, A: np.ndarray, E: Dict[str, np.ndarray], depth: Dict[str, np.ndarray],
                 image_hw: Tuple[int, int], points_xyz: Optional[np.ndarray] = None, device="cuda",
                 color: Optional[Dict[str, np.ndarray]] = None, depth_scale: float = 0.001):
        self.K, self.A = np.asarray(K, np.float64), np.asarray(A, np.float64)
        self.depth_scale = float(depth_scale)       # the handler's depth_value_scale (IH:76): metres per depth-image unit
        self.ids = valid_image_ids(E)
        self.index = {k: n for n, k in enumerate(self.ids)}
        self.image_hw = tuple(int(v) for v in image_hw)
        self.device = device
        self.E_aligned = [self.A @ np.asarray(E[k], np.float64) for k in self.ids]        # IH:113-124
        if self.ids:
            self.depth = engine.depth_to_device(np.stack([depth[k] for k in self.ids]), device)
        else:       # no frame with a finite pose: the reference returns empty tables for such a scene (CFR:176-189 loops over nothing)
            any_frame = next(iter(depth.values()), None)
            dh, dw = (any_frame.shape if any_frame is not None else self.image_hw)
            self.depth = torch.zeros((0, int(dh), int(dw)), dtype=torch.int16, device=device)
        self.frame_mats = torch.from_numpy(engine.frame_matrices(self.K, self.A, [E[k] for k in self.ids])).to(device)
        self.cam_mats = torch.from_numpy(engine.camera_matrices(self.K, self.E_aligned)).to(device)
        self.rgb = None
        if color:
            self.rgb = torch.from_numpy(np.stack([color[k] for k in self.ids])).to(device)
        self.xyz = None
        if points_xyz is not None:
            self.xyz = torch.from_numpy(np.ascontiguousarray(np.asarray(points_xyz, np.float64)[:, :3])).to(device)
        self._vis = None
        self._pose = None

    @classmethod
    def from_resident(cls, K, A, ids, E_aligned, depth, frame_mats, cam_mats, xyz, image_hw, device, pose_tables=None):
        """A scene whose tensors are already on the device (mspa/upload.py: staged through pinned memory on a copy stream).
        ``pose_tables`` = (E_aligned [F,16], yaw [F], pitch [F]) device tensors when the uploader staged them as well."""
        self = cls.__new__(cls)
        self.K, self.A = np.asarray(K, np.float64), np.asarray(A, np.float64)
        self.depth_scale = 0.001
        self.ids = list(ids)
        self.index = {k: n for n, k in enumerate(self.ids)}
        self.image_hw = tuple(int(v) for v in image_hw)
        self.device = device
        self.E_aligned = list(E_aligned)
        self.depth, self.frame_mats, self.cam_mats, self.xyz = depth, frame_mats, cam_mats, xyz
        self.rgb = None
        self._vis = None
        # Nothing may be COMPUTED from the resident tensors here: this runs on the uploader's thread while the copies are still
        # queued on the copy stream, and a kernel launched now (e.g. ``cam_mats[:, 0, :].contiguous()`` for K4's inverse table)
        # would read the slot's previous scene.  The inverse table is cut out by ``pose_tables()``, on the consumer's stream,
        # which waits for the upload event.  (Round 5 built it here: the pair table never reads it, the camera-movement head
        # does -- found when the pipeline's heads moved onto prefetched scenes.)
        self._pose = None
        self._staged_pose = pose_tables
        return self

    def pose_tables(self):
        """K4's per-frame inputs on the device: (A @ E [F,16], inv(A @ E) [F,16], yaw [F], pitch [F]); the angles are the
        reference's own host arithmetic (engine.extract_yaw_pitch_host), uploaded once per scene."""
        if getattr(self, "_pose", None) is None and getattr(self, "_staged_pose", None) is not None:
            e, yaw, pitch = self._staged_pose
            self._pose = (e, self.cam_mats[:, 0, :].contiguous(), yaw, pitch)
        if getattr(self, "_pose", None) is None:
            F = len(self.ids)
            yaw, pitch = engine.extract_yaw_pitch_host(self.E_aligned)
            E_t = torch.from_numpy(np.stack(self.E_aligned).reshape(F, 16)).to(self.device) if F else \
                torch.zeros((0, 16), dtype=torch.float64, device=self.device)
            self._pose = (E_t, self.cam_mats[:, 0, :].contiguous(), torch.from_numpy(yaw).to(self.device),
                          torch.from_numpy(pitch).to(self.device))
        return self._pose

    # ---- K1 -------------------------------------------------------------------------------
    def vertex_visibility(self, want=("bits", "count")) -> Dict[str, torch.Tensor]:
        if self.xyz is None:
            raise ValueError("scene uploaded without vertices")
        return engine.vertex_visibility(self.xyz, self.cam_mats, self.depth, self.image_hw, want, depth_scale=self.depth_scale)

    def _visibility(self):
        if self._vis is None:
            self._vis = self.vertex_visibility(("bits", "count"))
        return self._vis

    # ---- CFR.process_scene ------------------------------------------------------------------
    def frames_relations_arrays(self) -> Dict[str, np.ndarray]:
        """The pair table in columns: frame indices i < j (key order of CFR:176-178) and overlap / distance / yaw / pitch."""
        F = len(self.ids)
        if F < 2:                                                  # nothing to pair (CFR:176-178 loops over nothing)
            e = np.zeros(0, dtype=np.float64)
            z = np.zeros(0, dtype=np.int32)
            return {"i": z, "j": z.copy(), "overlap": e, "distance": e.copy(), "yaw": e.copy(), "pitch": e.copy()}
        vis = self._visibility()
        pairs = engine.all_pairs(F, self.device)
        overlap = engine.scene_overlap(vis["bits"])               # tiled K2: every pair of the scene in one pass
        pose = engine.pair_pose(*self.pose_tables(), pairs)
        overlap, pose, pairs = overlap.cpu().numpy(), pose.cpu().numpy(), pairs.cpu().numpy()
        return {"i": pairs[:, 0], "j": pairs[:, 1], "overlap": overlap, "distance": pose[:, 0], "yaw": pose[:, 1],
                "pitch": pose[:, 2]}

    def frames_relations(self) -> Dict[Tuple[str, str], Dict[str, float]]:
        """{(id1, id2): {overlap, distance, yaw, pitch}} for all i < j in key order (CFR:176-189)."""
        t = self.frames_relations_arrays()
        # one pass over plain Python lists (tolist) instead of a NumPy scalar per value: 8 064 pairs in ~6 ms instead of 100
        ids = self.ids
        keys = zip([ids[i] for i in t["i"].tolist()], [ids[j] for j in t["j"].tolist()])
        return {k: {"overlap": o, "distance": d, "yaw": y, "pitch": p}
                for k, o, d, y, p in zip(keys, t["overlap"].tolist(), t["distance"].tolist(), t["yaw"].tolist(), t["pitch"].tolist())}

    def empty_frames(self) -> List[str]:
        """Frames that see no vertex at all (the reference logs them, CFR:159-161 / MVI:110-113)."""
        cnt = self._visibility()["count"].cpu().numpy()
        return [k for k, c in zip(self.ids, cnt) if c == 0]

    # ---- MVI.process_scene ------------------------------------------------------------------
    def visibility_csr(self):
        """The visibility index as two CSR tables compacted on the device (mspa/visindex.py): what ``run_split`` streams to
        parquet without ever building a Python list."""
        from . import visindex
        if self.xyz is None:
            raise ValueError("scene uploaded without vertices")
        n = int(self.xyz.shape[0])
        if not self.ids or n == 0:                                  # MVI:103-123 with nothing to loop over
            return visindex.from_bits(None, self.ids, n)
        return visindex.from_bits(self._visibility()["bits"], self.ids, n)

    def visibility_index(self) -> Dict[str, dict]:
        """{"image_to_points": {img: [idx...]}, "point_to_images": {idx: [img...]}} (MVI:103-123): the reference's nested
        dict, built from the CSR tables (image ids per vertex in sorted() order, MVI:117)."""
        return self.visibility_csr().to_dict()

    # ---- correspondence primitives (VC_C:280-344) ---------------------------------------------
    def common_visible_points(self, image_id1: str, image_id2: str) -> np.ndarray:
        """np.intersect1d of the two frames' visible-vertex lists == set bits of (bits1 & bits2)."""
        bits = self._visibility()["bits"]
        both = bits[self.index[image_id1]] & bits[self.index[image_id2]]
        b = both.cpu().numpy().view(np.uint8)
        idx = np.nonzero(np.unpackbits(b, bitorder="little"))[0]
        return idx[idx < self.xyz.shape[0]]

    def point_2d_in_image(self, image_id: str, point_ids: Sequence[int], check_visible: bool = True):
        """get_point_2d_coordinates_in_image (IH:291-305) for a handful of vertices of one image."""
        k = self.index[image_id]
        sel = self.xyz[torch.as_tensor(list(point_ids), device=self.device, dtype=torch.long)].contiguous()
        out = engine.vertex_visibility(sel, self.cam_mats[k:k + 1].contiguous(), self.depth[k:k + 1].contiguous(),
                                       self.image_hw, ("mask", "uv", "depth"), depth_scale=self.depth_scale)
        uv, d, m = out["uv"][0].cpu().numpy(), out["depth"][0].cpu().numpy(), out["mask"][0].cpu().numpy().astype(bool)
        if check_visible:
            return uv[m], d[m]
        return uv, d

    # ---- object-level visibility (compute_object_visibility.process_scene, COVIS:72-152) --------------
    def object_visibility(self, object_point_indices: Dict[int, np.ndarray], min_fraction: float = 0.05):
        """Per (object, image): how many of the object's vertices the image sees -- a masked popcount of
        K1's bitsets instead of Python set intersections on the 13 GB JSON index.
        Returns {"object_to_images": {obj: [{image_id, intersection_count, visibility}, ...]},
                 "image_to_objects": {img: [{object_id, intersection_count, visibility}, ...]}} with the
        reference's threshold max(1, int(0.05 * n_object_points)) and iteration order."""
        bits = self._visibility()["bits"]
        return object_visibility_from_bits(bits, self.ids, self.xyz.shape[0], object_point_indices, min_fraction)

    def object_coverage(self, object_point_indices: Dict[int, np.ndarray], bboxes: Dict[int, Sequence[float]],
                        tolerance: float = 0.1, min_fraction: float = 0.05, rng=None, points_dtype=np.float64):
        """COVIS.process_scene + COV.process_scene_for_coverage for the resident scene: object visibility
        (masked popcount, K2) and per-(object, image) extents (K8) straight from K1's bitsets, then the
        minimal-combination search per object and axis.  ``bboxes[o]`` = (cx, cy, cz, dx, dy, dz) of the
        axis-aligned box.  Returns ({obj: {"height"|"length"|"width": {k: [combinations]}}}, visibility)."""
        import random as _random
        from . import coverage
        rng = rng or _random
        visibility = self.object_visibility(object_point_indices, min_fraction)
        per_object = visibility["object_to_images"]
        out = {}
        if not per_object:
            return out, visibility
        objects = {o: np.asarray(object_point_indices[o]) for o in per_object}
        ext = coverage.scene_extents(self._visibility()["bits"], self.ids, self.xyz, objects, dtype=points_dtype)
        for o, entries in per_object.items():
            b = bboxes[o]
            width_axis = 0 if b[3] < b[4] else 1                         # IH:224-230
            out[o] = coverage.object_coverage(ext, o, [e["image_id"] for e in entries], b[5], max(b[3], b[4]),
                                              min(b[3], b[4]), width_axis, tolerance, rng)
        return out, visibility

    # ---- K3 ---------------------------------------------------------------------------------
    def _require_millimetres(self, what: str):
        """K3 folds the reference's 0.001 m per depth unit (OPS:292-294, a literal there) into its matrices; a handler built
        with another ``depth_value_scale`` (IH:76) would get vertex visibility at one scale and pair visibility at another."""
        if self.depth_scale != 0.001:
            raise ValueError(f"SceneOnDevice.{what}: the frame-pair kernels work in millimetres (project_mask_to_3d's literal "
                             f"0.001, OPS:292-294); this scene was built with depth_scale={self.depth_scale!r}")

    def pair_reproject(self, pairs_ids: Sequence[Tuple[str, str]], outputs: Sequence[str], fast: bool = True):
        self._require_millimetres("pair_reproject")
        pairs = torch.tensor([[self.index[a], self.index[b]] for a, b in pairs_ids], dtype=torch.int32,
                             device=self.device).reshape(-1, 2)
        out = engine.alloc_pair_outputs(pairs.shape[0], self.image_hw, outputs, self.device)
        flags = engine._lib.PAIR_FAST if (fast and engine.fast_path_ok(self.K)) else 0
        engine.pair_reproject(self.depth, self.frame_mats, pairs, self.image_hw, out, rgb=self.rgb, flags=flags)
        return out

    def pair_correspondences(self, pairs_ids: Sequence[Tuple[str, str]], fast: bool = True):
        """K3 with the compacted output (include/mspa.h, mspa_pair_correspondences): per pair the visibility bitset and, per
        64 x 48 tile of the first image, the second image's depth pixel (xi, yi) of the VISIBLE pixels only.
        ``engine.correspondences_rowmajor(out, self.image_hw, k)`` gives pair k's flat (pixel, xi, yi) in np.nonzero order."""
        self._require_millimetres("pair_correspondences")
        pairs = torch.tensor([[self.index[a], self.index[b]] for a, b in pairs_ids], dtype=torch.int32,
                             device=self.device).reshape(-1, 2)
        out = engine.alloc_pair_correspondences(pairs.shape[0], self.image_hw, self.device)
        flags = engine._lib.PAIR_FAST if (fast and engine.fast_path_ok(self.K)) else 0
        engine.pair_correspondences(self.depth, self.frame_mats, pairs, self.image_hw, out, flags=flags)
        return out


def pack_index_lists(index_lists: Sequence[Sequence[int]], n_points: int) -> np.ndarray:
    """Vertex-index lists -> [len, ceil(n_points/64)] int64 bitset rows (bit i of word w = vertex 64w+i)."""
    n_words = (n_points + 63) // 64
    out = np.zeros((len(index_lists), n_words * 64), dtype=bool)
    for r, idx in enumerate(index_lists):
        out[r, np.asarray(idx, dtype=np.int64)] = True
    return np.packbits(out, axis=1, bitorder="little").view(np.int64)


def object_visibility_from_bits(image_bits: torch.Tensor, image_ids: Sequence[str], n_points: int,
                                object_point_indices: Dict[int, np.ndarray], min_fraction: float = 0.05):
    objs = [(o, np.unique(np.asarray(p, dtype=np.int64))) for o, p in object_point_indices.items() if len(p) > 0]
    result = {"object_to_images": {}, "image_to_objects": {}}
    if not objs or len(image_ids) == 0:
        return result
    dev = image_bits.device
    obj_bits = torch.from_numpy(pack_index_lists([p for _, p in objs], n_points)).to(dev)
    F, O = image_bits.shape[0], len(objs)
    inter = engine.overlap_matrix(obj_bits, image_bits.contiguous()).cpu().numpy()       # [O, F] masked popcounts
    for k, (obj, pts) in enumerate(objs):                     # object-major, then image order: as upstream
        total = len(pts)
        threshold = max(1, int(min_fraction * total))
        for f, image_id in enumerate(image_ids):
            c = int(inter[k, f])
            if c >= threshold:
                vis = (c / total) * 100.0
                result["object_to_images"].setdefault(obj, []).append(
                    {"image_id": image_id, "intersection_count": c, "visibility": vis})
                result["image_to_objects"].setdefault(image_id, []).append(
                    {"object_id": obj, "intersection_count": c, "visibility": vis})
    return result
