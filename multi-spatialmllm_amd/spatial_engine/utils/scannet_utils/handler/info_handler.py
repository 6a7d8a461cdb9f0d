"""Mirror of the reference's handler/info_handler.py: same names, arguments, defaults and return
types; the geometry runs on the MI355X through libmspa.so.

Scene-info layout (unchanged): ``infos[scene_id]`` holds ``intrinsic_matrix`` (4x4),
``axis_align_matrix`` (4x4), ``num_posed_images``, ``num_objects``, per-object boxes under integer
keys and ``images_info[image_id]["extrinsic_matrix"]`` (4x4 camera->world, may contain -inf).
"""
from __future__ import annotations

import json
import os
import pickle

import numpy as np
import torch

from mspa import engine

from . import _images
from .ops import project_mask_to_3d


def _load_any(path):
    try:
        import mmengine
        return mmengine.load(path)
    except ImportError:
        with open(path, "rb") as f:
            return pickle.load(f)


def _dev(a, dtype=np.float64):
    """NumPy / list input -> device tensor; a torch tensor stays a tensor (moved to the GPU / float64 if need be)."""
    if isinstance(a, torch.Tensor):
        return a.to(device="cuda", dtype=torch.float64 if dtype == np.float64 else None).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype))).cuda()


def _like(result: torch.Tensor, *inputs, as_bool=False):
    """Results go back the way the inputs came: torch tensors in -> device tensors out (callers can stay on the GPU),
    NumPy in -> NumPy out (the reference's return types)."""
    if as_bool:
        result = result.to(torch.bool)
    if any(isinstance(x, torch.Tensor) for x in inputs):
        return result
    return result.cpu().numpy()


def project_points(points, K, E):
    """[N,4] homogeneous world points, 4x4 K, 4x4 camera->world E -> ([N,2] un-rounded pixel
    coordinates, [N] signed camera depth), float64 (reference: info_handler.py:46-72).
    Any homogeneous coordinate is taken as it comes: E_inv @ points.T and K @ (.) are evaluated as the full 4x4 products the
    reference forms (w = 1, the only value its own call sites pass, gives the same bits as the affine kernels)."""
    given = points
    if not isinstance(points, torch.Tensor):
        points = np.asarray(points, dtype=np.float64)
    if points.ndim != 2 or points.shape[1] != 4:
        raise ValueError("points must be [N, 4] homogeneous coordinates")
    K = K.cpu().numpy() if isinstance(K, torch.Tensor) else K
    E = E.cpu().numpy() if isinstance(E, torch.Tensor) else E
    cam = torch.from_numpy(engine.camera_matrices(np.asarray(K, np.float64), [np.asarray(E, np.float64)])).cuda()
    xyzw = _dev(points)
    dummy = torch.zeros((1, 2, 2), dtype=torch.int16, device="cuda")
    out = engine.vertex_visibility(xyzw, cam, dummy, (2, 2), ("uv", "depth"), homogeneous=True)
    return _like(out["uv"][0], given), _like(out["depth"][0], given)


class SceneInfoHandler:
    def __init__(self, info_path, posed_images_root="data/scannet/posed_images",
                 instance_data_root="data/scannet/scannet_instance_data", mask_image_root="data/scannet/scans",
                 depth_value_scale=0.001):
        if isinstance(info_path, dict):
            self.infos = info_path                       # convenience: an already-loaded infos dict
        else:
            try:
                self.infos = _load_any(info_path)
                print(f"Data from {info_path} loaded successfully.")
            except Exception as e:                       # same observable behaviour as upstream (IH:80-82)
                print(f"Failed to load data from {info_path}: {e}")
                raise SystemExit(1)
        if not (isinstance(depth_value_scale, (int, float)) and 0.0 < float(depth_value_scale) < float("inf")):
            raise ValueError("depth_value_scale must be a positive finite number (metres per depth-image unit)")
        self.posed_images_root = posed_images_root
        self.instance_data_root = instance_data_root
        self.mask_image_root = mask_image_root
        self.depth_value_scale = depth_value_scale

    # ---- plain accessors ----------------------------------------------------------------------
    def __len__(self):
        return len(self.infos)

    def get_sorted_keys(self):
        return sorted(self.infos.keys())

    def get_all_scene_ids(self):
        return list(self.infos.keys())

    def get_intrinsic_matrix(self, scene_id, image_id=None):
        return self.infos[scene_id]["intrinsic_matrix"]

    def convert_image_id_to_key(self, image_id):
        try:
            image_id = int(image_id)
        except Exception as e:
            print(f"Failed to convert image_id: {image_id}: {e}")
            return None
        return None if image_id < 0 else f"{image_id:05d}"

    def get_extrinsic_matrix(self, scene_id, image_id, warning=True):
        key = self.convert_image_id_to_key(image_id)
        E = self.infos[scene_id]["images_info"][key]["extrinsic_matrix"]
        if warning and not np.all(np.isfinite(E)):
            print(f"[SceneInfoHanlder] Warning: extrinsics matrix of {scene_id}: {key} contains inf or nan.")
        return E

    def get_world_to_axis_align_matrix(self, scene_id, image_id=None):
        return self.infos[scene_id]["axis_align_matrix"]

    def get_extrinsic_matrix_align(self, scene_id, image_id):
        return self.get_world_to_axis_align_matrix(scene_id) @ self.get_extrinsic_matrix(scene_id, image_id)

    def get_num_posed_images(self, scene_id):
        return self.infos[scene_id]["num_posed_images"]

    def get_num_objects(self, scene_id):
        return self.infos[scene_id]["num_objects"]

    def get_all_image_ids(self, scene_id):
        return list(self.infos[scene_id]["images_info"].keys())

    def is_posed_image_valid(self, scene_id, image_id):
        key = self.convert_image_id_to_key(image_id)
        if key is None:
            return False
        return bool(np.all(np.isfinite(self.get_extrinsic_matrix(scene_id, key, warning=False))))

    def get_all_extrinsic_valid_image_ids(self, scene_id):
        # one isfinite over the scene's stacked poses (0.15 ms for 320 frames) instead of one call per image (0.94 ms): every
        # rank asks this of EVERY scene when it prices the split, and again of each scene it owns
        from mspa.scene import valid_image_ids
        images = self.infos[scene_id]["images_info"]
        if all(isinstance(k, str) and len(k) == 5 and k.isdigit() for k in images):      # canonical keys ("%05d": their own key)
            return valid_image_ids({k: v["extrinsic_matrix"] for k, v in images.items()})
        return [i for i in images if self.is_posed_image_valid(scene_id, i)]              # anything else: image by image, as upstream

    def get_image_path(self, scene_id, image_id):
        key = self.convert_image_id_to_key(image_id)
        return None if key is None else os.path.join(self.posed_images_root, scene_id, f"{key}.jpg")

    def get_depth_image_path(self, scene_id, image_id):
        key = self.convert_image_id_to_key(image_id)
        return None if key is None else os.path.join(self.posed_images_root, scene_id, f"{key}.png")

    def get_image_shape(self, scene_id, image_id=None):
        if image_id is None:
            image_id = self.get_all_image_ids(scene_id)[0]
        return tuple(_images.image_shape(self.get_image_path(scene_id, image_id)))     # (H, W)

    get_image_size = get_image_shape   # the name visual_correspondence calls (upstream defect)

    def get_depth_image(self, scene_id, image_id):
        return _images.read_depth(self.get_depth_image_path(scene_id, image_id))

    def get_depth_image_shape(self, scene_id, image_id=0):
        return self.get_depth_image(scene_id, image_id).shape[:2]

    # ---- objects / point clouds ---------------------------------------------------------------
    def get_object_gt_bbox(self, scene_id, object_id, axis_aligned=True, with_class_id=False):
        bbox = self.infos[scene_id][object_id]["aligned_bbox" if axis_aligned else "unaligned_bbox"]
        return bbox if with_class_id else bbox[0:-1]

    def get_object_raw_category(self, scene_id, object_id):
        return self.infos[scene_id][object_id]["raw_category"]

    def get_scene_raw_categories(self, scene_id):
        return [self.get_object_raw_category(scene_id, o) for o in range(self.get_num_objects(scene_id))]

    def get_object_height(self, scene_id, object_id):
        return self.get_object_gt_bbox(scene_id, object_id)[5]

    def get_object_length(self, scene_id, object_id):
        b = self.get_object_gt_bbox(scene_id, object_id)
        return max(b[3], b[4])

    def get_object_width(self, scene_id, object_id):
        b = self.get_object_gt_bbox(scene_id, object_id)
        return min(b[3], b[4])

    def get_object_length_axis_aligned(self, scene_id, object_id):
        b = self.get_object_gt_bbox(scene_id, object_id)
        return 0 if b[3] > b[4] else 1

    def get_object_width_axis_aligned(self, scene_id, object_id):
        b = self.get_object_gt_bbox(scene_id, object_id)
        return 0 if b[3] < b[4] else 1

    def get_object_volume(self, scene_id, object_id):
        b = self.get_object_gt_bbox(scene_id, object_id)
        return b[3] * b[4] * b[5]

    def get_scene_points_align(self, scene_id):
        return np.load(os.path.join(self.instance_data_root, scene_id, "aligned_points.npy"))

    def get_scene_points(self, scene_id):
        return np.load(os.path.join(self.instance_data_root, scene_id, "unaligned_points.npy"))

    def get_scene_instance_mask(self, scene_id):
        return np.load(os.path.join(self.instance_data_root, scene_id, "instance_mask.npy"))

    def get_object_points_aligned(self, scene_id, object_id):
        return np.load(os.path.join(self.instance_data_root, scene_id, f"object_{object_id}_aligned_points.npy"),
                       allow_pickle=True)

    def get_object_point_index(self, scene_id, object_id):
        idx = np.where(self.get_scene_instance_mask(scene_id) == object_id + 1)[0]
        if len(idx) == 0:
            print(f"[SceneInfoHanlder] Warning: {scene_id} does not have object {object_id}.")
        return idx

    def get_point_3d_coordinates(self, scene_id, point_id, align=True):
        pts = self.get_scene_points_align(scene_id) if align else self.get_scene_points(scene_id)
        return pts[point_id]

    # ---- geometry: HIP -------------------------------------------------------------------------
    def project_3d_point_to_image(self, scene_id, image_id, points_3d, align=True):
        """[N,3] or (3,) world points -> ([N,2] pixel coordinates, [N] depth)  (IH:313-335)."""
        K = self.get_intrinsic_matrix(scene_id, image_id)
        E = self.get_extrinsic_matrix_align(scene_id, image_id) if align else self.get_extrinsic_matrix(scene_id, image_id)
        if isinstance(points_3d, torch.Tensor):
            pts = points_3d.to(device="cuda", dtype=torch.float64)
            pts = pts[None, :] if pts.ndim == 1 else pts
            return project_points(torch.cat([pts[:, :3], torch.ones((pts.shape[0], 1), dtype=torch.float64, device="cuda")], 1), K, E)
        pts = np.asarray(points_3d, dtype=np.float64)
        pts = pts[None, :] if pts.ndim == 1 else pts
        return project_points(np.hstack([pts[:, :3], np.ones((pts.shape[0], 1))]), K, E)

    def check_point_in_image_boundary(self, scene_id, points_2d):
        out = engine.check_visibility(_dev(points_2d), None, None, self.get_image_shape(scene_id), ("in_bounds",))
        return _like(out["in_bounds"], points_2d, as_bool=True)

    def _depth_dev(self, scene_id, image_id):
        return engine.depth_to_device(self.get_depth_image(scene_id, image_id), "cuda")

    def check_point_visibility_by_depth(self, scene_id, image_id, points_2d, points_depth):
        out = engine.check_visibility(_dev(points_2d), _dev(points_depth), self._depth_dev(scene_id, image_id),
                                      self.get_image_shape(scene_id, image_id), ("by_depth",), depth_scale=self.depth_value_scale)
        return _like(out["by_depth"], points_2d, points_depth, as_bool=True)

    def check_point_visibility(self, scene_id, image_id, points_2d, points_depth):
        out = engine.check_visibility(_dev(points_2d), _dev(points_depth), self._depth_dev(scene_id, image_id),
                                      self.get_image_shape(scene_id), ("visible",), depth_scale=self.depth_value_scale)
        return _like(out["visible"], points_2d, points_depth, as_bool=True)

    def get_point_2d_coordinates_in_image(self, scene_id, image_id, point_id, align=True, check_visible=False,
                                          return_depth=False):
        point_3d = self.get_point_3d_coordinates(scene_id, point_id, align)[:3]
        uv, d = self.project_3d_point_to_image(scene_id, image_id, point_3d, align)
        if check_visible:
            m = self.check_point_visibility(scene_id, image_id, uv, d)
            uv, d = uv[m], d[m]
        return (uv, d) if return_depth else uv

    def project_image_to_3d_with_mask(self, scene_id, image_id, mask=None, with_color=False):
        color = self.get_image_path(scene_id, image_id) if with_color else None
        return project_mask_to_3d(self.get_depth_image_path(scene_id, image_id),
                                  self.get_intrinsic_matrix(scene_id, image_id),
                                  self.get_extrinsic_matrix(scene_id, image_id), mask,
                                  self.get_world_to_axis_align_matrix(scene_id), color_image=color)

    def get_instance_mask(self, scene_id, image_id, target_id) -> np.ndarray:
        path = os.path.join(self.mask_image_root, scene_id, "instance-filt", f"{int(image_id)}.png")
        try:
            mask_image = _images.read_depth(path)
        except Exception:
            mask_image = None
        if mask_image is None:
            raise FileNotFoundError(f"Mask image not found at path: {path}")
        return np.where(mask_image == target_id + 1, 1, 0)

    # ---- resident scene (what the per-scene scripts use) -----------------------------------------
    def host_scene(self, scene_id, num_workers=8, with_points=True, decode=None, prepare=False):
        """The scene in host memory (``mspa.sweep.HostScene``): poses, the axis-aligned vertices and the depth frames of the
        frames with a finite pose (one ``cv2.imread`` per frame upstream, IH:149-155).

        ``decode="device"``: the frames stay COMPRESSED -- ``num_workers`` native threads read the PNG files and pack their
        scanline zlib streams into one page-locked buffer (``mspa.ingest.pack_scene_depth``); the upload stage copies that
        (half the bytes of the decoded frames) and the MI355X inflates and un-filters them (csrc/device_ingest.hip).
        ``decode="host"``: read and inflated here by the native threads (``mspa.ingest.read_depth_frames``) into one
        [F, h, w] block.  Default: the environment's ``MSPA_DEPTH_DECODE``, else "host" (the streaming sweeps ask for "device").
        Frames the device path cannot take (registered in memory, another pixel format) make the whole scene use the host path.
        ``prepare``: the tables the upload stage derives from the poses and vertices (``mspa.upload.prepare_tables``) are computed
        here, on the caller's (a loader's) thread, and travel with the scene (measured slower end to end: off by default)."""
        from mspa import ingest
        from mspa.scene import valid_image_ids
        from mspa.sweep import HostScene
        ids = self.get_all_image_ids(scene_id)
        E = {i: self.infos[scene_id]["images_info"][i]["extrinsic_matrix"] for i in ids}
        valid = valid_image_ids(E)
        paths = [self.get_depth_image_path(scene_id, i) for i in valid]
        decode = decode or os.environ.get("MSPA_DEPTH_DECODE", "host")
        packed = None
        if decode == "device" and paths and not any(q in _images.MEMORY for q in paths):
            packed = ingest.pack_scene_depth(paths, num_workers)
        pts = self.get_scene_points_align(scene_id)[:, :3] if with_points else None
        prepared = None
        if prepare:
            from mspa import upload
            prepared = upload.prepare_tables(self.get_intrinsic_matrix(scene_id), self.get_world_to_axis_align_matrix(scene_id), E, pts,
                                             valid)
        if packed is not None:
            return HostScene(scene_id, self.get_intrinsic_matrix(scene_id), self.get_world_to_axis_align_matrix(scene_id), E, {},
                             tuple(self.get_image_shape(scene_id)), pts, float(self.depth_value_scale), packed=packed,
                             depth_ids=valid, prepared=prepared)
        pool = ingest.DEFAULT_POOL                        # reused destinations: no page faults under the decode threads
        block = ingest.read_depth_frames(paths, num_workers, general_reader=_images.read_depth, memory=_images.MEMORY, pool=pool)
        hs = HostScene(scene_id, self.get_intrinsic_matrix(scene_id), self.get_world_to_axis_align_matrix(scene_id), E,
                       {i: block[k] for k, i in enumerate(valid)}, tuple(self.get_image_shape(scene_id)), pts,
                       float(self.depth_value_scale), prepared=prepared)
        if len(valid):
            import weakref
            weakref.finalize(hs, pool.give, block)       # the block goes back when nothing holds the scene any more
        return hs

    def scene_on_device(self, scene_id, with_points=True, num_workers=8):
        from mspa.scene import SceneOnDevice
        hs = self.host_scene(scene_id, num_workers, with_points, decode="host")
        return SceneOnDevice(hs.K, hs.A, hs.E, hs.depth, hs.color_hw, hs.points, depth_scale=self.depth_value_scale)

    def scene_cost(self, scene_id):
        """Estimated cost of a scene for the longest-first assignment (F^2 N / 64 + F N) without loading it:
        F from the poses, N from the header of the vertex file."""
        from mspa import shard
        F = len(self.get_all_extrinsic_valid_image_ids(scene_id))
        try:
            with open(os.path.join(self.instance_data_root, scene_id, "aligned_points.npy"), "rb") as f:
                version = np.lib.format.read_magic(f)
                shape = (np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0)(f)[0]
            N = int(shape[0])
        except Exception:
            N = 1
        return shard.scene_cost(F, N)

    def scene_costs(self, scene_ids, world=1):
        """``scene_cost`` of every scene of a split, as the windows of a sharded sweep want them.  One rank deals nothing: every
        window's scenes are its own whatever they cost, and pricing 1 500 scenes is 0.3 s that the first kernel would wait for."""
        if world <= 1:
            return [1.0] * len(scene_ids)
        return [self.scene_cost(s) for s in scene_ids]

    def prefetched_scenes(self, scene_ids, num_workers=8, device="cuda", timings=None, with_points=True, lookahead=None,
                          decode=None):
        """``scene_ids`` -> resident scenes, one after the other: scene n+1 is decoded by ``num_workers`` host threads and
        copied on the copy stream while the caller runs scene n's kernels (mspa/sweep.py, mspa/upload.py).  ``lookahead`` scenes
        are in flight on the host at once (default: 2, more -- up to 4 -- when the process has CPUs to spare for them: its
        affinity mask and cgroup quota count, not the machine's ``os.cpu_count()``, mspa/hostinfo.py)."""
        from mspa import hostinfo, sweep
        if lookahead is None and os.environ.get("MSPA_LOOKAHEAD"):
            lookahead = max(1, int(os.environ["MSPA_LOOKAHEAD"]))
        if lookahead is None:
            lookahead = min(4, max(2, hostinfo.effective_cpus() // (2 * max(1, int(num_workers)))))
        # The streaming sweeps decode the depth frames on the MI355X (MSPA_DEPTH_DECODE=host / device overrides) -- when there
        # are enough of them: one wave inflates one frame and takes ~33 ms for it however idle the chip is (inflate v6; 70 ms with
        # v4, when the threshold was 1 536 frames).  Measured on one box, device against 16 host CPUs, pair-table sweep: 8 scenes x
        # 64 frames 125 against 112 scenes/s, 16 x 64 137 against 123, 4 x 320 51 against 25, 48 x 320 110 against 31
        # (profiles/r06_dropin_decode.md).  At 512 frames the two are level (and with 16 loader threads the host is ahead: 126
        # against 112 scenes/s in bench.py's small leg), so the device takes over from 1 024 frames on.
        scene_ids = list(scene_ids)
        decode = decode or os.environ.get("MSPA_DEPTH_DECODE")
        if decode is None:
            frames = 0
            for sid in scene_ids:                          # (stops at the threshold: not one pass over a 1 500-scene split)
                frames += len(self.get_all_extrinsic_valid_image_ids(sid))
                if frames >= 1024:
                    break
            decode = "device" if frames >= 1024 else "host"
        loader = sweep.SceneLoader(lambda sid: self.host_scene(sid, num_workers, with_points, decode, prepare=os.environ.get("MSPA_PREPARE_ON_LOADER", "0") == "1"), list(scene_ids),
                                   lookahead, timings)
        return sweep.prefetched_scenes(loader, device, timings, decode_on_device=(decode == "device"))


class VisibilityInfoHandler:
    """Reader of the vertex<->image visibility index (parquet with JSON-string values, or nested pkl)."""

    def __init__(self, visibility_info_path):
        self.visibility_info_path = visibility_info_path
        print(f"[VisibilityInfoHandler] Reading visibility info from {self.visibility_info_path}.")
        if visibility_info_path.endswith(".parquet"):
            import pandas as pd
            self.info_format = "parquet"
            print("[VisibilityInfoHandler] Converting parquet file to dict.")
            self.visibility_info = self.convert_parquet_to_dict(pd.read_parquet(visibility_info_path))
        elif visibility_info_path.endswith(".pkl"):
            self.info_format = "pkl"
            self.visibility_info = _load_any(visibility_info_path)
        else:
            raise ValueError(f"Unsupported file format: {self.visibility_info_path}")

    def convert_parquet_to_dict(self, parquet_df):
        """The (key, values) frame back as {key: JSON text} (reference: IH:486-500)."""
        return dict(zip(parquet_df["key"].tolist(), parquet_df["values"].tolist()))

    def _get(self, scene_id, kind, item):
        if self.info_format == "parquet":
            key = f"{scene_id}:{kind}:{item}"
            if key not in self.visibility_info:
                raise ValueError(f"Key {key} not found in visibility info.")
            return json.loads(self.visibility_info[key])
        if scene_id not in self.visibility_info:
            raise ValueError(f"Scene {scene_id} not found in visibility info.")
        if item not in self.visibility_info[scene_id][kind]:
            what = "Image" if kind == "image_to_points" else "Point"
            raise ValueError(f"{what} {item} not found in visibility info for scene {scene_id}.")
        return self.visibility_info[scene_id][kind][item]

    def get_image_to_points_info(self, scene_id, image_id):
        return self._get(scene_id, "image_to_points", image_id)

    def get_point_to_images_info(self, scene_id, point_index):
        return self._get(scene_id, "point_to_images", point_index)
