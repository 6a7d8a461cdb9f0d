"""Mirror of the reference's single_object_movement_engine_coord.py geometry entry points."""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from mspa import engine, heads
from mspa import templates as T
from mspa.hostinfo import quietly


def rigid_body_segmentation(points, threshold=0.1, smoothing_factor=0.01):
    """Groups of track indices that move rigidly together (reference: :49-92): the T x P^2 distance-change
    accumulation runs on the GPU (K7), average linkage + fcluster stay with SciPy."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import squareform
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)).cuda()
    loss = engine.track_rigidity_loss(pts, smoothing_factor).cpu().numpy()
    links = linkage(squareform(loss), method="average")
    labels = fcluster(links, threshold, criterion="distance")
    return [np.where(labels == i)[0].tolist() for i in range(1, max(labels) + 1)]


class _LoadedSample(dict):
    """A TAPVid-3D sample file read into memory by a loader thread (same access pattern as the lazy ``np.load`` object)."""

    @property
    def files(self):
        return list(self.keys())


def load_tapvid_sample(input_file):
    with np.load(input_file, allow_pickle=True) as gt:
        return _LoadedSample({k: gt[k] for k in gt.files})


def sharded_scenes(files, run_one, num_workers=20, ctx=None):
    """Records of every sample file, in file order, on rank 0 -- the reference's ``pool.map`` over scenes (OM_C:584-585).

    One process per GPU (``RANK`` / ``WORLD_SIZE`` from the environment, or ``ctx``): files dealt longest-first by size,
    ``num_workers`` (capped at 4) loader threads read the next files while the current one is on the GPU, and every rank's
    finished records travel to rank 0 as JSON-lines bytes (``shard.gather_bytes``): text is built where the numbers are.
    ``run_one(path, loaded)`` returns one file's records; other ranks get []."""
    import json
    from mspa import shard, sweep
    if ctx is None:
        ctx = shard.context_from_env()
    costs = [float(os.path.getsize(f)) if os.path.exists(f) else 1.0 for f in files]
    data, direct = [], {}

    def work_items(indices):
        return sweep.SceneLoader(lambda i: load_tapvid_sample(files[i]), indices, lookahead=max(1, min(int(num_workers), 4)))

    def produce(index, loaded):
        records = run_one(files[index], loaded)
        if ctx is None:                                      # nothing to exchange: the records themselves are handed on
            direct[index] = records
            return None, []
        return None, ["".join(json.dumps(r) + "\n" for r in records).encode()]

    def consume(index, _rows, blobs):
        if ctx is None:
            data.extend(direct.pop(index))
        else:
            data.extend(json.loads(line) for line in bytes(blobs[0]).splitlines())

    sweep.sharded_sweep(costs, ctx, work_items, produce, consume, per_rank=4)
    return data, ctx


def filter_large_groups(groups, min_size=5):
    return [g for g in groups if len(g) > min_size]


class TwoFrameVideoQAEngine:
    def __init__(self, question_type, sub_dataset):
        self.question_type = question_type
        self.sub_dataset = sub_dataset
        self.templates = T.OBJECT_MOVEMENT
        self.object_not_moving_threshold = 0.01
        self.camera_not_moving_threshold = 0.01
        self.future_frame_windows = 1e8

    def project_point(self, point_3d, intrinsics, image_height, image_width, id=""):
        """Normalised pinhole projection of one camera-space point, None when it falls outside the image
        or behind the camera (reference: :293-315); K5a with T = P = 1."""
        tr = torch.tensor(np.asarray(point_3d, dtype=np.float64).reshape(1, 1, 3), device="cuda")
        res = engine.track_to_world(tr, None, intrinsics, (int(image_height), int(image_width)), ("uvn", "ok"))
        if not bool(res["ok"][0, 0]):
            print(f"point {np.asarray(point_3d).tolist()} is invalid for intrinsics {list(intrinsics)}.")
            return None
        return res["uvn"][0, 0].cpu().numpy().tolist()

    def format_training_samples(self, sample_pairs, intrinsics, scene_id, points_pos_world, points_pos_cam,
                                image_height, image_width, extrinsics_w2c):
        """Records for chosen {frame1, frame2, point_index} samples (reference: :317-404); K5a + K5b.
        ``points_pos_world`` is accepted for signature compatibility -- the kernel recomputes it."""
        return heads.object_movement_records(scene_id, np.asarray(points_pos_cam), np.asarray(extrinsics_w2c), intrinsics,
                                             (int(image_height), int(image_width)), sample_pairs, self.question_type,
                                             self.templates, random)

    # -- scene level (reference: :405-575) ---------------------------------------------------------
    def generate_qa_training_single_scene(self, input_file, npoints_per_group=5, npairs_per_bin=1e8, img_output_dir="",
                                          augment=True, augment_ratio=1.0, _loaded=None):
        """All records of one TAPVid-3D sample file: frames to ``img_output_dir/<scene>/``, rigid groups (K7 + SciPy),
        frame-pair mining (K5c), records (K5a + K5b).  The JPEG payloads are written as stored -- upstream decodes
        and re-encodes them -- and the image size is read from the first payload's header."""
        scene_id = os.path.splitext(os.path.basename(input_file))[0]
        gt = _loaded if _loaded is not None else np.load(input_file, allow_pickle=True)
        scene_img_dir = os.path.join(img_output_dir, scene_id)
        os.makedirs(scene_img_dir, exist_ok=True)
        payloads = gt["images_jpeg_bytes"]
        have = [n for n in os.listdir(scene_img_dir) if n.endswith(".jpg")]
        if len(have) != payloads.shape[0]:
            print(f"Saving images for {scene_id}. Total images {payloads.shape[0]}.")
            for i, frame_bytes in enumerate(payloads):
                with open(os.path.join(scene_img_dir, f"{i:05d}.jpg"), "wb") as fh:
                    fh.write(bytes(frame_bytes))
        image_height, image_width = jpeg_size(bytes(payloads[0]))
        intrinsics = gt["fx_fy_cx_cy"]
        tracks_xyz = np.ascontiguousarray(gt["tracks_XYZ"], dtype=np.float64)
        visibility = gt["visibility"]
        extrinsics_w2c = gt["extrinsics_w2c"] if "extrinsics_w2c" in gt.files else None
        n_frames = tracks_xyz.shape[0]
        tracks_dev = torch.from_numpy(tracks_xyz).cuda()
        if extrinsics_w2c is not None:
            c2w = torch.from_numpy(np.linalg.inv(extrinsics_w2c).reshape(n_frames, 16)).cuda()
            world = engine.track_to_world(tracks_dev, c2w, intrinsics, (image_height, image_width), ("world",))["world"]
        else:
            world = tracks_dev
            extrinsics_w2c = np.array([np.eye(4) for _ in range(n_frames)])
        groups = filter_large_groups(rigid_body_segmentation(tracks_xyz), min_size=5)
        sample_pairs = heads.object_movement_mine_pairs(
            visibility, groups, lambda pts, frames: engine.track_pair_distances(world, pts, frames), npoints_per_group,
            npairs_per_bin, augment, augment_ratio, random, self.object_not_moving_threshold, self.future_frame_windows)
        return self.format_training_samples(sample_pairs, intrinsics=intrinsics, scene_id=scene_id,
                                            points_pos_world=None, points_pos_cam=tracks_xyz, image_height=image_height,
                                            image_width=image_width, extrinsics_w2c=extrinsics_w2c)

    @quietly
    def _all_scenes(self, scene_id_list, source_data_root, img_output_dir, npoints_per_group, npairs_per_bin, augment,
                    augment_ratio, num_workers=20, ctx=None):
        """Upstream maps the scenes over a fork pool (:584-585), so every scene starts from a copy of the parent's ``random``
        state and the parent's own stream is untouched by them; reproduced here scene by scene, the scenes sharded over the
        job's GPUs (``sharded_scenes``).  Returns the records in scene order on rank 0, [] on the other ranks."""
        parent = random.getstate()

        def run_one(path, loaded):
            random.setstate(parent)
            return self.generate_qa_training_single_scene(path, npoints_per_group, npairs_per_bin, img_output_dir, augment,
                                                          augment_ratio, _loaded=loaded)
        data, self._ctx = sharded_scenes([os.path.join(source_data_root, f"{scene_id}.npz") for scene_id in scene_id_list],
                                         run_one, num_workers, ctx)
        random.setstate(parent)
        return data

    def _is_writer(self):
        """Rank 0 writes the files (every rank of a multi-GPU job runs the same script)."""
        ctx = getattr(self, "_ctx", None)
        if ctx is not None:
            ctx.barrier()
        return ctx is None or ctx.rank == 0

    def _sync_generator(self):
        """Every rank leaves a dataset call with rank 0's ``random`` state.  Only the writer runs ``random.sample`` /
        ``random.shuffle`` on the collected records, and the NEXT call's scenes start from ``random.getstate()`` on whichever
        rank owns them (``_all_scenes``): without this the second call of a job (main() makes eight) would depend on the
        world size.  One pickled generator state (2.5 KB) per call."""
        from mspa import shard
        ctx = getattr(self, "_ctx", None)
        if ctx is not None:
            random.setstate(shard.broadcast_object(random.getstate(), ctx, src=0))

    @staticmethod
    def _report(kind, output_file, data):
        still = sum(1 for e in data if e["point_moving"] == 0)
        cam_still = sum(1 for e in data if e["cam_moving"] == 0)
        print(f"{kind} data saved to {output_file}. In total, there are {len(data)} samples.")
        print(f"Object not moving: {still}, Object moving: {len(data) - still}")
        print(f"Camera not moving: {cam_still}, Camera moving: {len(data) - cam_still}")

    def generate_qa_training_data(self, scene_id_list, source_data_root, output_dir, output_file, img_output_dir,
                                  npoints_per_group, npairs_per_bin, augment, augment_ratio=1.0, max_samples=-1, num_workers=20):
        data = self._all_scenes(scene_id_list, source_data_root, img_output_dir, npoints_per_group, npairs_per_bin, augment,
                                augment_ratio, num_workers)
        if self._is_writer():
            if max_samples > 0 and len(data) > max_samples:
                data = random.sample(data, max_samples)
            random.shuffle(data)
            heads.write_jsonl(output_file, data)
            self._report("Training", output_file, data)
        self._sync_generator()

    def format_eval_sample(self, training_sample):
        training_sample["text"] = training_sample["conversations"][0]["value"]
        return training_sample

    def generate_qa_eval_data(self, scene_id_list, source_data_root, output_dir, output_file, img_output_dir,
                              npoints_per_group, npairs_per_bin, augment, augment_ratio=0.3, max_samples=300, num_workers=20):
        data = self._all_scenes(scene_id_list, source_data_root, img_output_dir, npoints_per_group, npairs_per_bin, augment,
                                augment_ratio, num_workers)
        if self._is_writer():
            if max_samples > 0 and len(data) > max_samples:
                data = random.sample(data, max_samples)
            heads.write_jsonl(output_file, [self.format_eval_sample(s) for s in data])
            self._report("Evaluation", output_file, data)
        self._sync_generator()


def jpeg_size(data: bytes):
    """(height, width) from a JPEG stream's start-of-frame segment."""
    i = 2
    if data[:2] != b"\xff\xd8":
        raise ValueError("not a JPEG stream")
    while i + 9 < len(data):
        if data[i] != 0xFF:
            i += 1
            continue
        marker = data[i + 1]
        if marker in (0xD8, 0x01) or 0xD0 <= marker <= 0xD7 or marker == 0xFF:
            i += 2 if marker != 0xFF else 1
            continue
        length = int.from_bytes(data[i + 2:i + 4], "big")
        if 0xC0 <= marker <= 0xCF and marker not in (0xC4, 0xC8, 0xCC):
            return int.from_bytes(data[i + 5:i + 7], "big"), int.from_bytes(data[i + 7:i + 9], "big")
        i += 2 + length
    raise ValueError("no start-of-frame segment in the JPEG stream")


def main(engine_cls=None, dot=False):
    """Same splits, budgets and file names as upstream's __main__ (coord: :660-706; dot: OM_D:700-776)."""
    version = "v1_0"
    engine_cls = engine_cls or TwoFrameVideoQAEngine
    qtypes = ("tapvid3d_total_distance", "tapvid3d_displacement_vector")
    if dot:
        train_dir, val_dir = f"training_data_v2/object_movement_dot/{version}", f"evaluation_data_v2/object_movement_dot/{version}"
    else:
        train_dir, val_dir = f"training_data/object_movement_coord/{version}", f"evaluation_data/object_movement_coord/{version}"
    base_img_dir, base_npz, meta = "data/my_tapvid3d_images", "data/tapvid3d_dataset", "data/tapvid3d_dataset/meta_data"

    def ids(sub, split):
        with open(f"{meta}/{sub}/{split}.txt") as f:
            return [line.rstrip("\n") for line in f]
    for split, out_root, npoints, npairs, augment in (("val", val_dir, 1, 1, False), ("train", train_dir, 15, 30, True)):
        for q_type in qtypes:
            for sub in ("adt", "pstudio"):
                out_dir = f"{out_root}/{sub}"
                os.makedirs(out_dir, exist_ok=True)
                name = f"{sub}_{q_type}_val.jsonl" if split == "val" else f"{sub}_{q_type}_train_{npoints}points_{npairs}pairs.jsonl"
                eng = engine_cls(question_type=q_type, sub_dataset=sub)
                common = dict(scene_id_list=ids(sub, split), source_data_root=f"{base_npz}/{sub}", output_dir=out_dir,
                              output_file=os.path.join(out_dir, name), npoints_per_group=npoints, npairs_per_bin=npairs,
                              augment=augment)
                if dot:
                    img_dir = f"{out_dir}/{q_type}_images"
                    common.update(base_img_dir=base_img_dir, img_output_dir=img_dir)
                else:
                    common.update(img_output_dir=base_img_dir)
                if split == "val":
                    eng.generate_qa_eval_data(max_samples=300, **common)
                else:
                    eng.generate_qa_training_data(augment_ratio=0.05, **common)


if __name__ == "__main__":
    main()
