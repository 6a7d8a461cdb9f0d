"""Mirror of the reference's single_object_movement_engine_dot.py: the object-movement questions about a point that is
marked with a disc on the first frame instead of being named by coordinates."""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from mspa import engine, heads
from mspa import templates as T
from mspa.annotate import Mark
from mspa.hostinfo import quietly
from spatial_engine.object_movement.single_object_movement_engine_coord import (TwoFrameVideoQAEngine, filter_large_groups,
                                                                                jpeg_size, rigid_body_segmentation,
                                                                                sharded_scenes)
from spatial_engine.object_movement.single_object_movement_engine_coord import main as _main

random.seed(1)
np.random.seed(1)


class TwoFrameVideoQAEngineDot(TwoFrameVideoQAEngine):
    def __init__(self, question_type, sub_dataset):
        super().__init__(question_type, sub_dataset)
        self.templates = T.OBJECT_MOVEMENT_DOT
        drive = sub_dataset == "drivetrack"                     # looser thresholds and a 5-frame window there (reference: :301-303)
        self.object_not_moving_threshold = 0.1 if drive else 0.01
        self.camera_not_moving_threshold = 0.1 if drive else 0.01
        self.future_frame_windows = 5 if drive else 1e8
        self.annotator = None

    def _annotator(self):
        if self.annotator is None:
            from mspa.annotate import PillowAnnotator
            self.annotator = PillowAnnotator()
        return self.annotator

    def format_training_samples(self, sample_pairs, intrinsics, scene_id, points_pos_world, points_pos_cam, image_height,
                                image_width, extrinsics_w2c, base_img_dir, img_output_dir):
        """Records + annotated first frames (reference: :341-436).  A first frame is annotated once per (frame, point);
        the colour is only drawn from ``random`` when that file is still missing, as upstream."""
        save_dir = os.path.join(img_output_dir, scene_id)
        os.makedirs(save_dir, exist_ok=True)
        radius = int(image_width) // 100

        def needs(name):
            return not os.path.exists(os.path.join(save_dir, name))

        def on_mark(frame1, frame2, point_index, pixel, color):
            if color is not None:
                self._annotator().annotate(os.path.join(base_img_dir, scene_id, f"{frame1:05d}.jpg"),
                                           os.path.join(save_dir, f"{frame1:05d}_{point_index}_annotated.jpg"),
                                           [Mark(pixel[0], pixel[1], radius, color)])
            if needs(f"{frame2:05d}.jpg"):
                self._annotator().copy(os.path.join(base_img_dir, scene_id, f"{frame2:05d}.jpg"),
                                       os.path.join(save_dir, f"{frame2:05d}.jpg"))
        return heads.object_movement_records(scene_id, np.asarray(points_pos_cam), np.asarray(extrinsics_w2c), intrinsics,
                                             (int(image_height), int(image_width)), sample_pairs, self.question_type,
                                             self.templates, random, dot=True, needs_annotation=needs, on_mark=on_mark,
                                             obj_threshold=self.object_not_moving_threshold,
                                             cam_threshold=self.camera_not_moving_threshold)

    def generate_qa_training_single_scene(self, input_file, base_img_dir, npoints_per_group=5, npairs_per_bin=1e8,
                                          img_output_dir="", augment=True, augment_ratio=1.0, _loaded=None):
        scene_id = os.path.splitext(os.path.basename(input_file))[0]
        gt = _loaded if _loaded is not None else np.load(input_file, allow_pickle=True)
        scene_img_dir = os.path.join(base_img_dir, scene_id)
        os.makedirs(scene_img_dir, exist_ok=True)
        payloads = gt["images_jpeg_bytes"]
        if len([n for n in os.listdir(scene_img_dir) if n.endswith(".jpg")]) != payloads.shape[0]:
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
        return self.format_training_samples(sample_pairs, intrinsics, scene_id, None, tracks_xyz, image_height, image_width,
                                            extrinsics_w2c, base_img_dir, img_output_dir)

    @quietly
    def _all_scenes_dot(self, scene_id_list, source_data_root, base_img_dir, img_output_dir, npoints_per_group, npairs_per_bin,
                        augment, augment_ratio, num_workers=20, ctx=None):
        parent = random.getstate()                  # fork-pool semantics, see TwoFrameVideoQAEngine._all_scenes

        def run_one(path, loaded):
            random.setstate(parent)
            return self.generate_qa_training_single_scene(path, base_img_dir, npoints_per_group, npairs_per_bin, img_output_dir,
                                                          augment, augment_ratio, _loaded=loaded)
        data, self._ctx = sharded_scenes([os.path.join(source_data_root, f"{scene_id}.npz") for scene_id in scene_id_list],
                                         run_one, num_workers, ctx)
        random.setstate(parent)
        return data

    def generate_qa_training_data(self, scene_id_list, source_data_root, base_img_dir, output_dir, output_file, img_output_dir,
                                  npoints_per_group, npairs_per_bin, augment, augment_ratio=1.0, max_samples=-1, num_workers=20):
        data = self._all_scenes_dot(scene_id_list, source_data_root, base_img_dir, img_output_dir, npoints_per_group,
                                    npairs_per_bin, augment, augment_ratio, num_workers)
        if self._is_writer():
            if max_samples > 0 and len(data) > max_samples:
                data = random.sample(data, max_samples)
            random.shuffle(data)
            heads.write_jsonl(output_file, data)
            self._report("Training", output_file, data)
        self._sync_generator()

    def generate_qa_eval_data(self, scene_id_list, source_data_root, base_img_dir, output_dir, output_file, img_output_dir,
                              npoints_per_group, npairs_per_bin, augment, augment_ratio=0.3, max_samples=300, num_workers=20):
        """Writes ``*_orig.jsonl`` (everything) and the subsampled file (reference: :640-683)."""
        data = self._all_scenes_dot(scene_id_list, source_data_root, base_img_dir, img_output_dir, npoints_per_group,
                                    npairs_per_bin, augment, augment_ratio, num_workers)
        if self._is_writer():
            eval_data = [self.format_eval_sample(s) for s in data]
            heads.write_jsonl(output_file.replace(".jsonl", "_orig.jsonl"), eval_data)
            subsampled = random.sample(eval_data, max_samples) if max_samples > 0 and len(eval_data) > max_samples else eval_data
            heads.write_jsonl(output_file, subsampled)
            self._report("Original evaluation", output_file.replace(".jsonl", "_orig.jsonl"), data)
            self._report("Subsampled evaluation", output_file, subsampled)
        self._sync_generator()


if __name__ == "__main__":
    _main(TwoFrameVideoQAEngineDot, dot=True)
