"""Shared scaffolding of the two coordinate depth engines (reference: depth_estimation_coor_engine.py and
depth_comparison_coor_engine.py share their constructor, scene sampling and writers line for line)."""
from __future__ import annotations

import os
import random

from mspa import heads
from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler, VisibilityInfoHandler


class _LazyCounts:
    """len(visible points) per image, read from the visibility index only for the images that are sampled."""

    def __init__(self, lists):
        self._lists = lists

    def __getitem__(self, image_id):
        return len(self._lists(image_id))


class DepthCoorEngineBase:
    task_name = ""
    TEMPLATE_SET = None

    def __init__(self, scene_info_path, version_num="v1_0", all_max_samples=-1, image_output_dir=None,
                 visibility_info_path=None, max_n_points_per_image=1, warning_file=None):
        self.scene_info = SceneInfoHandler(scene_info_path)
        self.version_num = version_num
        self.image_output_dir = image_output_dir
        self.all_max_samples = all_max_samples
        self.max_n_points_per_image = max_n_points_per_image
        self.warning_file = warning_file
        self.visibility_info = VisibilityInfoHandler(visibility_info_path)
        self.max_samples = -1
        self.templates = self.TEMPLATE_SET
        self.annotator = None          # dot variants: mspa.annotate.PillowAnnotator() unless the caller sets another one

    # -- helpers ------------------------------------------------------------------------------
    def _warn(self, message):
        print(message.strip())
        if self.warning_file:
            with open(self.warning_file, "a") as wf:
                wf.write(message.strip())

    def _scene_inputs(self, scene_id):
        cache = {}

        def visible_points(image_id):
            if image_id not in cache:
                cache[image_id] = self.visibility_info.get_image_to_points_info(scene_id, image_id)
            return cache[image_id]
        scene = self.scene_info.scene_on_device(scene_id)
        return (self.scene_info.get_all_extrinsic_valid_image_ids(scene_id), _LazyCounts(visible_points),
                heads.list_point_numerics(scene, visible_points), self.scene_info.get_image_shape(scene_id))

    def _annotated_path(self, scene_id, name):
        return os.path.join(self.image_output_dir, scene_id, name)

    def _annotator(self):
        if self.annotator is None:
            from mspa.annotate import PillowAnnotator
            self.annotator = PillowAnnotator()
        return self.annotator

    def generate_qa_training_single_scene(self, scene_id):
        raise NotImplementedError

    # -- dataset level (reference: generate_qa_training_data / generate_qa_eval_data) --------------
    def generate_qa_training_data(self, output_dir, save_file=True):
        scene_ids = self.scene_info.get_sorted_keys()
        if self.all_max_samples > 0:
            self.max_samples = max(self.all_max_samples // len(scene_ids) + 1, 1)
            if self.max_samples == 1:
                scene_ids = random.sample(scene_ids, self.all_max_samples)
        else:
            self.max_samples = -1
        self.num_used_scenes = len(scene_ids)
        train_data = []
        for scene_id in scene_ids:
            train_data.extend(self.generate_qa_training_single_scene(scene_id))
        if len(train_data) > self.all_max_samples:
            train_data = random.sample(train_data, self.all_max_samples)
        random.shuffle(train_data)
        if not save_file:
            return train_data
        os.makedirs(output_dir, exist_ok=True)
        path = f"{output_dir}/{self.task_name}.jsonl"
        heads.write_jsonl(path, train_data)
        print(f"[Train] Training data saved to {path}. Generated {len(train_data)} samples in total.")

    def convert_train_sample_to_eval_sample(self, train_sample):
        train_sample["text"] = train_sample["conversations"][0]["value"]      # upstream keeps `conversations` here
        return train_sample

    def generate_qa_eval_data(self, output_dir):
        assert self.max_n_points_per_image == 1, "max_n_points_per_image should be 1 for evaluation"
        data = [self.convert_train_sample_to_eval_sample(s) for s in self.generate_qa_training_data(output_dir, save_file=False)]
        os.makedirs(output_dir, exist_ok=True)
        path = f"{output_dir}/{self.task_name}.jsonl"
        heads.write_jsonl(path, data)
        print(f"[Eval] Evaluation data saved to {path}. Generated {len(data)} samples in total.")


def run_cli(engine_cls, task_dir, argv=None):
    """The reference scripts' __main__: eval split first, then train, same defaults and paths."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--train_scene_info_path", default="data/scannet/scannet_instance_data/scenes_train_info_i_D5.pkl")
    ap.add_argument("--val_scene_info_path", default="data/scannet/scannet_instance_data/scenes_val_info_i_D5.pkl")
    ap.add_argument("--train_all_max_samples", type=int, default=500000)
    ap.add_argument("--val_all_max_samples", type=int, default=300)
    ap.add_argument("--output_dir_train", default=f"training_data/{task_dir}")
    ap.add_argument("--output_dir_val", default=f"evaluation_data/{task_dir}")
    ap.add_argument("--version_num", default="v1_0")
    args = ap.parse_args(argv)
    out_train = os.path.join(args.output_dir_train, args.version_num)
    out_val = os.path.join(args.output_dir_val, args.version_num)
    os.makedirs(out_train, exist_ok=True)
    os.makedirs(out_val, exist_ok=True)
    root = "data/scannet/scannet_instance_data"
    engine_cls(args.val_scene_info_path, args.version_num, args.val_all_max_samples, os.path.join(out_val, "images"),
               f"{root}/val_visibility_info_D5.parquet", warning_file=f"{out_val}/val_warning.txt").generate_qa_eval_data(out_val)
    engine_cls(args.train_scene_info_path, args.version_num, args.train_all_max_samples, os.path.join(out_train, "images"),
               f"{root}/train_visibility_info_D5.parquet",
               warning_file=f"{out_train}/train_warning.txt").generate_qa_training_data(out_train)
