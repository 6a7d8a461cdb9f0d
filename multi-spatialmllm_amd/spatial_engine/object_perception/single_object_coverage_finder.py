"""Mirror of the reference's object_perception/single_object_coverage_finder.py: minimal image
combinations whose union sees an object's full height / length / width.  The per-(object, image) extents
come from the GPU (K8), the breadth-first search over combinations from ``mspa.coverage``."""
from __future__ import annotations

import argparse
import json
import os
import pickle
import random

import numpy as np
import torch

from mspa import coverage as _cov
from mspa.scene import pack_index_lists
from mspa.hostinfo import quietly

random.seed(0)           # as upstream: the module seeds ``random`` when it is imported
TOLERANCE = _cov.TOLERANCE
DEBUG = False


def load_visibility_dict(parquet_file):
    import pandas as pd
    df = pd.read_parquet(parquet_file)
    return dict(zip(df["key"].tolist(), df["values"].tolist()))


def compute_coverage(scene_pts, indices_bool_mask, axis):
    if not indices_bool_mask.any():
        return None
    coords = scene_pts[indices_bool_mask][:, axis]
    return max(coords) - min(coords)


covers_dimension = _cov.covers_dimension


def _scene_bits(scene_id, image_ids, images_to_visible_points_dict, n_points):
    """(image ids present in the index, their bitset rows on the GPU); warns for the missing ones."""
    present, lists = [], []
    for img in image_ids:
        key = f"{scene_id}:image_to_points:{img}"
        if key not in images_to_visible_points_dict:
            print(f"[Warning] Scene {scene_id}, image {img} not found in visibility dict. Skip this combination.")
            continue
        present.append(img)
        lists.append(json.loads(images_to_visible_points_dict[key]))
    bits = torch.from_numpy(pack_index_lists(lists, n_points)).cuda() if present else None
    return present, bits


def find_minimal_combinations(scene_id, scene_pts, object_points_indices, visible_images,
                              images_to_visible_points_dict, axis, target_dim, tolerance, max_images=5):
    """{k: [minimal combinations of k images]} for one object along one axis."""
    present, bits = _scene_bits(scene_id, visible_images, images_to_visible_points_dict, len(scene_pts))
    if not present:
        return {}
    ext = _cov.scene_extents(bits, present, scene_pts, {0: np.asarray(object_points_indices)})
    lo, hi = ext.axis(0, present, axis)
    return _cov.minimal_combinations(present, lo, hi, target_dim, tolerance, max_images, rng=random)


def _targets(scene_info_handler, scene_id, object_id):
    return (scene_info_handler.get_object_height(scene_id, object_id),
            scene_info_handler.get_object_length(scene_id, object_id),
            scene_info_handler.get_object_width(scene_id, object_id),
            scene_info_handler.get_object_width_axis_aligned(scene_id, object_id))


def process_object(scene_id, object_id, scene_info_handler, visible_images, images_to_visible_points_dict):
    scene_pts = scene_info_handler.get_scene_points_align(scene_id)[:, :3]
    present, bits = _scene_bits(scene_id, visible_images, images_to_visible_points_dict, len(scene_pts))
    height, length, width, width_axis = _targets(scene_info_handler, scene_id, object_id)
    if not present:
        return {"height": {}, "length": {}, "width": {}}
    idx = scene_info_handler.get_object_point_index(scene_id, object_id)
    ext = _cov.scene_extents(bits, present, scene_pts, {object_id: np.asarray(idx)})
    return _cov.object_coverage(ext, object_id, present, height, length, width, width_axis, TOLERANCE, rng=random)


def process_scene_for_coverage(scene_id, scene_info_handler, images_to_visible_points_dict, object_visibility_dict):
    """(scene_id, {object_id: {"height": {...}, "length": {...}, "width": {...}}}); one K8 launch per scene."""
    print(f"Processing scene {scene_id} for object coverage.")
    scene_result = {}
    per_object = object_visibility_dict[scene_id]["object_to_images"]
    if not per_object:
        return scene_id, scene_result
    scene_pts = scene_info_handler.get_scene_points_align(scene_id)[:, :3]
    wanted = []
    for visibility_list in per_object.values():
        for entry in visibility_list:
            if entry["image_id"] not in wanted:
                wanted.append(entry["image_id"])
    index_key = f"{scene_id}:image_to_points:"
    present = [img for img in wanted if index_key + img in images_to_visible_points_dict]
    bits = (torch.from_numpy(pack_index_lists([json.loads(images_to_visible_points_dict[index_key + img]) for img in present],
                                              len(scene_pts))).cuda() if present else None)
    objects = {o: np.asarray(scene_info_handler.get_object_point_index(scene_id, o)) for o in per_object}
    ext = _cov.scene_extents(bits, present, scene_pts, objects) if present else None
    for object_id, visibility_list in per_object.items():
        visible_images = [entry["image_id"] for entry in visibility_list]
        height, length, width, width_axis = _targets(scene_info_handler, scene_id, object_id)
        res = {}
        length_axis = 1 if width_axis == 0 else 0
        for name, axis, target in (("height", 2, height), ("length", length_axis, length), ("width", width_axis, width)):
            images = []
            for img in visible_images:       # upstream warns once per axis for every image missing from the index
                if ext is not None and img in ext.image_index:
                    images.append(img)
                else:
                    print(f"[Warning] Scene {scene_id}, image {img} not found in visibility dict. Skip this combination.")
            if images:
                lo, hi = ext.axis(object_id, images, axis)
                res[name] = _cov.minimal_combinations(images, lo, hi, target, TOLERANCE, rng=random)
            else:
                res[name] = {}
        scene_result[object_id] = res
    return scene_id, scene_result


def _load_inputs(scene_info_path, visibility_parquet_file, object_visibility_file):
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler, _load_any
    return (SceneInfoHandler(scene_info_path), load_visibility_dict(visibility_parquet_file),
            _load_any(object_visibility_file))


def _run_scenes(scene_ids, handler, vis_dict, obj_vis):
    tables = {"height": {}, "length": {}, "width": {}}
    for scene_id in scene_ids:
        scene_id, scene_result = process_scene_for_coverage(scene_id, handler, vis_dict, obj_vis)
        if scene_result:
            for dim in tables:
                tables[dim][scene_id] = {o: res[dim] for o, res in scene_result.items()}
    return tables


@quietly
def process_split_objects(split_name, scene_info_path, visibility_parquet_file, object_visibility_file, output_dir):
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "coverage_finding_warning_objects.txt"), "w") as wf:
        wf.write("")
    handler, vis_dict, obj_vis = _load_inputs(scene_info_path, visibility_parquet_file, object_visibility_file)
    scene_ids = ["scene0011_00"] if DEBUG else handler.get_all_scene_ids()
    tables = _run_scenes(scene_ids, handler, vis_dict, obj_vis)
    for dim, table in tables.items():
        with open(os.path.join(output_dir, f"{split_name}_object_coverage_{dim}.pkl"), "wb") as f:
            pickle.dump(table, f)
    print(f"Finished processing split '{split_name}' for object coverage.")


SPLITS = {
    "val": {"scene_info_path": "data/scannet/scannet_instance_data/scenes_val_info_i_D5.pkl",
            "visibility_parquet_file": "data/scannet/scannet_instance_data/val_visibility_info_D5.parquet",
            "object_visibility_file": "evaluation_data/object_perception/object_visibility.pkl",
            "output_dir": "evaluation_data/object_perception"},
    "train": {"scene_info_path": "data/scannet/scannet_instance_data/scenes_train_info_i_D5.pkl",
              "visibility_parquet_file": "data/scannet/scannet_instance_data/train_visibility_info_D5.parquet",
              "object_visibility_file": "training_data/object_perception/object_visibility.pkl",
              "output_dir": "training_data/object_perception"},
}


def main(argv=None):
    parser = argparse.ArgumentParser(description="Object coverage for a split and a scene index range.")
    parser.add_argument("--split", type=str, required=True)
    parser.add_argument("--start", type=int, required=True)
    parser.add_argument("--end", type=int, default=None)
    args = parser.parse_args(argv)
    if args.split not in SPLITS:
        raise ValueError("Invalid split. Choose train or val.")
    cfg = dict(SPLITS[args.split])
    out_dir = os.path.join(cfg["output_dir"], f"{args.split}_{args.start}_{args.end}")
    os.makedirs(out_dir, exist_ok=True)
    handler, vis_dict, obj_vis = _load_inputs(cfg["scene_info_path"], cfg["visibility_parquet_file"], cfg["object_visibility_file"])
    scene_ids = handler.get_all_scene_ids()[args.start:args.end]
    tables = _run_scenes(scene_ids, handler, vis_dict, obj_vis)
    for dim, table in tables.items():
        with open(os.path.join(out_dir, f"{args.split}_object_coverage_{dim}_{args.start}_{args.end}.pkl"), "wb") as f:
            pickle.dump(table, f)
    print(f"Finished processing split '{args.split}' for scenes {args.start} to {args.end}.")


if __name__ == "__main__":
    main()
