"""Mirror of the reference's object_perception/compute_object_visibility.py: per (object, image) visible
vertex counts as a masked popcount on the GPU."""
from __future__ import annotations

import json

import numpy as np
import torch

from mspa.scene import object_visibility_from_bits, pack_index_lists

NONINFORMATIVE_DESC = {"wall", "object", "floor", "ceiling", "window"}


def process_scene(scene_id, scene_info_handler, visibility_dict):
    """(scene_id, {"object_to_images": ..., "image_to_objects": ...}, warnings) for one scene.
    ``visibility_dict`` is the parquet-derived mapping "scene:image_to_points:img" -> JSON list string."""
    print(f"Processing scene {scene_id}.")
    warnings_list = []
    result = {"object_to_images": {}, "image_to_objects": {}}
    if scene_id not in scene_info_handler.infos:
        msg = f"[Warning] Scene {scene_id} not found in scene_info."
        warnings_list.append(msg)
        print(msg)
        return scene_id, result, warnings_list
    objects = {}
    for object_id in range(scene_info_handler.get_num_objects(scene_id)):
        if scene_info_handler.get_object_raw_category(scene_id, object_id) in NONINFORMATIVE_DESC:
            continue
        pts = scene_info_handler.get_object_point_index(scene_id, object_id)
        if len(pts) == 0:
            msg = f"[Warning] Scene {scene_id}, object {object_id} has no point indices, skipping."
            warnings_list.append(msg)
            print(msg)
            continue
        objects[object_id] = np.asarray(pts)
    image_ids, lists = [], []
    for image_id in scene_info_handler.get_all_extrinsic_valid_image_ids(scene_id):
        key = f"{scene_id}:image_to_points:{image_id}"
        if key not in visibility_dict:
            for object_id in objects:      # upstream warns once per (object, image)
                msg = f"[Warning] Scene {scene_id}, image {image_id} not found in visibility dict."
                warnings_list.append(msg)
                print(msg)
            continue
        image_ids.append(image_id)
        lists.append(json.loads(visibility_dict[key]))
    if not objects or not image_ids:
        return scene_id, result, warnings_list
    n_points = 1 + max(max((max(l) for l in lists if len(l)), default=0), max(int(p.max()) for p in objects.values()))
    image_bits = torch.from_numpy(pack_index_lists(lists, n_points)).cuda()
    result = object_visibility_from_bits(image_bits, image_ids, n_points, objects)
    return scene_id, result, warnings_list


def load_visibility_dict(parquet_file):
    import pandas as pd
    df = pd.read_parquet(parquet_file)
    return dict(zip(df["key"].tolist(), df["values"].tolist()))


def process_split(split_name, scene_info_path, visibility_parquet_file, output_dir):
    """All scenes of a split -> ``output_dir/object_visibility.pkl`` + ``warning.txt`` (reference: :153-195)."""
    import os
    import pickle
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    os.makedirs(output_dir, exist_ok=True)
    output_pkl_file = os.path.join(output_dir, "object_visibility.pkl")
    warning_file = os.path.join(output_dir, "warning.txt")
    scene_info_handler = SceneInfoHandler(scene_info_path)
    print(f"Loading visibility dict from {visibility_parquet_file}.")
    visibility_dict = load_visibility_dict(visibility_parquet_file)
    results, all_warnings = {}, []
    for scene_id in scene_info_handler.get_all_scene_ids():
        scene_id, scene_result, warnings = process_scene(scene_id, scene_info_handler, visibility_dict)
        results[scene_id] = scene_result
        all_warnings.extend(warnings)
    with open(warning_file, "w") as wf:
        for w in all_warnings:
            wf.write(w + "\n")
    with open(output_pkl_file, "wb") as f:
        pickle.dump(results, f, protocol=pickle.HIGHEST_PROTOCOL)
    print(f"Finished processing split '{split_name}'.")
    print(f"Result saved to {output_pkl_file}")
    print(f"Warnings saved to {warning_file}")


def main():
    """Same paths as upstream's main (:198-218)."""
    root = "data/scannet/scannet_instance_data"
    for split_name, out in (("val", "evaluation_data/object_perception"), ("train", "training_data/object_perception")):
        process_split(split_name, f"{root}/scenes_{split_name}_info_i_D5.pkl", f"{root}/{split_name}_visibility_info_D5.parquet", out)


if __name__ == "__main__":
    main()
