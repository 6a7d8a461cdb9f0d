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
