"""Mirror of the reference's object_perception/compute_object_visibility.py: which objects each image shows, and how much of
them -- per (object, image) the number of the object's vertices the image sees, as a masked popcount on the GPU instead of one
Python set intersection per pair (COVIS:103-150)."""
from __future__ import annotations

import json
import os
import pickle

import numpy as np
import torch

from mspa.scene import object_visibility_from_bits, pack_index_lists
from mspa.hostinfo import quietly

NONINFORMATIVE_DESC = {"wall", "object", "floor", "ceiling", "window"}


def _labelled_objects(scene_id, handler, warn):
    """{object id: vertex indices} of the objects worth asking about (COVIS:107-119)."""
    objects = {}
    for object_id in range(handler.get_num_objects(scene_id)):
        if handler.get_object_raw_category(scene_id, object_id) in NONINFORMATIVE_DESC:
            continue
        vertices = np.asarray(handler.get_object_point_index(scene_id, object_id))
        if vertices.size == 0:
            warn(f"[Warning] Scene {scene_id}, object {object_id} has no point indices, skipping.")
        else:
            objects[object_id] = vertices
    return objects


def _indexed_images(scene_id, handler, visibility_dict, n_objects, warn):
    """(image ids, their visible-vertex lists) for the frames the index knows; upstream's inner loop reports a missing frame
    once per object (COVIS:121-128), so does this."""
    image_ids, lists = [], []
    for image_id in handler.get_all_extrinsic_valid_image_ids(scene_id):
        text = visibility_dict.get(f"{scene_id}:image_to_points:{image_id}") if hasattr(visibility_dict, "get") else None
        if text is None:
            for _ in range(n_objects):
                warn(f"[Warning] Scene {scene_id}, image {image_id} not found in visibility dict.")
            continue
        image_ids.append(image_id)
        lists.append(json.loads(text))
    return image_ids, lists


def process_scene(scene_id, scene_info_handler, visibility_dict):
    """(scene_id, {"object_to_images": {obj: [{image_id, intersection_count, visibility}]}, "image_to_objects": {img: [...]}},
    warnings) of one scene (reference: :72-151; an image counts for an object from max(1, int(5 % of its vertices)) visible
    ones on).  ``visibility_dict``: "scene:image_to_points:img" -> JSON list text."""
    print(f"Processing scene {scene_id}.")
    warnings_list = []

    def warn(message):
        warnings_list.append(message)
        print(message)
    empty = {"object_to_images": {}, "image_to_objects": {}}
    if scene_id not in scene_info_handler.infos:
        warn(f"[Warning] Scene {scene_id} not found in scene_info.")
        return scene_id, empty, warnings_list
    objects = _labelled_objects(scene_id, scene_info_handler, warn)
    image_ids, lists = _indexed_images(scene_id, scene_info_handler, visibility_dict, len(objects), warn)
    if not objects or not image_ids:
        return scene_id, empty, warnings_list
    top = max(max((max(seen) for seen in lists if len(seen)), default=0), max(int(v.max()) for v in objects.values()))
    image_bits = torch.from_numpy(pack_index_lists(lists, top + 1)).cuda()
    return scene_id, object_visibility_from_bits(image_bits, image_ids, top + 1, objects), warnings_list


def load_visibility_dict(parquet_file):
    import pandas as pd
    df = pd.read_parquet(parquet_file)
    return dict(zip(df["key"].tolist(), df["values"].tolist()))


@quietly
def process_split(split_name, scene_info_path, visibility_parquet_file, output_dir, ctx=None):
    """``output_dir/object_visibility.pkl`` ({scene: result}, scenes in the split's order) and ``warning.txt`` for a split
    (reference: :153-195, one process, the whole index in memory).

    Sharded over the job's GPUs like the other split sweeps (``RANK`` / ``WORLD_SIZE`` from the environment, or ``ctx``;
    mspa/sweep.py): scenes dealt longest-first by objects x frames, every rank reads only ITS scenes' rows of the index
    (``visindex.SceneRowGroups``: row groups chosen by the footer's key statistics), counts on its GPU and ships the pickled
    result and the warning lines; rank 0 writes both files, identical for any number of ranks."""
    from mspa import shard, sweep, visindex
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    if ctx is None:
        ctx = shard.context_from_env()
    writer_rank = ctx is None or ctx.rank == 0
    handler = SceneInfoHandler(scene_info_path)
    scene_ids = handler.get_all_scene_ids()
    print(f"Loading visibility dict from {visibility_parquet_file}.")
    index = visindex.SceneRowGroups(visibility_parquet_file)
    costs = [float(max(1, handler.get_num_objects(s)) * max(1, len(handler.get_all_extrinsic_valid_image_ids(s)))) for s in scene_ids]
    results, lines = {}, []

    def produce(position, scene_id):
        _, result, warnings = process_scene(scene_id, handler, index.scene_dict(scene_id))
        return None, [pickle.dumps(result, protocol=pickle.HIGHEST_PROTOCOL), "".join(w + "\n" for w in warnings).encode()]

    def consume(position, _rows, blobs):
        results[scene_ids[position]] = pickle.loads(bytes(blobs[0]))
        lines.append(bytes(blobs[1]).decode())

    sweep.sharded_sweep(costs, ctx, lambda positions: (scene_ids[k] for k in positions), produce, consume)
    if ctx is not None:
        ctx.barrier()
    if not writer_rank:
        return
    os.makedirs(output_dir, exist_ok=True)
    pkl_path, warn_path = os.path.join(output_dir, "object_visibility.pkl"), os.path.join(output_dir, "warning.txt")
    with open(warn_path, "w") as f:
        f.writelines(lines)
    with open(pkl_path, "wb") as f:
        pickle.dump(results, f, protocol=pickle.HIGHEST_PROTOCOL)
    print(f"Finished processing split '{split_name}'.")
    print(f"Result saved to {pkl_path}")
    print(f"Warnings saved to {warn_path}")


def main():
    """Same paths as upstream's main (:198-218)."""
    root = "data/scannet/scannet_instance_data"
    for split_name, out in (("val", "evaluation_data/object_perception"), ("train", "training_data/object_perception")):
        process_split(split_name, f"{root}/scenes_{split_name}_info_i_D5.pkl", f"{root}/{split_name}_visibility_info_D5.parquet", out)


if __name__ == "__main__":
    main()
