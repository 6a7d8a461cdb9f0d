"""Mirror of the reference's object_perception/single_object_perception_engine.py: QA records asking
for an object's height / length / width from the image combinations of the coverage tables."""
from __future__ import annotations

import json
import os
import pickle
import random
import shutil

import numpy as np

from mspa import heads
from mspa import templates as T
from mspa.hostinfo import quietly

random.seed(1)
np.random.seed(1)

max_train_samples = -1
val_max_samples = 3000
TEMPLATES = T.OBJECT_PERCEPTION      # swap in with TemplateSet.from_module(<module holding the upstream tables>)


def convert_train_sample_to_eval_sample(train_sample):
    return heads.to_eval_sample(train_sample)


def _image_hw(handler):
    """Upstream reads handler.image_height / image_width (OPE:207-208), attributes its own SceneInfoHandler
    does not define; fall back to the per-scene colour image shape."""
    if hasattr(handler, "image_height") and hasattr(handler, "image_width"):
        return (handler.image_height, handler.image_width)
    return lambda scene_id: tuple(handler.get_image_shape(scene_id))


@quietly
def build_lwh_qa_samples(scene_info_handler, dimension_info_path, dimension_name, split, output_dir, max_k=6, max_samples=-1):
    """One JSONL per combination size K: object_perception_{dimension}_k{K}_{split}_{max_samples}.jsonl."""
    print(f"Processing dimension: {dimension_name}, split: {split}")
    with open(dimension_info_path, "rb") as f:
        dim_info = pickle.load(f)
    os.makedirs(output_dir, exist_ok=True)
    getter = {"height": scene_info_handler.get_object_height, "length": scene_info_handler.get_object_length,
              "width": scene_info_handler.get_object_width}.get(dimension_name, lambda s, o: 0.0)
    by_k = heads.object_perception_records(dim_info, dimension_name, getter, scene_info_handler.get_object_raw_category,
                                           _image_hw(scene_info_handler), max_k,
                                           TEMPLATES, random)
    for k in range(1, max_k + 1):
        if not by_k[k]:
            continue
        if max_samples > 0 and len(by_k[k]) > max_samples:
            by_k[k] = random.sample(by_k[k], max_samples)
        path = os.path.join(output_dir, f"object_perception_{dimension_name}_k{k}_{split}_{max_samples}.jsonl")
        heads.write_jsonl(path, by_k[k])
        print(f"Written K={k} {len(by_k[k])} samples to {path}")
    print(f"Finished building QA samples for {dimension_name}.")


def build_train_and_val_datasets():
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    handler = SceneInfoHandler("data/scannet/scannet_instance_data/scenes_train_val_info_i_D5.pkl")
    train_dir, val_dir = "training_data/object_perception", "evaluation_data/object_perception"
    os.makedirs(train_dir, exist_ok=True)
    os.makedirs(val_dir, exist_ok=True)
    for dim in ("height", "length", "width"):
        build_lwh_qa_samples(handler, f"{train_dir}/merged_train_object_coverage_{dim}.pkl", dim, "train", train_dir,
                             max_k=6, max_samples=max_train_samples)
    temp = os.path.join(val_dir, "temp")
    os.makedirs(temp, exist_ok=True)
    for dim in ("height", "length", "width"):
        build_lwh_qa_samples(handler, f"{val_dir}/merged_val_object_coverage_{dim}.pkl", dim, "val", temp,
                             max_k=6, max_samples=val_max_samples)
    for fname in os.listdir(temp):
        with open(os.path.join(temp, fname)) as fin, open(os.path.join(val_dir, fname), "w") as fout:
            for line in fin:
                fout.write(json.dumps(convert_train_sample_to_eval_sample(json.loads(line))) + "\n")
    shutil.rmtree(temp)


def main():
    build_train_and_val_datasets()


if __name__ == "__main__":
    main()
