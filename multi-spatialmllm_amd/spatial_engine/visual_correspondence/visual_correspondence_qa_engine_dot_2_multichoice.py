"""Mirror of the reference's visual_correspondence_qa_engine_dot_2_multichoice.py: a dot on a point of Image-1, four
lettered candidates in Image-2, answer = the letter.  Numerics batched per scene (K2 / K6a / K6b); annotated images
through ``mspa.annotate``."""
from __future__ import annotations

import os
import random

import numpy as np

from mspa import heads
from mspa import templates as T
from mspa.annotate import Mark
from mspa.sampling import sample_dataframe  # noqa: F401
from spatial_engine.visual_correspondence.visual_correspondence_qa_engine_coor_2_coor import (_ResidentScenes, _load_visibility,
                                                                                               _shuffle_and_write)

random.seed(2)
np.random.seed(2)

TEMPLATE_SET = T.VISUAL_CORRESPONDENCE_DOT
USE_PICKLE = True
DEBUG = False
ANNOTATOR = None          # mspa.annotate.PillowAnnotator() unless the caller installs another one


def _annotator():
    global ANNOTATOR
    if ANNOTATOR is None:
        from mspa.annotate import PillowAnnotator
        ANNOTATOR = PillowAnnotator()
    return ANNOTATOR


def convert_parquet_to_dict(parquet_df):
    return dict(zip(parquet_df["key"].tolist(), parquet_df["values"].tolist()))      # values stay JSON strings here (:262-277)


def _marker(scene_infos, image_output_dir):
    def on_mark(idx, scene_id, image1, image2, vertex, p1_pixel, color1, labeled_points, colors):
        stem = os.path.join(image_output_dir, scene_id, f"{idx}_point{vertex}_{image1}_{image2}")
        _annotator().annotate(scene_infos.get_image_path(scene_id, image1), stem + "_img1.jpg",
                              [Mark(p1_pixel[0], p1_pixel[1], 10, color1)])
        _annotator().annotate(scene_infos.get_image_path(scene_id, image2), stem + "_img2.jpg",
                              [Mark(x, y, 10, colors[label], label, (15, 0)) for label, (x, y) in labeled_points.items()])
    return on_mark


def _records(rows, scene_infos, visibility_info_dict, warning_file, image_output_dir, resident=None, ctx=None, transform=None):
    resident = resident or _ResidentScenes(scene_infos, visibility_info_dict)

    def warn(message):
        print(message.strip())
        with open(warning_file, "a") as wf:
            wf.write(message)
    backend = heads.GpuCorrespondenceBackend(resident.get, resident.get_bits)
    return heads.visual_correspondence_dot_dataset(rows, backend, TEMPLATE_SET, random, warn, _marker(scene_infos, image_output_dir),
                                                   ctx=ctx, transform=transform)


def build_training_sample(scene_infos, row, idx: int, visibility_info_dict, warning_file, max_points_per_pair=1,
                          image_output_dir="images_debug"):
    """One multiple-choice record (reference: :279-433).  ``idx`` names the record and its two annotated images."""
    assert max_points_per_pair == 1, "[build_training_sample] max_points_per_pair should be 1."
    rec = _records([row], scene_infos, visibility_info_dict, warning_file, image_output_dir)[0]
    if rec is not None and idx != 0:                      # the batch numbers its rows from 0: rename to the caller's idx
        vertex = rec["id"].split("_p")[1]
        old = f"0_point{vertex}_"
        for k, path in enumerate(rec["image"]):
            new_path = path.replace(os.sep + old, os.sep + f"{idx}_point{vertex}_")
            src, dst = os.path.join(image_output_dir, path), os.path.join(image_output_dir, new_path)
            if os.path.exists(src):
                os.replace(src, dst)
            rec["image"][k] = new_path
        rec["id"] = f"{idx}_p{vertex}"
    return rec


def convert_train_sample_to_eval_sample(train_sample):
    return heads.to_eval_sample(train_sample)


def _build(parquet_path, output_dir, scene_infos, desired_count, overlap_min, overlap_max, interval, visibility_info_path,
           warning_file, tag, transform=None):
    """(records in row order -- rows without one dropped --, how many there are on every rank, communicator or None).  In a job
    with one process per GPU the scenes of the sampled rows are dealt over the ranks (``heads.visual_correspondence_dot_dataset``):
    each rank reads and uploads its own scenes and draws the two annotated JPEGs of its own records -- the expensive part of
    this head; upstream loops over the rows in one process (VC_D:455-466)."""
    import pandas as pd
    import torch
    from mspa import shard
    ctx = shard.context_from_env()
    df = pd.read_parquet(parquet_path)
    print(f"[{tag}] Loaded DataFrame with {len(df)} rows from {parquet_path}")
    print(f"[{tag}] Sampling {desired_count} samples with overlap in [{overlap_min}, {overlap_max}]")
    df_sampled = sample_dataframe(df, all_overlap_samples=desired_count, non_overlap_samples=0, overlap_min=overlap_min,
                                  overlap_max=overlap_max, interval=interval)
    print(f"[{tag}] Got {len(df_sampled)} sampled rows")
    image_output_dir = os.path.join(output_dir, "images")
    os.makedirs(image_output_dir, exist_ok=True)
    rows = [df_sampled.iloc[k] for k in range(len(df_sampled))]
    samples = _records(rows, scene_infos, _load_visibility(visibility_info_path), warning_file, image_output_dir, ctx=ctx,
                       transform=transform)
    kept = [s for s in samples if s]
    if ctx is None:
        return kept, len(kept), None
    import torch.distributed as dist                          # every rank shuffles an index list of rank 0's length (the generator stays in step)
    n_kept = torch.tensor([len(kept)], dtype=torch.int64, device=ctx.collective_device)
    dist.broadcast(n_kept, src=0, group=ctx.group)
    return kept, int(n_kept.item()), ctx


def build_train_dataset(parquet_path, output_dir, scene_infos, desired_count, overlap_min, overlap_max, interval,
                        visibility_info_path, warning_file, max_points_per_pair=1):
    samples, n, ctx = _build(parquet_path, output_dir, scene_infos, desired_count, overlap_min, overlap_max, interval,
                             visibility_info_path, warning_file, "Train")
    _shuffle_and_write(samples, n, ctx, os.path.join(output_dir, "train_visual_correspondence_dot_2_multichoice.jsonl"), "Train")


def build_val_dataset(parquet_path, output_dir, scene_infos, desired_count, overlap_min, overlap_max, interval,
                      visibility_info_path, warning_file, max_points_per_pair=1):
    assert max_points_per_pair == 1, "[Val] max_points_per_pair should be 1."
    samples, n, ctx = _build(parquet_path, output_dir, scene_infos, desired_count, overlap_min, overlap_max, interval,
                             visibility_info_path, warning_file, "Val", transform=convert_train_sample_to_eval_sample)
    _shuffle_and_write(samples, n, ctx, os.path.join(output_dir, "val_visual_correspondence_dot_2_multichoice.jsonl"), "Val")


def main():
    """Same paths and budgets as upstream's main (500 K train / 300 val)."""
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    random.seed(2)
    np.random.seed(2)
    info_path = "data/scannet/scannet_instance_data/scenes_train_val_info_i_D5.pkl"
    ext = "pkl" if USE_PICKLE else "parquet"
    train_vis = f"data/scannet/scannet_instance_data/train_visibility_info_D5.{ext}"
    val_vis = f"data/scannet/scannet_instance_data/val_visibility_info_D5.{ext}"
    version = "v1_0" + ("_debug" if DEBUG else "")
    suffix = "_debug_nonzero" if DEBUG else ""
    train_parquet = f"training_data/camera_movement/train_camera_info_D5{suffix}.parquet"
    val_parquet = f"evaluation_data/camera_movement/val_camera_info_D5{suffix}.parquet"
    train_max, val_max = (100, 100) if DEBUG else (500000, 300)
    train_dir = os.path.join("training_data/visual_correspondence_dot_2_multichoice", version)
    val_dir = os.path.join("evaluation_data/visual_correspondence_dot_2_multichoice", version)
    os.makedirs(train_dir, exist_ok=True)
    os.makedirs(val_dir, exist_ok=True)
    scene_infos = SceneInfoHandler(info_path)
    build_val_dataset(val_parquet, val_dir, scene_infos, val_max, 6, 35, 1, val_vis, os.path.join(val_dir, "val_warning.txt"))
    build_train_dataset(train_parquet, train_dir, scene_infos, train_max, 6, 35, 1, train_vis,
                        os.path.join(train_dir, "train_warning.txt"))


if __name__ == "__main__":
    main()
