"""Mirror of the reference's camera_movement/camera_movement_engine_train_val.py: per-row record builder
with the reference's signature on top of ``mspa.heads`` (K4 for the relative pose)."""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from mspa import engine, heads
from mspa import templates as T
from mspa.sampling import sample_dataframe  # noqa: F401  (same name as upstream)

try:    # the reference imports its tables from a sibling TEMPLATES module; use one if the user provides it
    import TEMPLATES as _user_tables
    TEMPLATE_SET = T.TemplateSet.from_module(_user_tables)
except ImportError:
    TEMPLATE_SET = T.CAMERA_MOVEMENT


def build_training_sample(scene_infos, row, idx: int, question_type: str):
    """One camera-movement record for a pair-table row (reference: :153-245)."""
    scene_id, image1, image2 = row["scene_id"], row["image_id1"], row["image_id2"]
    E1 = scene_infos.get_extrinsic_matrix_align(scene_id, image1)
    E2 = scene_infos.get_extrinsic_matrix_align(scene_id, image2)
    assert not np.isnan(E1).any(), f"E1 is nan for {scene_id} {image1}"
    assert not np.isnan(E2).any(), f"E2 is nan for {scene_id} {image2}"
    E_t = torch.from_numpy(np.stack([E1, E2]).reshape(2, 16)).cuda()
    Einv_t = torch.from_numpy(np.stack([np.linalg.inv(E1), np.linalg.inv(E2)]).reshape(2, 16)).cuda()
    zeros = torch.zeros(2, dtype=torch.float64, device="cuda")
    pairs = torch.tensor([[0, 1], [1, 0]], dtype=torch.int32, device="cuda")
    out = engine.pair_pose(E_t, Einv_t, zeros, zeros, pairs).cpu().numpy()
    return heads.camera_movement_record(row, idx, question_type, out[0, 3:6], out[1, 3:6],
                                        scene_infos.get_image_shape(scene_id, image1), TEMPLATE_SET, random)


convert_train_sample_to_eval_sample = heads.to_eval_sample


def _build_samples(parquet_path, scene_infos, qtype, desired_count, overlap_min, overlap_max, interval, tag, transform=None):
    """(records in row order, communicator or None).  In a job with one process per GPU (RANK / WORLD_SIZE from the
    launcher) every rank samples the same rows (seeded) and makes every row's draws, formats its own slice of the records
    and rank 0 receives the finished lines (``heads.camera_movement_dataset``); upstream builds them in one loop
    (CME:295-299)."""
    import pandas as pd
    from mspa import shard
    ctx = shard.context_from_env()
    df = pd.read_parquet(parquet_path)
    print(f"[{tag}: {qtype}] Loaded DF with {len(df)} rows from {parquet_path}")
    print(f"[{tag}: {qtype}] sampling {desired_count} samples in overlap=[{overlap_min}..{overlap_max}]")
    df_sampled = sample_dataframe(df, all_overlap_samples=desired_count, non_overlap_samples=0, overlap_min=overlap_min,
                                  overlap_max=overlap_max, interval=interval)
    print(f"[{tag}: {qtype}] got {len(df_sampled)} sampled rows")
    rows = df_sampled.to_dict("records")          # plain dicts in row order (a pandas Series per row costs ~10 us: 30 s per 3 M rows)
    # one K4 launch for the relative poses of the rows a rank formats, then the records in row order
    samples = heads.camera_movement_dataset(rows, scene_infos.get_extrinsic_matrix_align, scene_infos.get_image_shape, qtype,
                                            TEMPLATE_SET, random, device=ctx.device if ctx is not None else "cuda", ctx=ctx,
                                            transform=transform)
    return samples, len(rows), ctx


def _shuffle_and_write(samples, n_rows, ctx, out_file, tag, qtype):
    """``random.shuffle(out_samples)`` + the JSONL (CME:301-308).  The shuffle is applied to an index list of the same
    length -- the same permutation, and every rank of a sharded job advances its generator exactly as rank 0 does."""
    order = list(range(n_rows))
    random.shuffle(order)
    if ctx is not None and ctx.rank != 0:
        ctx.barrier()
        return
    print(f"[{tag}: {qtype}] writing {len(samples)} items to {out_file}")
    heads.write_jsonl(out_file, [samples[i] for i in order])
    if ctx is not None:
        ctx.barrier()


def build_train_dataset(parquet_path, output_dir, scene_infos, qtype, desired_count, overlap_min, overlap_max, interval):
    """{qtype}_train.jsonl from the pair table (reference: :271-308)."""
    samples, n, ctx = _build_samples(parquet_path, scene_infos, qtype, desired_count, overlap_min, overlap_max, interval, "Train")
    _shuffle_and_write(samples, n, ctx, os.path.join(output_dir, f"{qtype}_train.jsonl"), "Train", qtype)


def build_val_dataset(parquet_path, output_dir, scene_infos, qtype, desired_count, overlap_min, overlap_max, interval):
    """{qtype}_val.jsonl in the eval form (reference: :314-354)."""
    samples, n, ctx = _build_samples(parquet_path, scene_infos, qtype, desired_count, overlap_min, overlap_max, interval, "Val",
                                     transform=convert_train_sample_to_eval_sample)
    _shuffle_and_write(samples, n, ctx, os.path.join(output_dir, f"{qtype}_val.jsonl"), "Val", qtype)


DEBUG = False
TRAIN_QUESTION_SAMPLES = {"x_movement": 1000000, "y_movement": 1000000, "z_movement": 1000000, "yaw_movement": 1000000,
                          "pitch_movement": 1000000, "yaw_angle": 1000000, "pitch_angle": 1000000, "total_distance": 3000000,
                          "displacement_vector": 3000000}
VAL_QUESTION_ORDER = ("x_movement", "y_movement", "z_movement", "yaw_movement", "pitch_movement", "total_distance", "yaw_angle",
                      "pitch_angle", "displacement_vector")


def main():
    """Same paths, budgets, seeds and order of question types as upstream's main (:360-444)."""
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    np.random.seed(0)
    random.seed(0)
    info_path = "data/scannet/scannet_instance_data/scenes_train_val_info_i_D5.pkl"
    overlap_min, overlap_max, interval, version = 6, 35, 1, "v1_0"
    train_counts = dict(TRAIN_QUESTION_SAMPLES)
    val_counts = {k: 300 for k in VAL_QUESTION_ORDER}
    if DEBUG:
        train_parquet = "training_data/camera_movement/train_camera_info_D5_debug_nonzero.parquet"
        val_parquet = "evaluation_data/camera_movement/val_camera_info_D5_debug_nonzero.parquet"
        train_counts = {k: 100 for k in train_counts}
        val_counts = {k: 100 for k in val_counts}
        version += "_debug"
    else:
        train_parquet = "training_data/camera_movement/train_camera_info_D5.parquet"
        val_parquet = "evaluation_data/camera_movement/val_camera_info_D5.parquet"
    train_dir, val_dir = f"training_data/camera_movement/{version}", f"evaluation_data/camera_movement/{version}"
    os.makedirs(train_dir, exist_ok=True)
    os.makedirs(val_dir, exist_ok=True)
    scene_infos = SceneInfoHandler(info_path)
    for qtype in train_counts:
        print(f"\n=== Processing question type: {qtype} ===")
        build_val_dataset(val_parquet, val_dir, scene_infos, qtype, val_counts[qtype], overlap_min, overlap_max, interval)
        build_train_dataset(train_parquet, train_dir, scene_infos, qtype, train_counts[qtype], overlap_min, overlap_max, interval)
    print("All question types processed. Done.")


if __name__ == "__main__":
    main()
