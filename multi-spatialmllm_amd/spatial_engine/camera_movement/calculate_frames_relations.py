"""Mirror of the reference's camera_movement/calculate_frames_relations.py (per-scene pair table)."""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch

from mspa import engine

DEBUG = False

_COLUMNS = ["scene_id", "image_id1", "image_id2", "overlap", "distance", "yaw", "pitch"]


def _rows(overlap_info, keep):
    return [{"scene_id": scene_id, "image_id1": a, "image_id2": b, "overlap": v["overlap"], "distance": v["distance"],
             "yaw": v["yaw"], "pitch": v["pitch"]}
            for scene_id, pairs in overlap_info.items() for (a, b), v in pairs.items() if keep(v)]


def save_overlap_info(overlap_info, parquet_file):
    """{scene: {(id1, id2): {overlap, distance, yaw, pitch}}} -> parquet with the pair-table columns
    scene_id, image_id1, image_id2, overlap, distance, yaw, pitch (reference: :28-57)."""
    rows = _rows(overlap_info, lambda v: True)
    if not rows:
        print(f"[save_overlap_info] Nothing to save to {parquet_file}.")
        return
    df = pd.DataFrame(rows, columns=_COLUMNS)
    df.to_parquet(parquet_file, index=False)
    print(f"[save_overlap_info] Saved {len(df)} records to {parquet_file}.")


def save_overlap_info_nonzero(overlap_info, parquet_file_nonzero):
    """Same table without the rows whose overlap == 0 (NaN overlaps stay: nan != 0.0, reference: :59-84)."""
    rows = _rows(overlap_info, lambda v: v["overlap"] != 0.0)
    if not rows:
        print("[save_overlap_info_nonzero] No nonzero-overlap pairs to save.")
        return
    df = pd.DataFrame(rows, columns=_COLUMNS)
    df.to_parquet(parquet_file_nonzero, index=False)
    print(f"[save_overlap_info_nonzero] Saved {len(df)} records to {parquet_file_nonzero}.")


def extract_yaw_pitch(R):
    """Yaw / pitch (degrees) of the camera's forward axis; 4x4 or 3x3 input (reference: :86-100)."""
    R = np.asarray(R)
    R3 = R[:3, :3] if R.shape == (4, 4) else R
    z = R3[:, 2]
    return np.degrees(np.arctan2(z[1], z[0])), np.degrees(np.arcsin(z[2] / np.linalg.norm(z)))


def calculate_camera_overlap(in_bounds_dict, image_id1, image_id2, use_cuda=False):
    """|a & b| / |a | b| * 100 of two boolean masks (reference: :102-137), on the GPU (K2).
    ``use_cuda`` is accepted for signature compatibility; the computation is always on the device."""
    a = np.ascontiguousarray(in_bounds_dict[image_id1], dtype=bool)
    b = np.ascontiguousarray(in_bounds_dict[image_id2], dtype=bool)
    n = a.shape[0]
    n_words = (n + 63) // 64
    packed = np.zeros((2, n_words * 8), dtype=np.uint8)
    packed[0, :(n + 7) // 8] = np.packbits(a, bitorder="little")
    packed[1, :(n + 7) // 8] = np.packbits(b, bitorder="little")
    bits = torch.from_numpy(packed.view(np.int64)).cuda()
    pairs = torch.tensor([[0, 1]], dtype=torch.int32, device="cuda")
    return np.float64(engine.pair_overlap(bits, pairs).cpu().numpy()[0])


def process_scene(scene_id, scene_infos, warning_file):
    """(scene_id, {(id1, id2): {overlap, distance, yaw, pitch}}) for all valid frame pairs i < j
    (reference: :139-197): K1 visibility bitsets, K2 overlaps, K4 pose differences."""
    print(f"Start processing {scene_id}.")
    scene = scene_infos.scene_on_device(scene_id)
    for image_id in scene.empty_frames():
        with open(warning_file, "a") as f:
            f.write(f"{scene_id}: {image_id} has no in bound points\n")
    table = scene.frames_relations()
    for key, vals in table.items():
        v = list(vals.values())
        if np.any(np.isnan(v)) or np.any(np.isinf(v)):
            with open(warning_file, "a") as f:
                f.write(f"{scene_id}: {key} has something wrong {v}. \n")
    print(f"Finished scene {scene_id}.")
    return scene_id, table


def run_split(scene_info_path, output_parquet, warning_file, num_workers=15, save_interval=20):
    """Pair tables of every scene of a split -> ``output_parquet`` (+ ``*_nonzero.parquet``), with the
    reference's periodic partial saves (reference: :200-253).  ``num_workers`` is accepted and ignored:
    scenes run back to back on the GPU (one ScanNet-sized scene takes ~0.25 ms of kernel time)."""
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    scene_infos = SceneInfoHandler(scene_info_path)
    overlap_info = {}
    all_scene_ids = scene_infos.get_all_scene_ids()
    print(f"[run_split] Found {len(all_scene_ids)} scenes in {scene_info_path}.")
    if DEBUG and len(all_scene_ids) > 1:
        all_scene_ids = all_scene_ids[:1]
        print("[run_split] DEBUG mode: processing only the first scene.")
    nonzero_parquet = output_parquet.replace(".parquet", "_nonzero.parquet")
    out_dir = os.path.dirname(output_parquet)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    for count, scene_id in enumerate(all_scene_ids):
        _, overlap_info[scene_id] = process_scene(scene_id, scene_infos, warning_file)
        if (count + 1) % save_interval == 0:
            save_overlap_info(overlap_info, output_parquet)
            save_overlap_info_nonzero(overlap_info, nonzero_parquet)
            print(f"[run_split] Saved partial results for {count + 1} scenes to {output_parquet}")
    save_overlap_info(overlap_info, output_parquet)
    save_overlap_info_nonzero(overlap_info, nonzero_parquet)
    total = sum(len(v) for v in overlap_info.values())
    nonzero = sum(1 for scene in overlap_info.values() for pair in scene.values() if pair["overlap"] != 0.0)
    print(f"[run_split] Total number of records: {total}")
    print(f"[run_split] Total number of nonzero records: {nonzero}")
    return overlap_info
