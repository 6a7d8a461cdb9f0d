"""Mirror of the reference's camera_movement/calculate_frames_relations.py (per-scene pair table)."""
from __future__ import annotations

import numpy as np
import torch

from mspa import engine


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
