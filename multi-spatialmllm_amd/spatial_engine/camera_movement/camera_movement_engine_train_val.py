"""Mirror of the reference's camera_movement/camera_movement_engine_train_val.py: per-row record builder
with the reference's signature on top of ``mspa.heads`` (K4 for the relative pose)."""
from __future__ import annotations

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
