"""Mirror of the reference's single_object_movement_engine_coord.py geometry entry points."""
from __future__ import annotations

import random

import numpy as np
import torch

from mspa import engine, heads
from mspa import templates as T


def rigid_body_segmentation(points, threshold=0.1, smoothing_factor=0.01):
    """Groups of track indices that move rigidly together (reference: :49-92): the T x P^2 distance-change
    accumulation runs on the GPU (K7), average linkage + fcluster stay with SciPy."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import squareform
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)).cuda()
    loss = engine.track_rigidity_loss(pts, smoothing_factor).cpu().numpy()
    links = linkage(squareform(loss), method="average")
    labels = fcluster(links, threshold, criterion="distance")
    return [np.where(labels == i)[0].tolist() for i in range(1, max(labels) + 1)]


def filter_large_groups(groups, min_size=5):
    return [g for g in groups if len(g) > min_size]


class TwoFrameVideoQAEngine:
    def __init__(self, question_type, sub_dataset):
        self.question_type = question_type
        self.sub_dataset = sub_dataset
        self.templates = T.OBJECT_MOVEMENT
        self.object_not_moving_threshold = 0.01
        self.camera_not_moving_threshold = 0.01
        self.future_frame_windows = 1e8

    def project_point(self, point_3d, intrinsics, image_height, image_width, id=""):
        """Normalised pinhole projection of one camera-space point, None when it falls outside the image
        or behind the camera (reference: :293-315); K5a with T = P = 1."""
        tr = torch.tensor(np.asarray(point_3d, dtype=np.float64).reshape(1, 1, 3), device="cuda")
        res = engine.track_to_world(tr, None, intrinsics, (int(image_height), int(image_width)), ("uvn", "ok"))
        if not bool(res["ok"][0, 0]):
            print(f"point {np.asarray(point_3d).tolist()} is invalid for intrinsics {list(intrinsics)}.")
            return None
        return res["uvn"][0, 0].cpu().numpy().tolist()

    def format_training_samples(self, sample_pairs, intrinsics, scene_id, points_pos_world, points_pos_cam,
                                image_height, image_width, extrinsics_w2c):
        """Records for chosen {frame1, frame2, point_index} samples (reference: :317-404); K5a + K5b.
        ``points_pos_world`` is accepted for signature compatibility -- the kernel recomputes it."""
        return heads.object_movement_records(scene_id, np.asarray(points_pos_cam), np.asarray(extrinsics_w2c), intrinsics,
                                             (int(image_height), int(image_width)), sample_pairs, self.question_type,
                                             self.templates, random)
