"""Mirror of the reference's visual_correspondence_qa_engine_coor_2_coor.py record builder."""
from __future__ import annotations

import random

import numpy as np

from mspa import heads
from mspa import templates as T
from mspa.sampling import sample_dataframe  # noqa: F401

TEMPLATE_SET = T.VISUAL_CORRESPONDENCE


def build_training_sample(scene_infos, row, idx: int, visibility_info_dict, warning_file, max_points_per_pair=1):
    """One coordinate-to-coordinate correspondence record (reference: :264-394).  The visible-vertex lists
    come from ``visibility_info_dict`` as upstream; the two projections run on the GPU."""
    scene_id = row["scene_id"]

    def warn(message):
        print(message.strip())
        with open(warning_file, "a") as wf:
            wf.write(message)

    if scene_id not in visibility_info_dict:
        random.random()                     # upstream draws its swap coin before this check (:280)
        warn(f"[build_training_sample] Warning: Visibility info not found for scene {scene_id}\n")
        return None
    i2p = visibility_info_dict[scene_id].get("image_to_points", {})
    common = np.intersect1d(i2p.get(row["image_id1"], []), i2p.get(row["image_id2"], []))
    draw = heads.visual_correspondence_draws([row], [len(common)], TEMPLATE_SET, random, max_points_per_pair)[0]
    if draw is None:
        warn(f"[build_training_sample] Warning: No common visible points for scene {scene_id} "
             f"{row['image_id1']}, {row['image_id2']}\n")
        return None
    first, second = (row["image_id2"], row["image_id1"]) if draw["swap"] else (row["image_id1"], row["image_id2"])
    uv1, uv2 = [], []
    for j in draw["positions"]:
        v = int(common[j])
        a = scene_infos.get_point_2d_coordinates_in_image(scene_id, first, v, align=True, check_visible=True)
        b = scene_infos.get_point_2d_coordinates_in_image(scene_id, second, v, align=True, check_visible=True)
        if len(a) == 0 or len(b) == 0:
            warn(f"Warning: Point {v} is not visible in image {first if len(a) == 0 else second} in scene {scene_id}.\n")
            return None
        uv1.append(a[0])
        uv2.append(b[0])
    return heads.visual_correspondence_record(row, idx, draw, np.stack(uv1), np.stack(uv2),
                                              scene_infos.get_image_shape(scene_id), TEMPLATE_SET)


convert_train_sample_to_eval_sample = heads.to_eval_sample
