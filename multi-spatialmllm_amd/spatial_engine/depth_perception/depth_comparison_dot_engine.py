"""Mirror of the reference's depth_perception/depth_comparison_dot_engine.py: "which lettered dot is closer / farther";
a pair whose depths tie is re-drawn up to ten more times, as upstream."""
from __future__ import annotations

import random

import numpy

from mspa import heads
from mspa import templates as T
from mspa.annotate import Mark
from spatial_engine.depth_perception._coor_base import DepthCoorEngineBase, run_cli

numpy.random.seed(6)
random.seed(6)


class DepthComparisonDotQAEngine(DepthCoorEngineBase):
    task_name = "depth_comparison_dot"
    TEMPLATE_SET = T.DEPTH_COMPARISON_DOT

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.max_n_points_per_image == 1, "single-round QA only: one pair per image (as upstream)"

    def _mark(self, scene_id, image_id, vertices, points_info, colors):
        marks = [Mark(p["coords"][0], p["coords"][1], 10, c, p["letter"], (15, 15)) for p, c in zip(points_info, colors)]
        self._annotator().annotate(self.scene_info.get_image_path(scene_id, image_id),
                                   self._annotated_path(scene_id, f"{image_id}_p{vertices[0]}_p{vertices[1]}_annotated.jpg"), marks)

    def generate_qa_training_single_scene(self, scene_id):
        image_ids, n_visible, numeric_fn, image_hw = self._scene_inputs(scene_id)
        return heads.depth_comparison_records(
            scene_id, image_ids, n_visible, numeric_fn, image_hw, self.max_samples, self.templates, random,
            self.max_n_points_per_image, dot=True, on_mark=self._mark,
            on_skip=lambda s, img, v: self._warn(f"Warning: Points {v} in image {img} in scene {s} have the same depth.\n"
                                                 " Skip this pair."))


if __name__ == "__main__":
    run_cli(DepthComparisonDotQAEngine, "depth_comparison_dot")
