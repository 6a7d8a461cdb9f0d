"""Mirror of the reference's depth_perception/depth_comparison_coor_engine.py: "which of the two points is
closer / farther" records.  Pairs whose depths round to the same millimetre are skipped before their
templates are drawn, exactly as upstream (see mspa.heads.depth_comparison_records for how the batch on the GPU
and the sequential ``random`` stream are reconciled)."""
from __future__ import annotations

import random

import numpy

from mspa import heads
from mspa import templates as T
from spatial_engine.depth_perception._coor_base import DepthCoorEngineBase, run_cli

numpy.random.seed(7)
random.seed(7)


class DepthComparisonCoorQAEngine(DepthCoorEngineBase):
    task_name = "depth_comparison_coor"
    TEMPLATE_SET = T.DEPTH_COMPARISON

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.max_n_points_per_image == 1, "single-round QA only: one pair per image (as upstream)"

    def generate_qa_training_single_scene(self, scene_id):
        image_ids, n_visible, numeric_fn, image_hw = self._scene_inputs(scene_id)
        return heads.depth_comparison_records(
            scene_id, image_ids, n_visible, numeric_fn, image_hw, self.max_samples, self.templates, random,
            self.max_n_points_per_image,
            on_skip=lambda s, img, v: self._warn(f"Warning: Points {v} in image {img} in scene {s} have the same depth.\n"
                                                 " Skip this pair."))


if __name__ == "__main__":
    run_cli(DepthComparisonCoorQAEngine, "depth_comparison_coor")
