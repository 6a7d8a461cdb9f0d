"""Mirror of the reference's depth_perception/depth_comparison_coor_engine.py: "which of the two points is
closer / farther" records.  Pairs whose depths round to the same millimetre are skipped before their
templates are drawn, exactly as upstream (see mspa.heads.depth_comparison_records for how the batch on the GPU
and the sequential ``random`` stream are reconciled)."""
from __future__ import annotations

import random

import numpy

from mspa import heads
from mspa import templates as T
from spatial_engine.depth_perception._coor_base import DepthCoorEngineBase, _LazyCounts, run_cli

numpy.random.seed(7)
random.seed(7)


class DepthComparisonCoorQAEngine(DepthCoorEngineBase):
    task_name = "depth_comparison_coor"
    TEMPLATE_SET = T.DEPTH_COMPARISON

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.max_n_points_per_image == 1, "single-round QA only: one pair per image (as upstream)"

    CHAINED = True

    def _scene_records_on(self, scene, scene_id, _draws, dry_run=False):
        """The scene's records, the generator advanced as upstream advances it; ``dry_run``: only the draws of a scene without a
        skipped pair (host side: the visibility index is all they read)."""
        if dry_run:
            image_ids, n_visible = self.scene_info.get_all_extrinsic_valid_image_ids(scene_id), _LazyCounts(self._visible_points_of(scene_id))
            numeric_fn, image_hw = None, self.scene_info.get_image_shape(scene_id)
        else:
            image_ids, n_visible, numeric_fn, image_hw = self._scene_inputs(scene_id, scene)
        return heads.depth_comparison_records(
            scene_id, image_ids, n_visible, numeric_fn, image_hw, self.max_samples, self.templates, random,
            self.max_n_points_per_image, dry_run=dry_run,
            on_skip=lambda s, img, v: self._warn(f"Warning: Points {v} in image {img} in scene {s} have the same depth.\n"
                                                 " Skip this pair."))

    def generate_qa_training_single_scene(self, scene_id):
        return self._scene_records_on(None, scene_id, None)


if __name__ == "__main__":
    run_cli(DepthComparisonCoorQAEngine, "depth_comparison_coor")
