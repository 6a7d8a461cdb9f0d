"""Mirror of the reference's depth_perception/depth_comparison_dot_engine.py: "which lettered dot is closer / farther";
a pair whose depths tie is re-drawn up to ten more times, as upstream."""
from __future__ import annotations

import random

import numpy

from mspa import heads
from mspa import templates as T
from mspa.annotate import Mark
from spatial_engine.depth_perception._coor_base import DepthCoorEngineBase, _LazyCounts, run_cli

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
            self.max_n_points_per_image, dot=True, on_mark=self._mark, dry_run=dry_run,
            on_skip=lambda s, img, v: self._warn(f"Warning: Points {v} in image {img} in scene {s} have the same depth.\n"
                                                 " Skip this pair."))

    def generate_qa_training_single_scene(self, scene_id):
        return self._scene_records_on(None, scene_id, None)


if __name__ == "__main__":
    run_cli(DepthComparisonDotQAEngine, "depth_comparison_dot")
