"""Mirror of the reference's depth_perception/depth_estimation_coor_engine.py: "how far is the point at
[x, y]" records.  Image / vertex / template draws follow upstream's order; projection and the visibility
re-check of the drawn vertices run on the GPU in one batch per scene (K6b)."""
from __future__ import annotations

import random

import numpy

from mspa import heads
from mspa import templates as T
from spatial_engine.depth_perception._coor_base import DepthCoorEngineBase, _LazyCounts, run_cli

numpy.random.seed(4)
random.seed(4)


class DepthEstimationCoorQAEngine(DepthCoorEngineBase):
    task_name = "depth_estimation_coor"
    TEMPLATE_SET = T.DEPTH_ESTIMATION

    DRAWS_AHEAD = True

    def _scene_draws(self, scene_id):
        """The scene's random decisions (images, vertex positions, templates): they read the visibility index only."""
        return heads.depth_estimation_draws(self.scene_info.get_all_extrinsic_valid_image_ids(scene_id),
                                            _LazyCounts(self._visible_points_of(scene_id)), self.max_samples, self.templates, random,
                                            self.max_n_points_per_image, False)

    def _scene_records_on(self, scene, scene_id, draws):
        image_ids, n_visible, numeric_fn, image_hw = self._scene_inputs(scene_id, scene)
        return heads.depth_estimation_records_fn(
            scene_id, image_ids, n_visible, numeric_fn, image_hw, self.max_samples, self.templates, random,
            self.max_n_points_per_image, draws=draws,
            on_skip=lambda s, img, v: self._warn(f"Warning: Point-Id {v[0]} is not visible in image {img} in scene {s}.\n"))

    def generate_qa_training_single_scene(self, scene_id):
        return self._scene_records_on(None, scene_id, self._scene_draws(scene_id))


if __name__ == "__main__":
    run_cli(DepthEstimationCoorQAEngine, "depth_estimation_coor")
