"""Mirror of the reference's depth_perception/depth_estimation_dot_engine.py: the depth question about a point that
is marked with a coloured disc on a copy of the frame instead of being named by coordinates."""
from __future__ import annotations

import random

import numpy

from mspa import heads
from mspa import templates as T
from mspa.annotate import Mark
from spatial_engine.depth_perception._coor_base import DepthCoorEngineBase, _LazyCounts, run_cli

numpy.random.seed(5)
random.seed(5)


class DepthEstimationDotQAEngine(DepthCoorEngineBase):
    task_name = "depth_estimation_dot"
    TEMPLATE_SET = T.DEPTH_ESTIMATION_DOT

    def _mark(self, scene_id, image_id, vertex, pixel, color):
        self._annotator().annotate(self.scene_info.get_image_path(scene_id, image_id),
                                   self._annotated_path(scene_id, f"{image_id}_p{vertex}_annotated.jpg"),
                                   [Mark(pixel[0], pixel[1], 10, color)])

    DRAWS_AHEAD = True

    def _scene_draws(self, scene_id):
        """The scene's random decisions (images, vertex positions, templates, disc colours): they read the visibility index only."""
        return heads.depth_estimation_draws(self.scene_info.get_all_extrinsic_valid_image_ids(scene_id),
                                            _LazyCounts(self._visible_points_of(scene_id)), self.max_samples, self.templates, random,
                                            self.max_n_points_per_image, True)

    def _scene_records_on(self, scene, scene_id, draws):
        image_ids, n_visible, numeric_fn, image_hw = self._scene_inputs(scene_id, scene)
        return heads.depth_estimation_records_fn(
            scene_id, image_ids, n_visible, numeric_fn, image_hw, self.max_samples, self.templates, random,
            self.max_n_points_per_image, dot=True, on_mark=self._mark, draws=draws,
            on_skip=lambda s, img, v: self._warn(f"Warning: Point-Id {v[0]} is not visible in image {img} in scene {s}.\n"))

    def generate_qa_training_single_scene(self, scene_id):
        return self._scene_records_on(None, scene_id, self._scene_draws(scene_id))


if __name__ == "__main__":
    run_cli(DepthEstimationDotQAEngine, "depth_estimation_dot")
