"""Mirror of the reference's utils/scannet_utils/make_visibility_info.py (per-scene visibility index)."""
from __future__ import annotations


def process_scene(scene_id, scene_infos, warning_file):
    """(scene_id, {"image_to_points": {img: [vertex...]}, "point_to_images": {vertex: [img...]}})
    (reference: :75-125), from K1's masks."""
    print(f"[process_scene] Start: {scene_id}")
    scene = scene_infos.scene_on_device(scene_id)
    result = scene.visibility_index()
    for image_id, pts in result["image_to_points"].items():
        if len(pts) == 0:
            with open(warning_file, "a") as f:
                f.write(f"[Warning] {scene_id}: {image_id} has no in-bound points.\n")
    print(f"[process_scene] Done: {scene_id}")
    return scene_id, result
