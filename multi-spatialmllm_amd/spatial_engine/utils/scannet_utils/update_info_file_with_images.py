"""Mirror of the reference's utils/scannet_utils/update_info_file_with_images.py as a function (upstream is a script with
its paths at module level): add num_posed_images / images_info / intrinsic_matrix of every scene to the scene-info pickle,
keeping every ``frame_skip``-th exported image, from the posed_images folders -- or straight from ``.sens`` streams."""
from __future__ import annotations

import os
import pickle

import numpy as np


def _parse(path):
    with open(path) as f:
        return np.array([list(map(float, line.split())) for line in f.readlines()])


def scene_entries_from_folder(base_dir, scene_id, frame_skip=5):
    """What upstream's loop body builds for one scene (reference: :20-68)."""
    scene_path = os.path.join(base_dir, scene_id)
    jpgs = sorted(f for f in os.listdir(scene_path) if f.endswith(".jpg"))
    images = {}
    for i, filename in enumerate(jpgs):
        if i % frame_skip == 0:
            image_id = filename.split(".")[0]
            images[image_id] = {"image_path": f"posed_images/{scene_id}/{filename}",
                                "depth_image_path": f"posed_images/{scene_id}/{image_id}.png",
                                "extrinsic_matrix": _parse(os.path.join(scene_path, f"{image_id}.txt"))}
    return {"num_posed_images": len(images), "images_info": images,
            "intrinsic_matrix": _parse(os.path.join(scene_path, "intrinsic.txt"))}


def update_info_file(scene_infos_file="data/scannet/scannet_instance_data/scenes_train_val_info.pkl",
                     base_dir="data/scannet/posed_images", frame_skip=5, sens_root=None):
    """Writes ``<scene_infos_file minus .pkl>_i_D{frame_skip}.pkl`` and returns its path.  With ``sens_root`` the entries
    come from ``<sens_root>/<scene>/<scene>.sens`` (mspa.sens: same numbers, no intermediate files)."""
    with open(scene_infos_file, "rb") as f:
        scene_infos = pickle.load(f)
    for scene_id in scene_infos:
        if sens_root is not None:
            from mspa import sens
            # headers and poses of the kept frames only: no depth payload is inflated for the info file
            stream = sens.read_sens(os.path.join(sens_root, scene_id, f"{scene_id}.sens"), keep_every=frame_skip,
                                    want_depth=False)
            entries = sens.scene_info_entries(scene_id, stream, frame_skip)
        else:
            entries = scene_entries_from_folder(base_dir, scene_id, frame_skip)
        scene_infos[scene_id].update(entries)
    out = scene_infos_file.replace(".pkl", f"_i_D{frame_skip}.pkl")
    with open(out, "wb") as f:
        pickle.dump(scene_infos, f)
    return out


if __name__ == "__main__":
    print(update_info_file())
