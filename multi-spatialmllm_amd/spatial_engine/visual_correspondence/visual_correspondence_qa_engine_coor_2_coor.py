"""Mirror of the reference's visual_correspondence_qa_engine_coor_2_coor.py record builder."""
from __future__ import annotations

import json
import os
import random

import numpy as np
import torch

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
    state = random.getstate()
    draw = heads.visual_correspondence_draws([row], [len(common)], TEMPLATE_SET, random, max_points_per_pair)[0]
    if draw is None:
        warn(f"[build_training_sample] Warning: No common visible points for scene {scene_id} "
             f"{row['image_id1']}, {row['image_id2']}\n")
        return None
    first, second = (row["image_id2"], row["image_id1"]) if draw["swap"] else (row["image_id1"], row["image_id2"])
    uv1, uv2, hidden = [], [], set()
    for s_idx, j in enumerate(draw["positions"]):
        v = int(common[j])
        a = scene_infos.get_point_2d_coordinates_in_image(scene_id, first, v, align=True, check_visible=True)
        b = scene_infos.get_point_2d_coordinates_in_image(scene_id, second, v, align=True, check_visible=True)
        if len(a) == 0 or len(b) == 0:                       # only with a stale index; upstream draws no template for it
            if len(a) == 0:
                warn(f"Warning: Point {v} is not visible in image {first} in scene {scene_id}.\n")
            if len(b) == 0:
                warn(f"Warning: Point {v} is not visible in image {second} in scene {scene_id}.\n")
            hidden.add(s_idx)
            a = b = np.zeros((1, 2))
        uv1.append(a[0])
        uv2.append(b[0])
    if hidden:                                               # redo this row's draws without templates for those slots
        random.setstate(state)
        draw = heads.visual_correspondence_draws([row], [len(common)], TEMPLATE_SET, random, max_points_per_pair, [hidden])[0]
        if all(p is None for p in draw["picks"]):
            warn(f"[build_training_sample] Warning: No conversation for scene {scene_id} {first}, {second}\n")
            return None
    return heads.visual_correspondence_record(row, idx, draw, np.stack(uv1), np.stack(uv2),
                                              scene_infos.get_image_shape(scene_id), TEMPLATE_SET)


convert_train_sample_to_eval_sample = heads.to_eval_sample


def convert_parquet_to_dict(parquet_df):
    """key -> parsed list, as upstream (:247-262)."""
    return dict(zip(parquet_df["key"].tolist(), [json.loads(v) for v in parquet_df["values"]]))


def _load_visibility(visibility_info_path):
    print(f"Loading {visibility_info_path}.")
    if visibility_info_path.endswith(".parquet"):
        import pandas as pd
        print("Converting to dict.")
        return convert_parquet_to_dict(pd.read_parquet(visibility_info_path))
    from spatial_engine.utils.scannet_utils.handler.info_handler import _load_any
    return _load_any(visibility_info_path)


class _ResidentScenes:
    """Scenes uploaded on demand and kept while they fit a byte budget (a 320-frame scene is ~200 MB of depth)."""

    def __init__(self, scene_infos, visibility_info_dict, budget_bytes=96 << 30):
        self.scene_infos, self.vis, self.budget = scene_infos, visibility_info_dict, budget_bytes
        self.scenes, self.bits, self.used = {}, {}, 0

    def get(self, scene_id):
        if scene_id not in self.vis:                          # upstream's check (:282-287)
            return None
        if scene_id not in self.scenes:
            scene = self.scene_infos.scene_on_device(scene_id)
            size = scene.depth.numel() * 2 + scene.xyz.numel() * 8
            while self.scenes and self.used + size > self.budget:
                old, victim = next(iter(self.scenes.items()))
                self.used -= victim.depth.numel() * 2 + victim.xyz.numel() * 8
                del self.scenes[old]
                self.bits.pop(old, None)
            self.scenes[scene_id] = scene
            self.used += size
        return self.scenes[scene_id]

    def get_bits(self, scene_id, scene):
        """The scene's visibility rows from the visibility file (what upstream intersects), as GPU bitsets."""
        if scene_id not in self.bits:
            from mspa.scene import pack_index_lists
            i2p = self.vis[scene_id].get("image_to_points", {})
            self.bits[scene_id] = torch.from_numpy(pack_index_lists([i2p.get(i, []) for i in scene.ids],
                                                                    scene.xyz.shape[0])).to(scene.device)
        return self.bits[scene_id]


def _build_samples(parquet_path, scene_infos, desired_count, overlap_min, overlap_max, interval, visibility_info_path,
                   warning_file, max_points_per_pair, tag, transform=None):
    """(records in row order -- rows without one dropped --, how many there are on every rank, communicator or None).  In a job
    with one process per GPU the scenes of the sampled rows are dealt over the ranks: each reads and uploads only its own
    (``heads.visual_correspondence_dataset``); upstream loops over the rows in one process (VC_C:424-429)."""
    import pandas as pd
    from mspa import shard
    ctx = shard.context_from_env()
    df = pd.read_parquet(parquet_path)
    print(f"[{tag}] Loaded DataFrame with {len(df)} rows from {parquet_path}")
    print(f"[{tag}] Sampling {desired_count} samples with overlap in [{overlap_min}, {overlap_max}]")
    df_sampled = sample_dataframe(df, all_overlap_samples=desired_count, non_overlap_samples=0, overlap_min=overlap_min,
                                  overlap_max=overlap_max, interval=interval)
    print(f"[{tag}] Got {len(df_sampled)} sampled rows")
    resident = _ResidentScenes(scene_infos, _load_visibility(visibility_info_path))

    def warn(message):
        print(message.strip())
        with open(warning_file, "a") as wf:
            wf.write(message)
    rows = df_sampled.to_dict("records")
    samples = heads.visual_correspondence_dataset(rows, resident.get, resident.get_bits, TEMPLATE_SET, random,
                                                  max_points_per_pair, warn, ctx=ctx, transform=transform)
    if ctx is None:
        kept = [s for s in samples if s]
        return kept, len(kept), None
    import torch.distributed as dist                          # every rank shuffles an index list of rank 0's length (the generator stays in step)
    n_kept = torch.tensor([sum(1 for s in samples if s)], dtype=torch.int64, device=ctx.collective_device)
    dist.broadcast(n_kept, src=0, group=ctx.group)
    return [s for s in samples if s], int(n_kept.item()), ctx


def _shuffle_and_write(samples, n_kept, ctx, out_file, tag):
    """``random.shuffle(out_samples)`` + the JSONL (VC_C:430-433), the shuffle as an index permutation every rank computes."""
    order = list(range(n_kept))
    random.shuffle(order)
    if ctx is not None and ctx.rank != 0:
        ctx.barrier()
        return
    print(f"[{tag}] Writing {len(samples)} items to {out_file}")
    heads.write_jsonl(out_file, [samples[i] for i in order])
    if ctx is not None:
        ctx.barrier()


def build_train_dataset(parquet_path, output_dir, scene_infos, desired_count, overlap_min, overlap_max, interval,
                        visibility_info_path, warning_file, max_points_per_pair=1):
    """train_visual_correspondence_coor_2_coor.jsonl (reference: :401-433); all projections batched per scene."""
    samples, n, ctx = _build_samples(parquet_path, scene_infos, desired_count, overlap_min, overlap_max, interval,
                                     visibility_info_path, warning_file, max_points_per_pair, "Train")
    _shuffle_and_write(samples, n, ctx, os.path.join(output_dir, "train_visual_correspondence_coor_2_coor.jsonl"), "Train")


def build_val_dataset(parquet_path, output_dir, scene_infos, desired_count, overlap_min, overlap_max, interval,
                      visibility_info_path, warning_file, max_points_per_pair=1):
    assert max_points_per_pair == 1, "[Val] max_points_per_pair should be 1."
    samples, n, ctx = _build_samples(parquet_path, scene_infos, desired_count, overlap_min, overlap_max, interval,
                                     visibility_info_path, warning_file, max_points_per_pair, "Val",
                                     transform=convert_train_sample_to_eval_sample)
    _shuffle_and_write(samples, n, ctx, os.path.join(output_dir, "val_visual_correspondence_coor_2_coor.jsonl"), "Val")


DEBUG = False
USE_PICKLE = True


def main():
    """Same paths, budgets and seeds as upstream's main (:474-538)."""
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    random.seed(1)
    np.random.seed(1)
    info_path = "data/scannet/scannet_instance_data/scenes_train_val_info_i_D5.pkl"
    ext = "pkl" if USE_PICKLE else "parquet"
    train_vis = f"data/scannet/scannet_instance_data/train_visibility_info_D5.{ext}"
    val_vis = f"data/scannet/scannet_instance_data/val_visibility_info_D5.{ext}"
    overlap_min, overlap_max, interval, version = 6, 35, 1, "v1_0"
    if DEBUG:
        train_parquet = "training_data/camera_movement/train_camera_info_D5_debug_nonzero.parquet"
        val_parquet = "evaluation_data/camera_movement/val_camera_info_D5_debug_nonzero.parquet"
        train_max, val_max = 100, 100
        version += "_debug"
    else:
        train_parquet = "training_data/camera_movement/train_camera_info_D5.parquet"
        val_parquet = "evaluation_data/camera_movement/val_camera_info_D5.parquet"
        train_max, val_max = 1000000, 300
    train_dir = os.path.join("training_data/visual_correspondence_coor_2_coor", version)
    val_dir = os.path.join("evaluation_data/visual_correspondence_coor_2_coor", version)
    os.makedirs(train_dir, exist_ok=True)
    os.makedirs(val_dir, exist_ok=True)
    scene_infos = SceneInfoHandler(info_path)
    build_val_dataset(val_parquet, val_dir, scene_infos, val_max, overlap_min, overlap_max, interval, val_vis,
                      os.path.join(val_dir, "val_warning.txt"), max_points_per_pair=1)
    build_train_dataset(train_parquet, train_dir, scene_infos, train_max, overlap_min, overlap_max, interval, train_vis,
                        os.path.join(train_dir, "train_warning.txt"), max_points_per_pair=1)


if __name__ == "__main__":
    main()
