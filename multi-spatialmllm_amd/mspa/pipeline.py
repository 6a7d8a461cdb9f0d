"""End-to-end regeneration pipeline: posed RGB-D scenes -> pair table -> sampled QA records -> JSONL.

One process per GPU.  Scenes are assigned to ranks longest-processing-time-first (``shard.lpt_assign``);
every rank runs the geometry of its scenes on its GPU (K1 visibility, K2 overlap, K4 pose for the pair
table; K6/K4 for the heads) with **no collective on the data path**.  Two exchange steps, both
collations:
  1. the numeric pair tables (fixed-width rows: scene, image1, image2, overlap, distance, yaw, pitch)
     are all-gathered (``shard.collate_records``) so that every rank sees the same global table and the
     overlap-binned sampling (``sampling.sample_dataframe``, seeded) picks the same rows everywhere --
     exactly what a single process would pick;
  2. the finished QA records are gathered to rank 0 (``all_gather_object`` over RCCL), shuffled with the
     head's seed and written as JSONL.
This is BASELINE.json configs[4] in miniature: camera movement, visual correspondence, depth estimation / comparison and
object perception from the posed RGB-D scenes, object movement from TAPVid-style track blocks (``tracks=``).  Scenes and
tracks come in as arrays (``mspa.synth``, ``mspa.sens`` or the façade handler).

    python -m mspa.pipeline --scenes 4 --frames 12 --out /tmp/mspa_out            # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mspa.pipeline ...
"""
from __future__ import annotations

import argparse
import os
import random
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import heads, sampling, shard
from . import templates as T


def pair_table_rows(scene_idx: int, scene) -> np.ndarray:
    """[n_pairs, 7] float64 rows (scene, i, j, overlap, distance, yaw, pitch) of one resident scene."""
    import torch
    from . import engine
    F = len(scene.ids)
    if F < 2:
        return torch.zeros((0, 7), dtype=torch.float64, device=scene.device)
    pairs = engine.all_pairs(F, scene.device)
    vis = scene._visibility()
    overlap = engine.scene_overlap(vis["bits"])
    pose = engine.pair_pose(*scene.pose_tables(), pairs)
    out = torch.empty((pairs.shape[0], 7), dtype=torch.float64, device=scene.device)
    out[:, 0] = scene_idx
    out[:, 1:3] = pairs.to(torch.float64)
    out[:, 3] = overlap
    out[:, 4:7] = pose[:, 0:3]
    return out


def run(scenes: Sequence, out_dir: str, ctx: Optional[shard.DistContext] = None, device="cuda", seed: int = 0,
        n_camera: int = 64, n_correspondence: int = 64, depth_images_per_scene: int = 4,
        overlap_range=(6, 35), question_types: Sequence[str] = ("total_distance", "displacement_vector"),
        object_perception: bool = True, tracks: Sequence = ()) -> Dict[str, int]:
    """Run the ScanNet-side heads over ``scenes`` (objects with K, A, E, depth, points, color_hw, scene_id).
    Returns {jsonl name: record count} on rank 0 (empty dict elsewhere)."""
    import pandas as pd
    import torch
    import torch.distributed as dist
    from .scene import SceneOnDevice

    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    costs = [shard.scene_cost(len(sc.valid_image_ids), sc.points.shape[0]) for sc in scenes]
    mine = shard.lpt_assign(costs, world)[rank]

    # ---- geometry of my scenes; pair table collation --------------------------------------------
    resident = {k: SceneOnDevice(scenes[k].K, scenes[k].A, scenes[k].E, scenes[k].depth, scenes[k].color_hw,
                                 scenes[k].points, device) for k in mine}
    local = [pair_table_rows(k, resident[k]) for k in mine]
    local = torch.cat(local, 0) if local else torch.zeros((0, 7), dtype=torch.float64, device=device)
    table = shard.collate_records(local, ctx) if ctx is not None else local
    table = table.cpu().numpy()
    order = np.lexsort((table[:, 2], table[:, 1], table[:, 0]))          # rank-independent row order
    table = table[order]
    ids = {k: scenes[k].valid_image_ids for k in range(len(scenes))}
    df = pd.DataFrame({
        "scene_id": [scenes[int(s)].scene_id for s in table[:, 0]],
        "image_id1": [ids[int(s)][int(i)] for s, i in zip(table[:, 0], table[:, 1])],
        "image_id2": [ids[int(s)][int(j)] for s, j in zip(table[:, 0], table[:, 2])],
        "overlap": table[:, 3], "distance": table[:, 4], "yaw": table[:, 5], "pitch": table[:, 6],
        "_scene": table[:, 0].astype(int),
    })

    outputs: Dict[str, List[dict]] = {}

    def my_rows(sampled):
        return [(k, sampled.iloc[k].to_dict()) for k in range(len(sampled)) if int(sampled.iloc[k]["_scene"]) in resident]

    # ---- camera movement (seed as upstream: CME:17-18) -------------------------------------------
    for qt in question_types:
        np.random.seed(seed)
        random.seed(seed)
        sampled = sampling.sample_dataframe(df, n_camera, 0, overlap_range[0], overlap_range[1], 1)
        recs = []
        by_scene: Dict[int, List] = {}
        for k, row in my_rows(sampled):
            by_scene.setdefault(int(row["_scene"]), []).append((k, row))
        for s, items in by_scene.items():
            rng = random.Random(f"{seed}:{qt}:{s}")             # per-scene stream: result independent of the sharding
            t12, t21 = heads.camera_movement_numeric(resident[s], [r for _, r in items])
            for n, (k, row) in enumerate(items):
                recs.append(heads.camera_movement_record(row, k, qt, t12[n], t21[n], scenes[s].color_hw,
                                                         T.CAMERA_MOVEMENT, rng))
        outputs[f"camera_movement_{qt}"] = recs

    # ---- visual correspondence (VC_C:11-12 seeds 1) ------------------------------------------------
    np.random.seed(seed + 1)
    sampled = sampling.sample_dataframe(df, n_correspondence, 0, overlap_range[0], overlap_range[1], 1)
    recs = []
    by_scene = {}
    for k, row in my_rows(sampled):
        by_scene.setdefault(int(row["_scene"]), []).append((k, row))
    for s, items in by_scene.items():
        rng = random.Random(f"{seed}:vc:{s}")
        rows = [r for _, r in items]
        got = heads.visual_correspondence_records(resident[s], rows, scenes[s].color_hw, 0, T.VISUAL_CORRESPONDENCE, rng)
        recs.extend(got)
    outputs["visual_correspondence_coor_2_coor"] = recs

    # ---- depth estimation ------------------------------------------------------------------------
    recs = []
    for s in mine:
        rng = random.Random(f"{seed}:depth:{s}")
        recs.extend(heads.depth_estimation_records(resident[s], scenes[s].scene_id, scenes[s].color_hw,
                                                   depth_images_per_scene, T.DEPTH_ESTIMATION, rng))
    outputs["depth_estimation_coor"] = recs
    recs = []
    for s in mine:
        rng = random.Random(f"{seed}:depthcmp:{s}")
        recs.extend(heads.depth_comparison_records_gpu(resident[s], scenes[s].scene_id, scenes[s].color_hw,
                                                       depth_images_per_scene, T.DEPTH_COMPARISON, rng))
    outputs["depth_comparison_coor"] = recs

    # ---- object perception: visibility + coverage + records, per scene (COVIS / COV / OPE) ----------
    if object_perception:
        by_name: Dict[str, List[dict]] = {}
        for s in mine:
            if not hasattr(scenes[s], "objects"):
                continue
            idx, bbox, cat = scenes[s].objects()
            rng = random.Random(f"{seed}:op:{s}")
            cov, _ = resident[s].object_coverage(idx, bbox, rng=rng)
            for d, dim in enumerate(("height", "length", "width")):
                table = {scenes[s].scene_id: {o: res[dim] for o, res in cov.items()}}
                size = {"height": lambda o: bbox[o][5], "length": lambda o: max(bbox[o][3], bbox[o][4]),
                        "width": lambda o: min(bbox[o][3], bbox[o][4])}[dim]
                by_k = heads.object_perception_records(table, dim, lambda _s, o: size(o), lambda _s, o: cat[o],
                                                       scenes[s].color_hw, 6, T.OBJECT_PERCEPTION, rng)
                for k, recs in by_k.items():
                    if recs:
                        by_name.setdefault(f"object_perception_{dim}_k{k}", []).extend(recs)
        if ctx is not None:                       # every rank must enter the same collations: agree on the names
            names: List[Optional[list]] = [None] * world
            dist.all_gather_object(names, sorted(by_name), group=ctx.group)
            for n in sorted({x for part in names for x in part}):
                by_name.setdefault(n, [])
        outputs.update(by_name)

    # ---- object movement on TAPVid-style track blocks (OM_C): blocks sharded like scenes -----------------------
    if tracks:
        from scipy.cluster.hierarchy import fcluster, linkage
        from scipy.spatial.distance import squareform
        from . import engine
        mine_tr = shard.lpt_assign([float(t.tracks_XYZ.shape[0]) * t.tracks_XYZ.shape[1] ** 2 for t in tracks], world)[rank]
        for qt in T.OBJECT_MOVEMENT_TYPES:
            recs = []
            for k in mine_tr:
                tr = tracks[k]
                rng = random.Random(f"{seed}:om:{qt}:{k}")
                xyz = np.ascontiguousarray(tr.tracks_XYZ, dtype=np.float64)
                dev_tracks = torch.from_numpy(xyz).to(device)
                loss = engine.track_rigidity_loss(dev_tracks).cpu().numpy()                       # K7
                labels = fcluster(linkage(squareform(loss, checks=False), method="average"), 0.1, criterion="distance")
                groups = [g for g in (np.where(labels == i)[0].tolist() for i in range(1, max(labels) + 1)) if len(g) > 5]
                c2w = torch.from_numpy(np.linalg.inv(tr.extrinsics_w2c).reshape(-1, 16)).to(device)
                world_xyz = engine.track_to_world(dev_tracks, c2w, tr.fx_fy_cx_cy, tr.image_hw, ("world",))["world"]   # K5a
                pairs_k = heads.object_movement_mine_pairs(                                       # K5c
                    tr.visibility, groups, lambda p, f: engine.track_pair_distances(world_xyz, p, f), 5, 3, True, 0.05, rng)
                recs.extend(heads.object_movement_records(tr.scene_id, xyz, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                                          pairs_k, qt, T.OBJECT_MOVEMENT, rng, device))   # K5a + K5b
            outputs[f"object_movement_{qt}"] = recs

    # ---- collation of the finished records + JSONL --------------------------------------------------
    counts: Dict[str, int] = {}
    os.makedirs(out_dir, exist_ok=True)
    for name in sorted(outputs):
        local_recs = outputs[name]
        if ctx is not None:
            gathered: List[Optional[list]] = [None] * world
            dist.all_gather_object(gathered, local_recs, group=ctx.group)      # RCCL (or gloo) under the hood
            allrecs = [r for part in gathered for r in part]
        else:
            allrecs = list(local_recs)
        if rank == 0:
            allrecs.sort(key=lambda r: str(r["id"]))            # canonical order first: shuffle is sharding-independent
            random.Random(f"{seed}:{name}").shuffle(allrecs)
            heads.write_jsonl(os.path.join(out_dir, f"{name}.jsonl"), allrecs)
            counts[name] = len(allrecs)
    return counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--tracks", type=int, default=2, help="TAPVid-style track blocks for the object-movement family")
    ap.add_argument("--out", default="mspa_out")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    from . import synth
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("mspa.pipeline needs a ROCm GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ctx = shard.init_distributed(device) if world > 1 else None
    scenes = [synth.make_scene(7000 + k, n_points=args.points, n_frames=args.frames + 3 * (k % 3),
                               color_hw=(480, 640), with_color=False) for k in range(args.scenes)]
    tracks = [synth.make_tracks(300 + k, T=120, P=96, n_groups=4) for k in range(args.tracks)]
    counts = run(scenes, args.out, ctx, device, args.seed, tracks=tracks)
    if ctx is None or ctx.rank == 0:
        print({"out": args.out, "records": counts})
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
