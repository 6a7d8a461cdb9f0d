"""End-to-end regeneration pipeline: posed RGB-D scenes -> pair table -> sampled QA records -> JSONL.

One process per GPU.  Scenes are assigned to ranks longest-processing-time-first (``shard.lpt_assign``);
every rank runs the geometry of its scenes on its GPU (K1 visibility, K2 overlap, K4 pose for the pair
table; K6/K4 for the heads) with **no collective on the data path**.  Two exchange steps, both
collations:
  1. the numeric pair tables (fixed-width rows: scene, image1, image2, overlap, distance, yaw, pitch)
     are all-gathered (``shard.collate_records``) so that every rank sees the same global table and the
     overlap-binned sampling (``sampling.sample_dataframe``, seeded) picks the same rows everywhere --
     exactly what a single process would pick;
  2. the heads' NUMERIC results -- not their text -- are collated: every rank records the arrays its K2 / K4 / K5 / K6 / K7 /
     K8 launches return while it runs the heads of its scenes (``mspa/tape.py``), the tapes (float64 rows of width 8) go
     through ``shard.collate_records``, and rank 0 replays the same heads on them to build the record text, shuffles with
     the head's seed and writes the JSONL.  No pickled record ever crosses the fabric.
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
import time
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


LAST_TIMINGS: Dict[str, float] = {}      # of the most recent run() in this process: rank 0's serial replay of the other ranks' tapes


def run(scenes: Sequence, out_dir: str, ctx: Optional[shard.DistContext] = None, device="cuda", seed: int = 0,
        n_camera: int = 64, n_correspondence: int = 64, depth_images_per_scene: int = 4,
        overlap_range=(6, 35), question_types: Sequence[str] = ("total_distance", "displacement_vector"),
        object_perception: bool = True, tracks: Sequence = (), selftest_tape: bool = False) -> Dict[str, int]:
    """Run the ScanNet-side heads over ``scenes`` (objects with K, A, E, depth, points, color_hw, scene_id).
    Returns {jsonl name: record count} on rank 0 (empty dict elsewhere)."""
    import pandas as pd
    import torch
    from .scene import SceneOnDevice

    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    LAST_TIMINGS.clear()
    costs = [shard.scene_cost(len(sc.valid_image_ids), sc.points.shape[0]) for sc in scenes]
    scene_bins = shard.lpt_assign(costs, world)
    scene_owner = {k: r for r, b in enumerate(scene_bins) for k in b}
    mine = scene_bins[rank]

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

    # ---- the heads as units of work: (collation name, unit key, fn(scene) -> records or {name: records}) ----------------
    # A unit is one scene (or one track block) of one head; its generator is seeded per unit, so what it produces does not
    # depend on which rank runs it -- nor on whether the numbers come from the kernels or from a tape (mspa/tape.py).
    from . import engine, tape
    units: Dict[str, List] = {}

    def rows_by_scene(sampled):
        by: Dict[int, List] = {}
        for k in range(len(sampled)):
            row = sampled.iloc[k].to_dict()
            by.setdefault(int(row["_scene"]), []).append((k, row))
        return by

    # camera movement (seed as upstream: CME:17-18)
    for qt in question_types:
        np.random.seed(seed)
        random.seed(seed)
        sampled = sampling.sample_dataframe(df, n_camera, 0, overlap_range[0], overlap_range[1], 1)
        for s, items in sorted(rows_by_scene(sampled).items()):
            def cm_unit(scene, s=s, items=items, qt=qt):
                rng = random.Random(f"{seed}:{qt}:{s}")             # per-scene stream: result independent of the sharding
                t12, t21 = heads.camera_movement_numeric(scene, [r for _, r in items])
                return [heads.camera_movement_record(row, k, qt, t12[n], t21[n], scenes[s].color_hw, T.CAMERA_MOVEMENT, rng)
                        for n, (k, row) in enumerate(items)]
            units.setdefault(f"camera_movement_{qt}", []).append((s, cm_unit))

    # visual correspondence (VC_C:11-12 seeds 1)
    np.random.seed(seed + 1)
    sampled = sampling.sample_dataframe(df, n_correspondence, 0, overlap_range[0], overlap_range[1], 1)
    units["visual_correspondence_coor_2_coor"] = []
    for s, items in sorted(rows_by_scene(sampled).items()):
        def vc_unit(scene, s=s, items=items):
            rng = random.Random(f"{seed}:vc:{s}")
            return heads.visual_correspondence_records(scene, [r for _, r in items], scenes[s].color_hw, 0,
                                                       T.VISUAL_CORRESPONDENCE, rng)
        units["visual_correspondence_coor_2_coor"].append((s, vc_unit))

    # depth estimation / comparison: every scene
    units["depth_estimation_coor"], units["depth_comparison_coor"] = [], []
    for s in range(len(scenes)):
        def de_unit(scene, s=s):
            return heads.depth_estimation_records(scene, scenes[s].scene_id, scenes[s].color_hw, depth_images_per_scene,
                                                  T.DEPTH_ESTIMATION, random.Random(f"{seed}:depth:{s}"))

        def dc_unit(scene, s=s):
            return heads.depth_comparison_records_gpu(scene, scenes[s].scene_id, scenes[s].color_hw, depth_images_per_scene,
                                                      T.DEPTH_COMPARISON, random.Random(f"{seed}:depthcmp:{s}"))
        units["depth_estimation_coor"].append((s, de_unit))
        units["depth_comparison_coor"].append((s, dc_unit))

    # object perception: visibility + coverage + records, per scene (COVIS / COV / OPE); one unit yields several files
    if object_perception:
        units["object_perception"] = []
        for s in range(len(scenes)):
            if not hasattr(scenes[s], "objects"):
                continue

            def op_unit(scene, s=s):
                idx, bbox, cat = scenes[s].objects()
                rng = random.Random(f"{seed}:op:{s}")
                cov, _ = scene.object_coverage(idx, bbox, rng=rng)
                by_name: Dict[str, List[dict]] = {}
                for dim in ("height", "length", "width"):
                    table = {scenes[s].scene_id: {o: res[dim] for o, res in cov.items()}}
                    size = {"height": lambda o: bbox[o][5], "length": lambda o: max(bbox[o][3], bbox[o][4]),
                            "width": lambda o: min(bbox[o][3], bbox[o][4])}[dim]
                    by_k = heads.object_perception_records(table, dim, lambda _s, o: size(o), lambda _s, o: cat[o],
                                                           scenes[s].color_hw, 6, T.OBJECT_PERCEPTION, rng)
                    for k, recs in by_k.items():
                        if recs:
                            by_name.setdefault(f"object_perception_{dim}_k{k}", []).extend(recs)
                return by_name
            units["object_perception"].append((s, op_unit))

    # object movement on TAPVid-style track blocks (OM_C): blocks sharded like scenes; a block is its own "scene"
    track_owner = {}
    if tracks:
        from scipy.cluster.hierarchy import fcluster, linkage
        from scipy.spatial.distance import squareform
        bins = shard.lpt_assign([float(t.tracks_XYZ.shape[0]) * t.tracks_XYZ.shape[1] ** 2 for t in tracks], world)
        track_owner = {k: r for r, b in enumerate(bins) for k in b}
        for qt in T.OBJECT_MOVEMENT_TYPES:
            units[f"object_movement_{qt}"] = []
            for k in range(len(tracks)):
                def om_unit(_scene, k=k, qt=qt):
                    import torch
                    from . import engine as E                      # the proxy while a tape is being recorded / replayed
                    tr = tracks[k]
                    rng = random.Random(f"{seed}:om:{qt}:{k}")
                    xyz = np.ascontiguousarray(tr.tracks_XYZ, dtype=np.float64)
                    dev_tracks = torch.from_numpy(xyz).to(device)
                    loss = E.track_rigidity_loss(dev_tracks).cpu().numpy()                              # K7
                    labels = fcluster(linkage(squareform(loss, checks=False), method="average"), 0.1, criterion="distance")
                    groups = [g for g in (np.where(labels == i)[0].tolist() for i in range(1, max(labels) + 1)) if len(g) > 5]
                    c2w = torch.from_numpy(np.linalg.inv(tr.extrinsics_w2c).reshape(-1, 16)).to(device)
                    world_xyz = E.track_to_world(dev_tracks, c2w, tr.fx_fy_cx_cy, tr.image_hw, ("world",))["world"]   # K5a
                    pairs_k = heads.object_movement_mine_pairs(                                       # K5c
                        tr.visibility, groups, lambda p, f: E.track_pair_distances(world_xyz, p, f), 5, 3, True, 0.05, rng)
                    return heads.object_movement_records(tr.scene_id, xyz, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                                         pairs_k, qt, T.OBJECT_MOVEMENT, rng, device)   # K5a + K5b
                units[f"object_movement_{qt}"].append((k, om_unit))

    def owner_of(name, key):
        return track_owner[key] if name.startswith("object_movement_") else scene_owner[key]

    def replay_scene(name, key, player):
        if name.startswith("object_movement_"):
            return None
        sc = scenes[key]
        return tape.ReplayScene(SceneOnDevice, sc.K, sc.A, sc.valid_image_ids, sc.color_hw, sc.points.shape[0], device,
                                count=player.next())

    def merge(outputs, name, produced):
        if isinstance(produced, dict):
            for n, recs in produced.items():
                outputs.setdefault(n, []).extend(recs)
        else:
            outputs.setdefault(name, []).extend(produced)

    # ---- run: directly with one process; record -> collate the numeric tapes -> replay on rank 0 with several ----------
    outputs: Dict[str, List[dict]] = {}
    for name in sorted(units):
        if name != "object_perception":
            outputs.setdefault(name, [])
        if ctx is None and not selftest_tape:
            for key, fn in units[name]:
                merge(outputs, name, fn(resident.get(key)))
            continue
        local = []
        own = {}                                               # records of the units this rank ran itself
        for key, fn in units[name]:
            if owner_of(name, key) != rank:
                continue
            rec = tape.Recorder(engine)
            with tape.engine_as(rec):
                if not name.startswith("object_movement_"):
                    rec.note(resident[key]._visibility()["count"])
                direct = fn(resident.get(key))
            rows = rec.rows()
            if rank != 0:                                      # rank 0 keeps its own records: nothing of its units crosses the fabric
                local += tape.frame(key, rows)
            own[key] = direct
            if selftest_tape:                                 # self-check: the replay must reproduce the records
                player = tape.Player(engine, rows, device)
                with tape.engine_as(player):
                    again = fn(replay_scene(name, key, player))
                assert again == direct, f"tape replay of {name} / unit {key} differs from the direct run"
        if ctx is None:
            for key, fn in units[name]:                        # selftest without a communicator: every unit is this rank's
                merge(outputs, name, own[key])
            continue
        payload = torch.from_numpy(np.concatenate(local, 0) if local else np.zeros((0, tape.WIDTH))).to(ctx.collective_device)
        table = shard.collate_records(payload, ctx, dst=0)                    # the exchange: float64 rows, to rank 0 only
        if rank != 0:
            continue
        tapes = tape.unframe(table.cpu().numpy())
        t_replay = time.perf_counter()
        for key, fn in units[name]:                            # unit order, whoever ran it
            if key in own:                                     # rank 0's own units: the records it already has
                merge(outputs, name, own[key])
                continue
            player = tape.Player(engine, tapes[key], device)
            with tape.engine_as(player):
                merge(outputs, name, fn(replay_scene(name, key, player)))
        LAST_TIMINGS["rank0_replay_s"] = LAST_TIMINGS.get("rank0_replay_s", 0.0) + (time.perf_counter() - t_replay)

    # ---- rank 0: canonical order, seeded shuffle, JSONL ------------------------------------------------------
    counts: Dict[str, int] = {}
    os.makedirs(out_dir, exist_ok=True)
    if rank == 0:
        for name in sorted(outputs):
            allrecs = list(outputs[name])
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
        print({"out": args.out, "records": counts, "timings": dict(LAST_TIMINGS)})
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
