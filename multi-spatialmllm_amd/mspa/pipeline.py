"""End-to-end regeneration pipeline: posed RGB-D scenes -> pair table -> sampled QA records -> JSONL.

One process per GPU.  Scenes are assigned to ranks longest-processing-time-first (``shard.lpt_assign``);
every rank reads ONLY its own scenes (depth PNGs inflated by native loader threads, ``mspa.ingest`` / ``mspa.sweep``), keeps
them resident and runs their geometry on its GPU (K1 visibility, K2 overlap, K4 pose for the pair table; K6/K4/K5/K7/K8 for
the heads) with **no collective on the data path**.  Two exchange steps, both collations:
  1. the numeric pair tables (fixed-width rows: scene, image1, image2, overlap, distance, yaw, pitch)
     are all-gathered (``shard.collate_records``) so that every rank sees the same global table and the
     overlap-binned sampling (``sampling.sample_dataframe``, seeded) picks the same rows everywhere --
     exactly what a single process would pick;
  2. the FINISHED QA records: every rank builds the text of the units it owns (a unit = one scene or one track block of one
     head, its generator seeded per unit, so what it produces does not depend on who runs it), serialises them as JSON lines
     and ONE ``shard.gather_bytes`` (RCCL gather of uint8 tensors over xGMI) brings them to rank 0 -- BASELINE.json's "an RCCL
     all-gather over xGMI only to collate the final QA records".  Rank 0 never rebuilds a record: it splits the received
     bytes into lines, puts them into the canonical order (record id, then the line itself), applies the head's seeded
     shuffle and writes the JSONL.  Text building therefore scales with the number of ranks.
This is BASELINE.json configs[4] in miniature: camera movement, visual correspondence, depth estimation / comparison and
object perception from the posed RGB-D scenes, object movement from TAPVid-style track blocks (``tracks=``).  Scenes come in
as arrays (``mspa.synth``) or from disk in the reference's layout (``DiskScene``: scene-info pickle + posed_images +
scannet_instance_data).

    python -m mspa.pipeline --scenes 4 --frames 12 --out /tmp/mspa_out                         # synthetic, 1 GPU
    python -m mspa.pipeline --scene-info data/scannet/scannet_instance_data/scenes_train_info_i_D5.pkl --out ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mspa.pipeline ...
"""
from __future__ import annotations

import argparse
import json
import os
import random
import struct
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import heads, sampling, shard
from . import templates as T


class DiskScene:
    """A scene of a scene-info pickle, in the reference's on-disk layout, read on demand: what ``run`` needs of a scene it
    does not own is metadata only (ids, sizes); ``load()`` -- called by the owner rank alone -- decodes the depth frames."""

    def __init__(self, handler, scene_id: str, num_workers: int = 8):
        self.handler, self.scene_id, self.num_workers = handler, scene_id, num_workers
        self.valid_image_ids = handler.get_all_extrinsic_valid_image_ids(scene_id)
        self.color_hw = tuple(handler.get_image_shape(scene_id))
        path = os.path.join(handler.instance_data_root, scene_id, "aligned_points.npy")
        self.n_points = int(np.load(path, mmap_mode="r").shape[0])

    def load(self):
        return self.handler.host_scene(self.scene_id, self.num_workers)

    def has_objects(self) -> bool:
        return self.handler.get_num_objects(self.scene_id) > 0 and \
            os.path.exists(os.path.join(self.handler.instance_data_root, self.scene_id, "instance_mask.npy"))

    def objects(self):
        """({obj: vertex indices}, {obj: aligned bbox (cx, cy, cz, dx, dy, dz)}, {obj: category}) from the scene's instance
        mask and the info record, as the object_perception scripts read them (COVIS:86-101, OPE:60-75)."""
        h, sid = self.handler, self.scene_id
        mask = h.get_scene_instance_mask(sid)
        idx, bbox, cat = {}, {}, {}
        for o in range(h.get_num_objects(sid)):
            pts = np.where(mask == o + 1)[0]
            if len(pts) == 0:
                continue
            idx[o], bbox[o], cat[o] = pts, np.asarray(h.get_object_gt_bbox(sid, o), dtype=np.float64), h.get_object_raw_category(sid, o)
        return idx, bbox, cat


def _n_points(sc) -> int:
    return int(sc.n_points) if hasattr(sc, "n_points") else int(sc.points.shape[0])


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


LAST_TIMINGS: Dict[str, float] = {}      # of the most recent run() in this process: seconds per phase on this rank


def _pack_outputs(outputs: Dict[str, List[dict]]) -> bytes:
    """{file name: records} -> one byte string: per name the records as JSON lines (the text ``heads.write_jsonl`` writes)
    and their sort keys, str(record id), line by line."""
    parts = [struct.pack("<q", len(outputs))]
    for name in sorted(outputs):
        lines = "".join(json.dumps(r) + "\n" for r in outputs[name]).encode()
        keys = "".join(str(r["id"]).replace("\n", " ") + "\n" for r in outputs[name]).encode()
        nm = name.encode()
        parts += [struct.pack("<qqqq", len(nm), len(outputs[name]), len(lines), len(keys)), nm, lines, keys]
    return b"".join(parts)


def _unpack_outputs(buf, into: Dict[str, list]):
    """Append (key, line) pairs of one rank's packed outputs to ``into[name]``."""
    mv = memoryview(buf)
    if len(mv) == 0:
        return
    (n_names,) = struct.unpack_from("<q", mv, 0)
    pos = 8
    for _ in range(n_names):
        ln_name, n_rec, ln_lines, ln_keys = struct.unpack_from("<qqqq", mv, pos)
        pos += 32
        name = bytes(mv[pos:pos + ln_name]).decode()
        pos += ln_name
        lines = bytes(mv[pos:pos + ln_lines]).split(b"\n")[:-1] if ln_lines else []
        pos += ln_lines
        keys = bytes(mv[pos:pos + ln_keys]).decode().split("\n")[:-1] if ln_keys else []
        pos += ln_keys
        assert len(lines) == len(keys) == n_rec, f"{name}: framing of the record exchange is damaged"
        into.setdefault(name, []).extend(zip(keys, lines))


def run(scenes: Sequence, out_dir: str, ctx: Optional[shard.DistContext] = None, device="cuda", seed: int = 0,
        n_camera: int = 64, n_correspondence: int = 64, depth_images_per_scene: int = 4,
        overlap_range=(6, 35), question_types: Sequence[str] = ("total_distance", "displacement_vector"),
        object_perception: bool = True, tracks: Sequence = (), loader_threads: int = 2) -> Dict[str, int]:
    """Run the ScanNet-side heads over ``scenes`` -- in-memory scenes (K, A, E, depth, points, color_hw, scene_id,
    valid_image_ids; ``mspa.synth``) or ``DiskScene`` objects -- and the object-movement family over ``tracks``.
    Returns {jsonl name: record count} on rank 0 (empty dict elsewhere)."""
    import pandas as pd
    import torch
    from . import sweep
    from .scene import SceneOnDevice

    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    LAST_TIMINGS.clear()
    t_phase = time.perf_counter()
    costs = [shard.scene_cost(len(sc.valid_image_ids), _n_points(sc)) for sc in scenes]
    scene_bins = shard.lpt_assign(costs, world)
    scene_owner = {k: r for r, b in enumerate(scene_bins) for k in b}
    mine = scene_bins[rank]

    # ---- my scenes: read (loader threads decode the next scene while this one is uploaded), keep resident ----------------
    def load(k):
        return scenes[k].load() if hasattr(scenes[k], "load") else scenes[k]
    resident = {}
    for k, hs in zip(mine, sweep.SceneLoader(load, mine, lookahead=max(1, int(loader_threads)))):
        resident[k] = SceneOnDevice(hs.K, hs.A, hs.E, hs.depth, hs.color_hw, hs.points, device,
                                    depth_scale=getattr(hs, "depth_scale", 0.001))
    LAST_TIMINGS["load_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()

    # ---- geometry of my scenes; pair table collation --------------------------------------------
    local = [pair_table_rows(k, resident[k]) for k in mine]
    local = torch.cat(local, 0) if local else torch.zeros((0, 7), dtype=torch.float64, device=device)
    table = shard.collate_records(local, ctx) if ctx is not None else local
    table = table.cpu().numpy()
    order = np.lexsort((table[:, 2], table[:, 1], table[:, 0]))          # rank-independent row order
    table = table[order]
    # id columns by fancy indexing (no Python object per row: ScanNet's table has 10^8 of them)
    frame_ids = [list(scenes[k].valid_image_ids) for k in range(len(scenes))]
    first = np.concatenate([[0], np.cumsum([len(f) for f in frame_ids])]).astype(np.int64)
    flat_ids = np.array([i for f in frame_ids for i in f] or [""], dtype=object)
    scene_col = table[:, 0].astype(np.int64)
    df = pd.DataFrame({
        "scene_id": np.array([sc.scene_id for sc in scenes] or [""], dtype=object)[scene_col],
        "image_id1": flat_ids[first[scene_col] + table[:, 1].astype(np.int64)],
        "image_id2": flat_ids[first[scene_col] + table[:, 2].astype(np.int64)],
        "overlap": table[:, 3], "distance": table[:, 4], "yaw": table[:, 5], "pitch": table[:, 6],
        "_scene": scene_col,
    })

    # ---- the heads as units of work: (collation name, unit key, fn(scene) -> records or {name: records}) ----------------
    # A unit is one scene (or one track block) of one head; its generator is seeded per unit, so what it produces does not
    # depend on which rank runs it.
    from . import engine
    LAST_TIMINGS["pair_table_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()
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
            if not hasattr(scenes[s], "objects") or (isinstance(scenes[s], DiskScene) and not scenes[s].has_objects()):
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
                    from . import engine as E
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

    def merge(outputs, name, produced):
        if isinstance(produced, dict):
            for n, recs in produced.items():
                outputs.setdefault(n, []).extend(recs)
        else:
            outputs.setdefault(name, []).extend(produced)

    # ---- run the units this rank owns: numbers AND text, here ------------------------------------------------------
    outputs: Dict[str, List[dict]] = {}
    for name in sorted(units):
        if name != "object_perception":
            outputs.setdefault(name, [])
        for key, fn in units[name]:
            if owner_of(name, key) == rank:
                merge(outputs, name, fn(resident.get(key)))
    LAST_TIMINGS["heads_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()

    # ---- the one exchange of finished records: bytes to rank 0 -------------------------------------------------------
    packed = _pack_outputs(outputs)
    LAST_TIMINGS["records_bytes"] = float(len(packed))
    parts = shard.gather_bytes(packed, ctx, dst=0) if ctx is not None else [np.frombuffer(packed, dtype=np.uint8)]
    LAST_TIMINGS["exchange_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()

    # ---- rank 0: canonical order, seeded shuffle, JSONL ------------------------------------------------------
    counts: Dict[str, int] = {}
    os.makedirs(out_dir, exist_ok=True)
    if rank == 0:
        merged: Dict[str, list] = {}
        for p in parts:
            _unpack_outputs(p, merged)
        for name in sorted(merged):
            allrecs = merged[name]
            allrecs.sort()                                      # canonical order first: (str(id), line) -- sharding-independent
            random.Random(f"{seed}:{name}").shuffle(allrecs)
            with open(os.path.join(out_dir, f"{name}.jsonl"), "wb") as f:
                f.writelines(line + b"\n" for _, line in allrecs)
            counts[name] = len(allrecs)
        LAST_TIMINGS["write_s"] = time.perf_counter() - t_phase
    return counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene-info", default=None, help="scene-info pickle (info_handler.py:7-30): run on these on-disk scenes")
    ap.add_argument("--posed-images-root", default="data/scannet/posed_images")
    ap.add_argument("--instance-data-root", default="data/scannet/scannet_instance_data")
    ap.add_argument("--tapvid-root", default=None, help="directory of TAPVid-3D .npz sample files (object movement)")
    ap.add_argument("--num-workers", type=int, default=8, help="host threads decoding a scene's depth PNGs")
    ap.add_argument("--scenes", type=int, default=4, help="synthetic scenes when no --scene-info is given")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--tracks", type=int, default=2, help="synthetic TAPVid-style track blocks when no --tapvid-root is given")
    ap.add_argument("--out", default="mspa_out")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    from . import synth
    if not torch.cuda.is_available():
        raise SystemExit("mspa.pipeline needs a ROCm GPU (no CPU fallback)")
    ctx = shard.context_from_env()
    device = ctx.device if ctx is not None else torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    if args.scene_info:
        from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
        handler = SceneInfoHandler(args.scene_info, posed_images_root=args.posed_images_root,
                                   instance_data_root=args.instance_data_root)
        scenes = [DiskScene(handler, sid, args.num_workers) for sid in handler.get_all_scene_ids()]
    else:
        scenes = [synth.make_scene(7000 + k, n_points=args.points, n_frames=args.frames + 3 * (k % 3),
                                   color_hw=(480, 640), with_color=False) for k in range(args.scenes)]
    if args.tapvid_root:
        from spatial_engine.object_movement.single_object_movement_engine_coord import jpeg_size, load_tapvid_sample
        tracks = []
        for name in sorted(n for n in os.listdir(args.tapvid_root) if n.endswith(".npz")):
            gt = load_tapvid_sample(os.path.join(args.tapvid_root, name))
            n_frames = gt["tracks_XYZ"].shape[0]
            w2c = gt["extrinsics_w2c"] if "extrinsics_w2c" in gt else np.repeat(np.eye(4)[None], n_frames, 0)
            tracks.append(synth.SynthTracks(os.path.splitext(name)[0], np.asarray(gt["tracks_XYZ"], np.float64), gt["visibility"],
                                            w2c, gt["fx_fy_cx_cy"], jpeg_size(bytes(gt["images_jpeg_bytes"][0]))))
    else:
        tracks = [synth.make_tracks(300 + k, T=120, P=96, n_groups=4) for k in range(args.tracks)]
    counts = run(scenes, args.out, ctx, device, args.seed, tracks=tracks)
    if ctx is None or ctx.rank == 0:
        print({"out": args.out, "records": counts, "timings": dict(LAST_TIMINGS)})
    if ctx is not None:
        ctx.barrier()
        ctx.close()


if __name__ == "__main__":
    main()
