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
from .hostinfo import quietly
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


class TrackFile:
    """A TAPVid-3D sample file (``<scene>.npz``; keys as OM_C:441-444) known by name and size only: priced for the longest-first
    assignment without being opened, read -- ``load()`` -- by the rank that owns it alone (a 1 956-file split opened completely,
    JPEG payloads included, by every rank before sharding is what this replaces)."""

    def __init__(self, path: str):
        self.path = path
        self.scene_id = os.path.splitext(os.path.basename(path))[0]
        self.cost = float(os.path.getsize(path))

    def load(self):
        from spatial_engine.object_movement.single_object_movement_engine_coord import jpeg_size
        from . import synth
        with np.load(self.path, allow_pickle=True) as gt:
            xyz = np.asarray(gt["tracks_XYZ"], np.float64)
            w2c = gt["extrinsics_w2c"] if "extrinsics_w2c" in gt.files else np.repeat(np.eye(4)[None], xyz.shape[0], 0)
            return synth.SynthTracks(self.scene_id, xyz, gt["visibility"], w2c, gt["fx_fy_cx_cy"],
                                     jpeg_size(bytes(gt["images_jpeg_bytes"][0])))      # only the first payload is touched


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


class RecordSpill:
    """Rank 0's collector of finished records: (sort key, JSON line) pairs per output file, held in memory up to
    ``limit_bytes`` per file and in sorted run files on disk beyond that.  ``finish(name, rng)`` yields the lines in the
    pipeline's final order -- canonical order (key, then the line: independent of who produced what and when), then the seeded
    shuffle -- as an external merge, so what rank 0 holds is bounded by the limit, not by the 27 M records of a full
    regeneration.  Both routes give the same bytes (tests/test_pipeline_spill.py)."""

    def __init__(self, directory: str, limit_bytes: int):
        self.dir, self.limit = directory, max(1, int(limit_bytes))
        self.mem: Dict[str, list] = {}
        self.mem_bytes: Dict[str, int] = {}
        self.runs: Dict[str, List[str]] = {}
        self.counts: Dict[str, int] = {}
        self.total_bytes = 0
        self.runs_written = 0

    def names(self):
        return sorted(set(self.mem) | set(self.runs) | set(self.counts))

    def touch(self, name: str):
        self.counts.setdefault(name, 0)

    def add_packed(self, buf):
        got: Dict[str, list] = {}
        _unpack_outputs(buf, got)
        for name, pairs in got.items():
            self.counts[name] = self.counts.get(name, 0) + len(pairs)
            self.mem.setdefault(name, []).extend(pairs)
            n = sum(len(k) + len(ln) + 2 for k, ln in pairs)
            self.mem_bytes[name] = self.mem_bytes.get(name, 0) + n
            self.total_bytes += n
            if self.mem_bytes[name] > self.limit:
                self._spill(name)

    @staticmethod
    def _write_run(path, pairs):
        with open(path, "wb") as f:
            for k, ln in pairs:
                kb = k.encode()
                f.write(struct.pack("<ii", len(kb), len(ln)))
                f.write(kb)
                f.write(ln)

    @staticmethod
    def _read_run(path):
        with open(path, "rb") as f:
            while True:
                head = f.read(8)
                if len(head) < 8:
                    return
                nk, nl = struct.unpack("<ii", head)
                yield f.read(nk).decode(), f.read(nl)

    def _spill(self, name):
        pairs = self.mem.pop(name, [])
        self.mem_bytes[name] = 0
        if not pairs:
            return
        pairs.sort()
        os.makedirs(self.dir, exist_ok=True)
        path = os.path.join(self.dir, f"{name}.run{len(self.runs.get(name, []))}")
        self._write_run(path, pairs)
        self.runs_written += 1
        self.runs.setdefault(name, []).append(path)

    def finish(self, name: str, rng: random.Random):
        """The file's lines in final order; the run files of ``name`` are removed."""
        import heapq
        n = self.counts.get(name, 0)
        if not self.runs.get(name):                           # everything in memory: sort, shuffle
            pairs = self.mem.pop(name, [])
            pairs.sort()
            rng.shuffle(pairs)
            for _k, ln in pairs:
                yield ln
            return
        self._spill(name)                                     # the remainder becomes the last run
        runs = self.runs.pop(name)
        # the seeded shuffle of the canonical order, as a permutation: after `shuffle`, position j holds element perm[j]
        from array import array
        perm = array("q", range(n))
        rng.shuffle(perm)
        dest = array("q", bytes(8 * n))
        for j in range(n):
            dest[perm[j]] = j
        del perm
        # second external sort, by destination: chunks of the canonical stream, each sorted by where its lines go
        chunk, size, second = [], 0, []
        def flush_chunk():
            nonlocal chunk, size
            if chunk:
                chunk.sort()
                path = os.path.join(self.dir, f"{name}.dst{len(second)}")
                with open(path, "wb") as f:
                    for d, ln in chunk:
                        f.write(struct.pack("<qi", d, len(ln)))
                        f.write(ln)
                second.append(path)
                chunk, size = [], 0
        for i, (_k, ln) in enumerate(heapq.merge(*[self._read_run(p) for p in runs])):
            chunk.append((dest[i], ln))
            size += len(ln) + 16
            if size > self.limit:
                flush_chunk()
        flush_chunk()
        for pth in runs:
            os.remove(pth)

        def read_dst(path):
            with open(path, "rb") as f:
                while True:
                    head = f.read(12)
                    if len(head) < 12:
                        return
                    d, nl = struct.unpack("<qi", head)
                    yield d, f.read(nl)
        for _d, ln in heapq.merge(*[read_dst(pth) for pth in second]):
            yield ln
        for pth in second:
            os.remove(pth)


@quietly
def run(scenes: Sequence, out_dir: str, ctx: Optional[shard.DistContext] = None, device="cuda", seed: int = 0,
        n_camera: int = 64, n_correspondence: int = 64, depth_images_per_scene: int = 4,
        overlap_range=(6, 35), question_types: Sequence[str] = ("total_distance", "displacement_vector"),
        object_perception: bool = True, tracks: Sequence = (), loader_threads: int = 2,
        spill_bytes: Optional[int] = None) -> Dict[str, int]:
    """Run the ScanNet-side heads over ``scenes`` -- in-memory scenes (K, A, E, depth, points, color_hw, scene_id,
    valid_image_ids; ``mspa.synth``) or ``DiskScene`` objects -- and the object-movement family over ``tracks`` (arrays or
    ``TrackFile``s).  Returns {jsonl name: record count} on rank 0 (empty dict elsewhere).

    Streamed: no rank keeps more scenes resident than its upload slots, and rank 0 holds a window of records, not the job's.
      pass 1  every scene once, sharded in windows (``sweep.sharded_sweep``: longest-first, prefetched -- depth PNGs decoded
              on the device for on-disk scenes): its pair-table rows (K1 + K2 + K4) AND the heads that need nothing but the
              scene itself (depth estimation / comparison, object perception) while it is resident; rows and finished record
              bytes go to rank 0 per window (``collate_records(dst=0)`` + ``gather_bytes``: RCCL);
      sample  rank 0 draws the camera-movement and correspondence rows from the whole table (seeded, the reference's
              overlap-binned sampler) and broadcasts the few sampled rows;
      pass 2  the scenes that own sampled rows once more, sharded the same way: camera movement (K4) and correspondences
              (K2 + K6), text built by the owner;
      tracks  the TAPVid blocks, sharded by file size, read by their owners on loader threads;
      write   rank 0 puts every file into canonical order and applies its seeded shuffle -- in memory, or as an external merge
              over sorted run files once a file's records exceed ``spill_bytes`` (``RecordSpill``)."""
    import pandas as pd
    import resource
    import torch
    from . import sweep
    from .scene import SceneOnDevice

    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    LAST_TIMINGS.clear()
    if spill_bytes is None:
        spill_bytes = int(os.environ.get("MSPA_PIPELINE_SPILL_MB", "512")) << 20
    os.makedirs(out_dir, exist_ok=True)
    spill = RecordSpill(os.path.join(out_dir, ".spill"), spill_bytes) if rank == 0 else None
    timings = sweep.Timings()

    def scene_items(indices):
        """Resident scenes for ``indices``, in order: prefetched from disk (a DiskScene's handler streams them: native loader
        threads, pinned staging, on-device decode) or uploaded from the arrays of an in-memory scene."""
        indices = list(indices)
        if indices and all(isinstance(scenes[k], DiskScene) for k in indices) and len({id(scenes[k].handler) for k in indices}) == 1:
            h = scenes[indices[0]].handler
            it = h.prefetched_scenes([scenes[k].scene_id for k in indices], scenes[indices[0]].num_workers, device, timings)
        else:
            def gen():
                for k in indices:
                    hs = scenes[k].load() if hasattr(scenes[k], "load") else scenes[k]
                    yield SceneOnDevice(hs.K, hs.A, hs.E, hs.depth, hs.color_hw, hs.points, device,
                                        depth_scale=getattr(hs, "depth_scale", 0.001))
            it = gen()
        yield from it

    def merge(outputs, name, produced):
        if isinstance(produced, dict):
            for n, recs in produced.items():
                outputs.setdefault(n, []).extend(recs)
        else:
            outputs.setdefault(name, []).extend(produced)

    def op_records(scene, s):
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

    # ---- pass 1: pair-table rows + the scene-local heads -----------------------------------------------------------------
    t_phase = time.perf_counter()
    costs = [shard.scene_cost(len(sc.valid_image_ids), _n_points(sc)) for sc in scenes]
    table_parts: List[np.ndarray] = []
    records_bytes = [0]

    def produce1(s, scene):
        rows = pair_table_rows(s, scene)
        outputs: Dict[str, List[dict]] = {}
        # a unit is one scene of one head, its generator seeded per unit: what it produces does not depend on who runs it
        merge(outputs, "depth_estimation_coor",
              heads.depth_estimation_records(scene, scenes[s].scene_id, scenes[s].color_hw, depth_images_per_scene,
                                             T.DEPTH_ESTIMATION, random.Random(f"{seed}:depth:{s}")))
        merge(outputs, "depth_comparison_coor",
              heads.depth_comparison_records_gpu(scene, scenes[s].scene_id, scenes[s].color_hw, depth_images_per_scene,
                                                 T.DEPTH_COMPARISON, random.Random(f"{seed}:depthcmp:{s}")))
        if object_perception and hasattr(scenes[s], "objects") and \
                not (isinstance(scenes[s], DiskScene) and not scenes[s].has_objects()):
            merge(outputs, "object_perception", op_records(scene, s))
        packed = _pack_outputs(outputs)
        records_bytes[0] += len(packed)
        return rows, [packed]

    def consume_records(_index, rows, blobs):
        if rows is not None and len(rows):
            table_parts.append(np.array(rows, copy=True))
        spill.add_packed(blobs[0])

    sweep.sharded_sweep(costs, ctx, scene_items, produce1, consume_records, record_width=7, timings=timings)
    LAST_TIMINGS["load_s"] = timings.as_dict().get("decode", 0.0)
    LAST_TIMINGS["pair_table_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()

    # ---- sample on rank 0, broadcast the sampled rows --------------------------------------------------------------------
    plan = None
    if rank == 0:
        table = np.concatenate(table_parts, 0) if table_parts else np.zeros((0, 7))
        del table_parts
        order = np.lexsort((table[:, 2], table[:, 1], table[:, 0]))      # rank-independent row order
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

        def rows_by_scene(sampled):
            by: Dict[int, List] = {}
            for k in range(len(sampled)):
                row = sampled.iloc[k].to_dict()
                by.setdefault(int(row["_scene"]), []).append((k, {c: (v.item() if hasattr(v, "item") else v) for c, v in row.items()}))
            return by
        plan = {"cm": {}, "vc": {}}
        for qt in question_types:                                  # camera movement (seed as upstream: CME:17-18)
            np.random.seed(seed)
            random.seed(seed)
            plan["cm"][qt] = rows_by_scene(sampling.sample_dataframe(df, n_camera, 0, overlap_range[0], overlap_range[1], 1))
        np.random.seed(seed + 1)                                   # visual correspondence (VC_C:11-12 seeds 1)
        plan["vc"] = rows_by_scene(sampling.sample_dataframe(df, n_correspondence, 0, overlap_range[0], overlap_range[1], 1))
        LAST_TIMINGS["pair_table_rows"] = float(len(df))
        del df, table
    plan = shard.broadcast_object(plan, ctx, src=0)

    # ---- pass 2: the scenes that own sampled rows ------------------------------------------------------------------------
    wanted = sorted({s for qt in plan["cm"] for s in plan["cm"][qt]} | set(plan["vc"]))

    def produce2(pos, scene):
        s = wanted[pos]
        outputs: Dict[str, List[dict]] = {}
        for qt in question_types:
            items = plan["cm"][qt].get(s)
            if items:
                rng = random.Random(f"{seed}:{qt}:{s}")          # per-scene stream: result independent of the sharding
                t12, t21 = heads.camera_movement_numeric(scene, [r for _, r in items])
                merge(outputs, f"camera_movement_{qt}",
                      [heads.camera_movement_record(row, k, qt, t12[n], t21[n], scenes[s].color_hw, T.CAMERA_MOVEMENT, rng)
                       for n, (k, row) in enumerate(items)])
        items = plan["vc"].get(s)
        if items:
            merge(outputs, "visual_correspondence_coor_2_coor",
                  heads.visual_correspondence_records(scene, [r for _, r in items], scenes[s].color_hw, 0,
                                                      T.VISUAL_CORRESPONDENCE, random.Random(f"{seed}:vc:{s}")))
        packed = _pack_outputs(outputs)
        records_bytes[0] += len(packed)
        return None, [packed]

    n_rows = [float(sum(len(plan["cm"][qt].get(s, ())) for qt in plan["cm"]) + len(plan["vc"].get(s, ()))) for s in wanted]
    sweep.sharded_sweep(n_rows, ctx, lambda positions: scene_items([wanted[p] for p in positions]), produce2, consume_records,
                        timings=timings)
    LAST_TIMINGS["heads_s"] = time.perf_counter() - t_phase
    t_phase = time.perf_counter()

    # ---- object movement on TAPVid-style track blocks (OM_C): sharded by size, read by their owners ----------------------
    if tracks:
        from scipy.cluster.hierarchy import fcluster, linkage
        from scipy.spatial.distance import squareform
        from . import engine as E
        tcosts = [t.cost if hasattr(t, "cost") else float(t.tracks_XYZ.shape[0]) * t.tracks_XYZ.shape[1] ** 2 for t in tracks]

        def track_items(indices):
            return sweep.SceneLoader(lambda k: tracks[k].load() if hasattr(tracks[k], "load") else tracks[k], list(indices),
                                     lookahead=max(1, int(loader_threads)), timings=timings)

        def produce_tracks(k, tr):
            outputs: Dict[str, List[dict]] = {}
            xyz = np.ascontiguousarray(tr.tracks_XYZ, dtype=np.float64)
            dev_tracks = torch.from_numpy(xyz).to(device)
            loss = E.track_rigidity_loss(dev_tracks).cpu().numpy()                                  # K7
            labels = fcluster(linkage(squareform(loss, checks=False), method="average"), 0.1, criterion="distance")
            groups = [g for g in (np.where(labels == i)[0].tolist() for i in range(1, max(labels) + 1)) if len(g) > 5]
            c2w = torch.from_numpy(np.linalg.inv(tr.extrinsics_w2c).reshape(-1, 16)).to(device)
            world_xyz = E.track_to_world(dev_tracks, c2w, tr.fx_fy_cx_cy, tr.image_hw, ("world",))["world"]       # K5a
            for qt in T.OBJECT_MOVEMENT_TYPES:
                rng = random.Random(f"{seed}:om:{qt}:{k}")
                pairs_k = heads.object_movement_mine_pairs(                                           # K5c
                    tr.visibility, groups, lambda p, f: E.track_pair_distances(world_xyz, p, f), 5, 3, True, 0.05, rng)
                merge(outputs, f"object_movement_{qt}",
                      heads.object_movement_records(tr.scene_id, xyz, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                                    pairs_k, qt, T.OBJECT_MOVEMENT, rng, device))        # K5a + K5b
            packed = _pack_outputs(outputs)
            records_bytes[0] += len(packed)
            return None, [packed]
        sweep.sharded_sweep(tcosts, ctx, track_items, produce_tracks, consume_records, timings=timings)
    LAST_TIMINGS["tracks_s"] = time.perf_counter() - t_phase
    LAST_TIMINGS["exchange_s"] = timings.as_dict().get("exchange", 0.0)
    LAST_TIMINGS["records_bytes"] = float(records_bytes[0])
    t_phase = time.perf_counter()

    # ---- rank 0: canonical order, seeded shuffle, JSONL ------------------------------------------------------------------
    counts: Dict[str, int] = {}
    if rank == 0:
        for qt in question_types:
            spill.touch(f"camera_movement_{qt}")
        for name in ("visual_correspondence_coor_2_coor", "depth_estimation_coor", "depth_comparison_coor"):
            spill.touch(name)
        if tracks:
            for qt in T.OBJECT_MOVEMENT_TYPES:
                spill.touch(f"object_movement_{qt}")
        for name in spill.names():
            n = 0
            with open(os.path.join(out_dir, f"{name}.jsonl"), "wb") as f:
                for line in spill.finish(name, random.Random(f"{seed}:{name}")):
                    f.write(line + b"\n")
                    n += 1
            counts[name] = n
        if os.path.isdir(spill.dir) and not os.listdir(spill.dir):
            os.rmdir(spill.dir)
        LAST_TIMINGS["write_s"] = time.perf_counter() - t_phase
        LAST_TIMINGS["spill_runs"] = float(spill.runs_written)
    from . import upload
    # what a rank keeps resident at once: the prefetcher's upload slots for on-disk scenes (3, or DECODE_SLOTS when the depth
    # frames are decoded on the device), one scene for in-memory ones -- never the split
    LAST_TIMINGS["max_resident_scenes"] = float(max(upload.UPLOAD_SLOTS, upload.DECODE_SLOTS)
                                                if any(isinstance(sc, DiskScene) for sc in scenes) else 1)
    LAST_TIMINGS["peak_rss_mb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
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
        tracks = [TrackFile(os.path.join(args.tapvid_root, name)) for name in sorted(os.listdir(args.tapvid_root)) if name.endswith(".npz")]
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
