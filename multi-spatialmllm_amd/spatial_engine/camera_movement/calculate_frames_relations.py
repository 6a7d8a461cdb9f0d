"""Mirror of the reference's camera_movement/calculate_frames_relations.py (per-scene pair table)."""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch

from mspa import engine

DEBUG = False

_COLUMNS = ["scene_id", "image_id1", "image_id2", "overlap", "distance", "yaw", "pitch"]


def _rows(overlap_info, keep):
    return [{"scene_id": scene_id, "image_id1": a, "image_id2": b, "overlap": v["overlap"], "distance": v["distance"],
             "yaw": v["yaw"], "pitch": v["pitch"]}
            for scene_id, pairs in overlap_info.items() for (a, b), v in pairs.items() if keep(v)]


def save_overlap_info(overlap_info, parquet_file):
    """{scene: {(id1, id2): {overlap, distance, yaw, pitch}}} -> parquet with the pair-table columns
    scene_id, image_id1, image_id2, overlap, distance, yaw, pitch (reference: :28-57)."""
    rows = _rows(overlap_info, lambda v: True)
    if not rows:
        print(f"[save_overlap_info] Nothing to save to {parquet_file}.")
        return
    df = pd.DataFrame(rows, columns=_COLUMNS)
    df.to_parquet(parquet_file, index=False)
    print(f"[save_overlap_info] Saved {len(df)} records to {parquet_file}.")


def save_overlap_info_nonzero(overlap_info, parquet_file_nonzero):
    """Same table without the rows whose overlap == 0 (NaN overlaps stay: nan != 0.0, reference: :59-84)."""
    rows = _rows(overlap_info, lambda v: v["overlap"] != 0.0)
    if not rows:
        print("[save_overlap_info_nonzero] No nonzero-overlap pairs to save.")
        return
    df = pd.DataFrame(rows, columns=_COLUMNS)
    df.to_parquet(parquet_file_nonzero, index=False)
    print(f"[save_overlap_info_nonzero] Saved {len(df)} records to {parquet_file_nonzero}.")


def extract_yaw_pitch(R):
    """Yaw / pitch (degrees) of the camera's forward axis; 4x4 or 3x3 input (reference: :86-100)."""
    R = np.asarray(R)
    R3 = R[:3, :3] if R.shape == (4, 4) else R
    z = R3[:, 2]
    return np.degrees(np.arctan2(z[1], z[0])), np.degrees(np.arcsin(z[2] / np.linalg.norm(z)))


def calculate_camera_overlap(in_bounds_dict, image_id1, image_id2, use_cuda=False):
    """|a & b| / |a | b| * 100 of two boolean masks (reference: :102-137), on the GPU (K2).
    ``use_cuda`` is accepted for signature compatibility; the computation is always on the device."""
    a = np.ascontiguousarray(in_bounds_dict[image_id1], dtype=bool)
    b = np.ascontiguousarray(in_bounds_dict[image_id2], dtype=bool)
    n = a.shape[0]
    n_words = (n + 63) // 64
    packed = np.zeros((2, n_words * 8), dtype=np.uint8)
    packed[0, :(n + 7) // 8] = np.packbits(a, bitorder="little")
    packed[1, :(n + 7) // 8] = np.packbits(b, bitorder="little")
    bits = torch.from_numpy(packed.view(np.int64)).cuda()
    pairs = torch.tensor([[0, 1]], dtype=torch.int32, device="cuda")
    return np.float64(engine.pair_overlap(bits, pairs).cpu().numpy()[0])


def process_scene(scene_id, scene_infos, warning_file):
    """(scene_id, {(id1, id2): {overlap, distance, yaw, pitch}}) for all valid frame pairs i < j
    (reference: :139-197): K1 visibility bitsets, K2 overlaps, K4 pose differences."""
    print(f"Start processing {scene_id}.")
    scene = scene_infos.scene_on_device(scene_id)
    for image_id in scene.empty_frames():
        with open(warning_file, "a") as f:
            f.write(f"{scene_id}: {image_id} has no in bound points\n")
    table = scene.frames_relations()
    for key, vals in table.items():
        v = list(vals.values())
        if np.any(np.isnan(v)) or np.any(np.isinf(v)):
            with open(warning_file, "a") as f:
                f.write(f"{scene_id}: {key} has something wrong {v}. \n")
    print(f"Finished scene {scene_id}.")
    return scene_id, table


class PairTable:
    """One scene's pair table in columns (what ``run_split`` keeps instead of a dict with an entry per pair)."""

    def __init__(self, scene_id, ids, arrays):
        self.scene_id, self.ids, self.arrays = scene_id, ids, arrays

    def __len__(self):
        return len(self.arrays["overlap"])

    def to_arrow(self, nonzero_only=False):
        """The scene's rows as an arrow table.  The id columns are ONE arrow string array of the scene's frame ids gathered by
        the int32 pair indices, the scene column one repeated scalar: no Python object per row (51 040 rows per 320-frame
        scene: 12 ms as object arrays, under 1 ms this way; the parquet bytes do not depend on how the arrays were built)."""
        import pyarrow as pa
        a = self.arrays
        if nonzero_only:
            sel = np.flatnonzero(a["overlap"] != 0.0)                                          # NaN != 0.0 stays (reference: :67)
            pick = lambda x: np.take(x, sel)                                                   # noqa: E731
        else:
            sel = None
            pick = lambda x: x                                                                 # noqa: E731
        n = len(self) if sel is None else int(sel.size)
        ids = pa.array(list(self.ids), type=pa.string())
        return pa.table({
            "scene_id": pa.repeat(pa.scalar(self.scene_id, type=pa.string()), n),
            "image_id1": ids.take(pa.array(pick(a["i"]), type=pa.int32())),
            "image_id2": ids.take(pa.array(pick(a["j"]), type=pa.int32())),
            "overlap": pa.array(pick(a["overlap"]), type=pa.float64()),
            "distance": pa.array(pick(a["distance"]), type=pa.float64()),
            "yaw": pa.array(pick(a["yaw"]), type=pa.float64()),
            "pitch": pa.array(pick(a["pitch"]), type=pa.float64()),
        })


def _scene_table(scene_id, scene):
    """(PairTable, warning lines) of a resident scene: the numbers of ``process_scene`` as columns."""
    lines = [f"{scene_id}: {image_id} has no in bound points\n" for image_id in scene.empty_frames()]
    arrays = scene.frames_relations_arrays()
    lines += _bad_value_lines(scene_id, scene.ids, arrays)
    return PairTable(scene_id, list(scene.ids), arrays), lines


def _bad_value_lines(scene_id, ids, arrays):
    vals = np.stack([arrays[k] for k in ("overlap", "distance", "yaw", "pitch")], axis=1)
    return [f"{scene_id}: {(ids[arrays['i'][n]], ids[arrays['j'][n]])} has something wrong "
            f"{[np.float64(v) for v in vals[n]]}. \n" for n in np.where(~np.isfinite(vals).all(axis=1))[0]]


def process_scene_columns(scene_id, scene_infos, warning_file) -> PairTable:
    """``process_scene`` without a Python object per pair: the same numbers as columns, the same warning lines."""
    print(f"Start processing {scene_id}.")
    table, lines = _scene_table(scene_id, scene_infos.scene_on_device(scene_id))
    if lines:
        with open(warning_file, "a") as f:
            f.writelines(lines)
    print(f"Finished scene {scene_id}.")
    return table


def _empty_frames(scene):
    """Frames that see no vertex (CFR:159-161).  K1 is launched NOW; the returned callable reads its per-frame counts back -- later,
    on the calling thread's stream.  (May also return the list itself: what the GPU-less tests stand in.)"""
    cnt, ids = scene._visibility()["count"], list(scene.ids)
    return lambda: [k for k, c in zip(ids, cnt.cpu().numpy()) if c == 0]


def _device_rows(scene):
    """[n_pairs, 6] float64 rows (i, j, overlap, distance, yaw, pitch) of a resident scene, left on the device: what crosses
    the fabric (``shard.collate_records``) when the split is sharded over several GPUs."""
    F = len(scene.ids)
    if F < 2:
        return torch.zeros((0, 6), dtype=torch.float64, device=scene.device)
    pairs = engine.all_pairs(F, scene.device)
    out = torch.empty((pairs.shape[0], 6), dtype=torch.float64, device=scene.device)
    out[:, 0:2] = pairs.to(torch.float64)
    out[:, 2] = engine.scene_overlap(scene._visibility()["bits"])
    out[:, 3:6] = engine.pair_pose(*scene.pose_tables(), pairs)[:, 0:3]
    return out


def run_split(scene_info_path, output_parquet, warning_file, num_workers=15, save_interval=20, keep=True, ctx=None,
              timings=None):
    """Pair tables of every scene of a split -> ``output_parquet`` (+ ``*_nonzero.parquet``) (reference: :200-253).

    The reference maps scenes over ``Pool(num_workers)`` (:222-229).  Here ``num_workers`` is the number of host threads that
    read and inflate the NEXT scene's depth PNGs while the current scene's kernels run (mspa/sweep.py), and the scenes are
    sharded over the GPUs of the job -- one process per GPU, ``RANK`` / ``WORLD_SIZE`` from the environment
    (torch.distributed.run) or an explicit ``ctx`` (mspa.shard.DistContext): longest-first within windows of scenes.  The
    rank that OWNS a scene also encodes its two row groups (all pairs / nonzero overlap) as self-contained parquet bytes;
    after each window these travel to rank 0 (``shard.gather_bytes``: RCCL), whose writer thread only splices them into the
    two files in the split's scene order (mspa/parquet_splice.py) -- nothing is encoded, compressed or converted on rank 0,
    and the files are byte for byte those of a one-process run.

    The tables stay columnar from the kernels to the parquet row groups -- ScanNet's 106.8 M pairs as a dict with one entry
    per pair would not fit in memory -- and both files are streamed, one row group per scene.  Returns {scene_id: PairTable}
    on rank 0 ({} elsewhere; the numeric rows then also cross the fabric, ``shard.collate_records``); with ``keep=False`` each
    table is dropped once written and {} is returned."""
    from mspa import parquet_splice, shard, sweep
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    scene_infos = SceneInfoHandler(scene_info_path)
    all_scene_ids = scene_infos.get_all_scene_ids()
    print(f"[run_split] Found {len(all_scene_ids)} scenes in {scene_info_path}.")
    if DEBUG and len(all_scene_ids) > 1:
        all_scene_ids = all_scene_ids[:1]
        print("[run_split] DEBUG mode: processing only the first scene.")
    if ctx is None:
        ctx = shard.context_from_env()
    rank = ctx.rank if ctx is not None else 0
    nonzero_parquet = output_parquet.replace(".parquet", "_nonzero.parquet")
    out_dir = os.path.dirname(output_parquet)
    if out_dir and rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    timings = timings if timings is not None else sweep.Timings()
    costs = scene_infos.scene_costs(all_scene_ids, ctx.world if ctx is not None else 1)
    device = ctx.device if ctx is not None else "cuda"
    tables, writers, paths = {}, [None, None], (output_parquet, nonzero_parquet)

    def work_items(indices):
        return scene_infos.prefetched_scenes([all_scene_ids[i] for i in indices], max(1, int(num_workers)), device, timings)

    def produce(index, scene):
        """K1 / K2 / K4 are launched here, on the sweep's thread and stream, and nothing is waited for: the pair table's download,
        its warning lines and the two row groups' encoding happen on an encoder thread with a stream of its own, behind an event --
        the sweep thread goes straight on to the next scene (it used to stand 8 ms per scene in two `.cpu()` calls, behind the decode
        waves of the scenes in flight)."""
        scene_id = all_scene_ids[index]
        print(f"Start processing {scene_id}.")
        empty = _empty_frames(scene)
        rows_dev = _device_rows(scene)
        launched = None
        if rows_dev.is_cuda:
            launched = torch.cuda.Event()
            launched.record(torch.cuda.current_stream(rows_dev.device))

        def download():
            return (empty() if callable(empty) else empty), rows_dev.cpu().numpy()

        def finish_scene():
            if launched is None:
                empty_ids, rows = download()
            else:
                with torch.cuda.device(rows_dev.device):
                    side = sweep.side_stream(rows_dev.device)
                    side.wait_event(launched)
                    with torch.cuda.stream(side):
                        empty_ids, rows = download()
            lines = [f"{scene_id}: {image_id} has no in bound points\n" for image_id in empty_ids]
            ids = list(scene_infos.get_all_extrinsic_valid_image_ids(scene_id))
            arrays = _row_arrays(rows)
            lines += _bad_value_lines(scene_id, ids, arrays)
            t = PairTable(scene_id, ids, arrays)
            # Dictionary pages for the three id columns only: on the float64 columns the encoder hashes every value, overflows its
            # dictionary page and falls back to plain anyway -- 4 x the encoding time of a row group and a LARGER file (measured:
            # 22.6 -> 5.9 ms, 1.92 -> 1.56 MB per 51 040-row scene); readers see the same table
            return ["".join(lines).encode()] + [parquet_splice.encode_row_group(t.to_arrow(nz), use_dictionary=_COLUMNS[:3])
                                                for nz in (False, True)]

        print(f"Finished scene {scene_id}.")
        return (rows_dev if keep else None), finish_scene

    def consume(index, rows, blobs):
        scene_id = all_scene_ids[index]
        if blobs[0].size:
            with open(warning_file, "a") as f:
                f.write(bytes(blobs[0]).decode())
        if keep:
            ids = scene_infos.get_all_extrinsic_valid_image_ids(scene_id)
            tables[scene_id] = PairTable(scene_id, list(ids), _row_arrays(rows))
        with timings.span("write"):
            for w in range(2):
                if writers[w] is None:
                    writers[w] = parquet_splice.SplicedParquetWriter(paths[w])
                writers[w].append(blobs[1 + w])
        if (index + 1) % save_interval == 0:
            print(f"[run_split] {index + 1} scenes written to {output_parquet}")

    ok = False
    try:
        sweep.sharded_sweep(costs, ctx, work_items, produce, consume, record_width=6 if keep else None, timings=timings)
        ok = True
    finally:
        for w in writers:
            if w is not None:
                w.__exit__(None if ok else RuntimeError, None, None)      # the footer only for a sweep that got through
    if ctx is not None:
        ctx.barrier()
    if rank == 0:
        print(f"[run_split] Total number of records: {writers[0].num_rows if writers[0] else 0}")
        print(f"[run_split] Total number of nonzero records: {writers[1].num_rows if writers[1] else 0}")
    return tables


def _row_arrays(rows):
    """[n, 6] float64 rows (i, j, overlap, distance, yaw, pitch) -> PairTable's column dict."""
    return {"i": rows[:, 0].astype(np.int32), "j": rows[:, 1].astype(np.int32),
            "overlap": np.ascontiguousarray(rows[:, 2]), "distance": np.ascontiguousarray(rows[:, 3]),
            "yaw": np.ascontiguousarray(rows[:, 4]), "pitch": np.ascontiguousarray(rows[:, 5])}


def main():
    """Same paths as upstream's main (:255-289): train and val pair tables + their warning files."""
    train_dir, val_dir = "training_data/camera_movement", "evaluation_data/camera_movement"
    os.makedirs(train_dir, exist_ok=True)
    os.makedirs(val_dir, exist_ok=True)
    suffix = "_debug" if DEBUG else ""
    print(f"[main] DEBUG mode: {DEBUG}")
    for split, info, out_dir in (("train", "data/scannet/scannet_instance_data/scenes_train_info_i_D5.pkl", train_dir),
                                 ("val", "data/scannet/scannet_instance_data/scenes_val_info_i_D5.pkl", val_dir)):
        out = os.path.join(out_dir, f"{split}_camera_info_D5{suffix}.parquet")
        print(f"[main] Processing {split} split -> {out}")
        run_split(info, out, os.path.join(out_dir, f"{split}_warning_D5{suffix}.txt"), num_workers=25, save_interval=20,
                  keep=False)


if __name__ == "__main__":
    main()
