"""Mirror of the reference's utils/scannet_utils/make_visibility_info.py (per-scene visibility index)."""
from __future__ import annotations

import json
import os
import pickle

import pandas as pd

DEBUG = False


def visibility_dict_to_frame(scene_visibility_dict) -> pd.DataFrame:
    """{scene: {"image_to_points": ..., "point_to_images": ...}} -> the (key, values) table every reader of
    the index expects: keys "scene:image_to_points:img" / "scene:point_to_images:idx", values JSON strings
    (the form ``convert_pkl_to_parquet`` writes upstream, :38-73; see SURVEY.md 2.1 on the key separator)."""
    data = []
    for scene_id, info in scene_visibility_dict.items():
        for image_id, points in info["image_to_points"].items():
            data.append((f"{scene_id}:image_to_points:{image_id}", json.dumps(points)))
        for point_idx, images in info["point_to_images"].items():
            data.append((f"{scene_id}:point_to_images:{point_idx}", json.dumps(images)))
    return pd.DataFrame(data, columns=["key", "values"])


def convert_pkl_to_parquet(pkl_file, parquet_file=None):
    """``x.pkl`` -> ``x.parquet`` next to it (reference signature), or to ``parquet_file`` when given."""
    if parquet_file is None:
        parquet_file = pkl_file.replace(".pkl", ".parquet")
    with open(pkl_file, "rb") as f:
        scene_visibility_dict = pickle.load(f)
    df = visibility_dict_to_frame(scene_visibility_dict)
    df.to_parquet(parquet_file, index=False)
    print(f"Converted {pkl_file} to {parquet_file}. The file has {len(df)} items in total.")


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


def process_scene_columns(scene_id, scene_infos, warning_file):
    """The same index as two CSR tables compacted on the device (mspa.visindex.VisibilityCSR): what ``run_split`` streams to
    parquet.  Empty images are logged exactly as ``process_scene`` logs them."""
    print(f"[process_scene] Start: {scene_id}")
    csr = scene_infos.scene_on_device(scene_id).visibility_csr()
    for image_id in csr.empty_images():
        with open(warning_file, "a") as f:
            f.write(f"[Warning] {scene_id}: {image_id} has no in-bound points.\n")
    print(f"[process_scene] Done: {scene_id}")
    return csr


def _visibility_csr(scene):
    """K1 launched NOW on the caller's stream, nothing waited for; the returned callable compacts the bit matrix into the two
    CSR tables (K9) and brings them to the host -- on the calling thread's own stream (``sweep.side_stream``), behind an event
    recorded after K1.  (May also return the finished ``VisibilityCSR``: what the GPU-less tests stand in.)"""
    import torch
    from mspa import sweep, visindex
    if scene.xyz is None:
        raise ValueError("scene uploaded without vertices")
    ids, n_points = list(scene.ids), int(scene.xyz.shape[0])
    if not ids or n_points == 0:                                       # MVI:103-123 with nothing to loop over
        return visindex.from_bits(None, ids, n_points)
    bits = scene._visibility()["bits"]
    launched = torch.cuda.Event()
    launched.record(torch.cuda.current_stream(bits.device))

    def finish(text=False, indices=True):
        with torch.cuda.device(bits.device):
            side = sweep.side_stream(bits.device)
            side.wait_event(launched)
            with torch.cuda.stream(side):
                return visindex.from_bits(bits, ids, n_points, text=text, indices=indices)     # returns with the tables on the host
    return finish


_CSR_FIELDS = ("i2p_offsets", "i2p_indices", "p2i_offsets", "p2i_indices")

def run_split(scene_info_path, output_file, warning_file, num_workers=8, keep=True, ctx=None, timings=None):
    """Visibility index of every scene of a split -> ``output_file`` (.parquet in the readers' format, or .pkl as the nested
    dict) (reference: :127-177, which maps scenes over ``Pool(num_workers)``, :151-156).

    ``num_workers`` = host threads that read and inflate the next scene's depth PNGs under the current scene's kernels; the
    scenes are sharded over the job's GPUs (one process per GPU: ``RANK`` / ``WORLD_SIZE`` from the environment, or ``ctx``).
    Every rank turns its scenes' bitsets into CSR tables on its GPU, formats the JSON text itself AND encodes each scene's row
    group as self-contained parquet bytes (81 MB of text per 320-frame scene: encoding + compression is the expensive part,
    54 ms per scene when rank 0 alone did it); the finished bytes of a window of scenes go to rank 0 (``shard.gather_bytes``),
    whose writer thread only splices them into the file in the split's order (mspa/parquet_splice.py) -- the file is byte for
    byte that of a one-process run.  ``keep=False`` drops each scene's index after it
    has been written (parquet output), for splits that do not fit in memory; the dict comes back on rank 0 only."""
    import numpy as np
    from mspa import parquet_splice, shard, sweep, visindex
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    scene_infos = SceneInfoHandler(scene_info_path)
    all_scene_ids = scene_infos.get_all_scene_ids()
    if ctx is None:
        ctx = shard.context_from_env()
    rank = ctx.rank if ctx is not None else 0
    out_dir = os.path.dirname(output_file)
    if out_dir and rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    if DEBUG and len(all_scene_ids) > 1:
        all_scene_ids = all_scene_ids[:1]
        print("[run_split] DEBUG mode. Only processing first scene.")
    print(f"[run_split] Found {len(all_scene_ids)} scenes in {scene_info_path}")
    as_pkl = output_file.endswith(".pkl")
    want_csr = as_pkl or keep                      # rank 0 rebuilds the nested dict from the CSR tables
    timings = timings if timings is not None else sweep.Timings()
    costs = scene_infos.scene_costs(all_scene_ids, ctx.world if ctx is not None else 1)
    device = ctx.device if ctx is not None else "cuda"
    scene_visibility_dict, state = {}, {"writer": None, "n": 0}

    def work_items(indices):
        return scene_infos.prefetched_scenes([all_scene_ids[i] for i in indices], max(1, int(num_workers)), device, timings)

    def produce(index, scene):
        """K1 is launched here, on the sweep's thread and stream, and nothing is waited for: compacting the bit matrix into the
        two CSR tables (K9), bringing ~100 MB of indices to the host, formatting and compressing the scene's text all happen on
        an encoder thread with a stream of its own -- the sweep thread goes straight on to the next scene."""
        scene_id = all_scene_ids[index]
        print(f"[process_scene] Start: {scene_id}")
        later = _visibility_csr(scene)

        def finish_scene():
            # the lists' JSON text is written on the device (K10) when a parquet file is what is asked for; the index arrays
            # themselves come to the host only if somebody wants them (the .pkl output, keep=True)
            csr = later(text=not as_pkl, indices=want_csr) if callable(later) else later
            lines = [f"[Warning] {scene_id}: {image_id} has no in-bound points.\n" for image_id in csr.empty_images()]
            blobs = ["".join(lines).encode()]
            if not as_pkl:
                # columns all the way: bitsets -> CSR on the device -> JSON text by libmspa's host formatters, straight into
                # arrow's buffers -> this scene's row group, encoded and compressed HERE; what leaves this rank is finished
                # parquet bytes.  No dictionary pages: every key and every JSON list is unique, one would be built, overflow and
                # be dropped.  63 ms of formatting + compression per 320-frame scene.
                blobs.append(parquet_splice.encode_row_group(csr.to_arrow(scene_id), use_dictionary=False))
            if want_csr:
                blobs += [np.ascontiguousarray(getattr(csr, f)) for f in _CSR_FIELDS]
            return blobs

        print(f"[process_scene] Done: {scene_id}")
        return None, finish_scene

    def consume(index, _rows, blobs):
        scene_id = all_scene_ids[index]
        if blobs[0].size:
            with open(warning_file, "a") as f:
                f.write(bytes(blobs[0]).decode())
        if want_csr:
            o1, i1, o2, i2 = blobs[-4:]
            ids = scene_infos.get_all_extrinsic_valid_image_ids(scene_id)
            csr = visindex.VisibilityCSR(list(ids), len(o2.view(np.int64)) - 1, o1.view(np.int64), i1.view(np.int32),
                                         o2.view(np.int64), i2.view(np.int32))
            scene_visibility_dict[scene_id] = csr.to_dict()
        if not as_pkl:
            with timings.span("write"):
                if state["writer"] is None:
                    state["writer"] = parquet_splice.SplicedParquetWriter(output_file)
                state["n"] += state["writer"].append(blobs[1])

    ok = False
    try:
        sweep.sharded_sweep(costs, ctx, work_items, produce, consume, timings=timings)
        ok = True
    finally:
        if state["writer"] is not None:
            state["writer"].__exit__(None if ok else RuntimeError, None, None)     # the footer only for a sweep that got through
    if as_pkl and rank == 0:
        with open(output_file, "wb") as f:
            pickle.dump(scene_visibility_dict, f)
        state["n"] = sum(len(v["image_to_points"]) + len(v["point_to_images"]) for v in scene_visibility_dict.values())
    if ctx is not None:
        ctx.barrier()
    if rank == 0:
        print(f"[run_split] Done. Wrote {state['n']} entries to {output_file}")
    return scene_visibility_dict if (keep or as_pkl) else {}


def main():
    """Same paths as upstream's main (:178-214): val, then train."""
    root = "data/scannet/scannet_instance_data"
    suffix = "_debug" if DEBUG else ""
    print("[main] DEBUG =", DEBUG)
    for split in ("val", "train"):
        out = os.path.join(root, f"{split}_visibility_info_D5{suffix}.parquet")
        print(f"[main] Generating {split} visibility -> {out}")
        run_split(os.path.join(root, f"scenes_{split}_info_i_D5.pkl"), out,
                  os.path.join(root, f"make_visibility_{split}_warning{suffix}.txt"), num_workers=25, keep=False)


if __name__ == "__main__":
    main()
