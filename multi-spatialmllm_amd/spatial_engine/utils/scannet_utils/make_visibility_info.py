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


def run_split(scene_info_path, output_file, warning_file, num_workers=8, keep=True):
    """Visibility index of every scene of a split -> ``output_file`` (.parquet in the readers' format, or
    .pkl as the nested dict).  ``num_workers`` is accepted and ignored (GPU loop); ``keep=False`` drops each scene's
    index after it has been written (parquet output), for splits that do not fit in memory."""
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    scene_infos = SceneInfoHandler(scene_info_path)
    all_scene_ids = scene_infos.get_all_scene_ids()
    out_dir = os.path.dirname(output_file)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    if DEBUG and len(all_scene_ids) > 1:
        all_scene_ids = all_scene_ids[:1]
        print("[run_split] DEBUG mode. Only processing first scene.")
    print(f"[run_split] Found {len(all_scene_ids)} scenes in {scene_info_path}")
    scene_visibility_dict = {}
    if output_file.endswith(".pkl"):
        for scene_id in all_scene_ids:
            _, scene_visibility_dict[scene_id] = process_scene(scene_id, scene_infos, warning_file)
        with open(output_file, "wb") as f:
            pickle.dump(scene_visibility_dict, f)
        n = sum(len(v["image_to_points"]) + len(v["point_to_images"]) for v in scene_visibility_dict.values())
    else:
        # parquet: one row group per scene, streamed -- the train split is 13 GB of JSON strings and need not sit in memory
        import pyarrow as pa
        import pyarrow.parquet as pq
        writer, n = None, 0
        try:
            for scene_id in all_scene_ids:
                # columns all the way: bitsets -> CSR on the device -> JSON text by arrow compute kernels; the nested dict is
                # only built when the caller wants it back
                csr = process_scene_columns(scene_id, scene_infos, warning_file)
                if keep:
                    scene_visibility_dict[scene_id] = csr.to_dict()
                table = csr.to_arrow(scene_id)
                if writer is None:
                    writer = pq.ParquetWriter(output_file, table.schema)
                writer.write_table(table)
                n += table.num_rows
        finally:
            if writer is not None:
                writer.close()
    print(f"[run_split] Done. Wrote {n} entries to {output_file}")
    return scene_visibility_dict


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
