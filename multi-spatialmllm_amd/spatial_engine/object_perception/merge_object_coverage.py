"""Mirror of the reference's object_perception/merge_object_coverage.py: stitch the per-range coverage
pickles (``{split}_{start}_{end}/{split}_object_coverage_{dim}_{start}_{end}.pkl``) into one table per
dimension.  Host-only bookkeeping."""
from __future__ import annotations

import glob
import os
import pickle
import re

DIMENSIONS = ("height", "length", "width")


def merge_dimension(split, base_dir, dimension):
    merged = {}
    ranges = []
    for d in os.listdir(base_dir):
        m = re.match(fr"{split}_(\d+)_(\d+)", d)
        if os.path.isdir(os.path.join(base_dir, d)) and d.startswith(f"{split}_"):
            if m:
                ranges.append((int(m.group(1)), d))
            else:
                print(f"Directory name format does not match requirements, skipping: {d}")
    if not ranges:
        print(f"No subdirectories starting with {split}_ found in {base_dir}.")
        return merged
    for _, d in sorted(ranges, key=lambda r: r[0]):
        files = glob.glob(os.path.join(base_dir, d, f"{split}_object_coverage_{dimension}_*_*.pkl"))
        if not files:
            print(f"No {dimension} files found in subdirectory {d}, skipping.")
        for path in files:
            with open(path, "rb") as f:
                merged.update(pickle.load(f))
    return merged


def merge_split(split, base_dir, output_dir):
    out = {}
    for dim in DIMENSIONS:
        out[dim] = merge_dimension(split, base_dir, dim)
        with open(os.path.join(output_dir, f"merged_{split}_object_coverage_{dim}.pkl"), "wb") as f:
            pickle.dump(out[dim], f)
        print(f"After merging {split} {dim}, there are {len(out[dim])} scene_ids in total.")
    return out


def main():
    merge_split("train", "training_data/object_perception", "training_data/object_perception")
    merge_split("val", "evaluation_data/object_perception", "evaluation_data/object_perception")


if __name__ == "__main__":
    main()
