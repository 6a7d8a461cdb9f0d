"""Mirror of the reference's utils/scannet_utils/extract_posed_images.py on top of ``mspa.sens``: the same ``SensorData``
surface (load with ``frame_skip``, export of depth PNGs, colour JPEGs, pose and intrinsic text files) and the same
``posed_images/<scene>/`` layout.  Differences, both deliberate: skipped frames are seeked over instead of being read, and
the JPEG payloads are written out as stored instead of being decoded and re-encoded."""
from __future__ import annotations

import os
import time
from argparse import ArgumentParser

import numpy as np

from mspa import sens as _sens

COMPRESSION_TYPE_COLOR = dict(_sens.COLOR_COMPRESSION)
COMPRESSION_TYPE_DEPTH = dict(_sens.DEPTH_COMPRESSION)


class RGBDFrame:
    """One kept frame: pose, timestamps and the undecoded payload sizes (reference: :28-56)."""

    def __init__(self, scene: "_sens.SensScene", k: int):
        self.camera_to_world = scene.camera_to_world[k]
        self.timestamp_color, self.timestamp_depth = (int(v) for v in scene.timestamps[k])
        self.color_data = scene.color_jpeg[k] if scene.color_jpeg is not None else b""
        self._depth = scene.depth[k]

    def decompress_depth(self, compression_type):
        assert compression_type in ("zlib_ushort", "raw_ushort")
        return self._depth.tobytes()

    def decompress_color(self, compression_type):
        assert compression_type == "jpeg"
        import io
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(self.color_data)).convert("RGB"))


class SensorData:
    def __init__(self, filename, frame_skip, want_color=True):
        self.version = _sens.VERSION
        self.load(filename, frame_skip, want_color)

    def load(self, filename, frame_skip, want_color=True):
        s = self._scene = _sens.read_sens(filename, frame_skip, want_color=want_color)
        self.sensor_name = s.sensor_name
        self.intrinsic_color, self.extrinsic_color = s.intrinsic_color, s.extrinsic_color
        self.intrinsic_depth, self.extrinsic_depth = s.intrinsic_depth, s.extrinsic_depth
        self.color_compression_type, self.depth_compression_type = s.color_compression, s.depth_compression
        self.color_height, self.color_width = s.color_hw
        self.depth_height, self.depth_width = s.depth_hw
        self.depth_shift = s.depth_shift
        print(f"Number of total frames: {s.n_frames_total}")
        self.frames = [RGBDFrame(s, k) for k in range(len(s.frame_index))]
        print(f"Exported {len(self.frames)} frames. Frame skip is {frame_skip}.")

    index_to_str = staticmethod(_sens.SensScene.index_to_str)

    @staticmethod
    def save_mat_to_file(matrix, filename):
        with open(filename, "w") as f:
            f.write(_sens.matrix_text(matrix))

    def export_depth_images(self, output_path):
        from PIL import Image
        os.makedirs(output_path, exist_ok=True)
        for k in range(len(self.frames)):
            Image.fromarray(self._scene.depth[k]).save(os.path.join(output_path, self.index_to_str(k) + ".png"))

    def export_color_images(self, output_path):
        os.makedirs(output_path, exist_ok=True)
        for k, frame in enumerate(self.frames):
            with open(os.path.join(output_path, self.index_to_str(k) + ".jpg"), "wb") as f:
                f.write(frame.color_data)

    def export_poses(self, output_path):
        os.makedirs(output_path, exist_ok=True)
        for k, frame in enumerate(self.frames):
            self.save_mat_to_file(frame.camera_to_world, os.path.join(output_path, self.index_to_str(k) + ".txt"))

    def export_intrinsics(self, output_path):
        os.makedirs(output_path, exist_ok=True)
        self.save_mat_to_file(self.intrinsic_color, os.path.join(output_path, "intrinsic.txt"))


def process_scene(path, frame_skip, idx):
    """posed_images/<idx>/ from <path>/<idx>/<idx>.sens (reference: :161-178)."""
    print(f"Processing {idx}.")
    t1 = time.time()
    output_path = os.path.join("posed_images", idx)
    if os.path.exists(output_path):
        print(f"{output_path} already exists. Skip.")
        return
    data = SensorData(os.path.join(path, idx, f"{idx}.sens"), frame_skip)
    data.export_color_images(output_path)
    data.export_intrinsics(output_path)
    data.export_poses(output_path)
    data.export_depth_images(output_path)
    print(f"Finish processing {idx}. Using {time.time() - t1}s.")


def process_directory(path, frame_skip, nproc):
    print(f"processing {path}")
    scan_ids = sorted(os.listdir(path))
    if nproc and nproc > 1:
        from functools import partial
        from multiprocessing import Pool
        with Pool(nproc) as pool:
            pool.map(partial(process_scene, path, frame_skip), scan_ids)
    else:
        for idx in scan_ids:
            process_scene(path, frame_skip, idx)


if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--frame_skip", type=int, default=1, help="export every nth frame")
    parser.add_argument("--nproc", type=int, default=20)
    args = parser.parse_args()
    if os.path.exists("scans"):
        process_directory("scans", args.frame_skip, args.nproc)
    if os.path.exists("scans_test"):
        process_directory("scans_test", args.frame_skip, args.nproc)
