"""ScanNet ``.sens`` streams -> posed RGB-D frames, without the detour over PNG / JPEG / text files.

Reference: spatial_engine/utils/scannet_utils/extract_posed_images.py (SENS).  Upstream unpacks every frame of a
scene into ``posed_images/<scene>/{idx}.jpg|.png|.txt`` (SENS:59-157), a second script keeps every 5th of
those and parses the pose text back into the scene-info pickle (update_info_file_with_images.py:30-68, UPD),
and the task scripts re-read the PNGs frame by frame.  Here one pass over the file yields the depth frames as
one uint16 block ready for ``engine.depth_to_device`` and the scene-info entries the handler expects --
including the ``%f`` text round trip the poses and intrinsics go through upstream (six decimals; that rounding
is part of the reference's numbers, SENS:138-142 + UPD:32-35,50-53).

File layout (version 4, little endian; SENS:68-104, 31-48):
    u32 version | u64 strlen | sensor name | 4 x float32[16] (intrinsic_color, extrinsic_color, intrinsic_depth,
    extrinsic_depth) | i32 colour compression | i32 depth compression | u32 colour W, H | u32 depth W, H |
    f32 depth_shift | u64 n_frames | n_frames x { float32[16] camera_to_world | u64 t_colour | u64 t_depth |
    u64 colour bytes | u64 depth bytes | colour payload | depth payload }
"""
from __future__ import annotations

import dataclasses
import io
import os
import struct
import zlib
from typing import Dict, List, Optional, Sequence

import numpy as np

VERSION = 4
COLOR_COMPRESSION = {-1: "unknown", 0: "raw", 1: "png", 2: "jpeg"}
DEPTH_COMPRESSION = {-1: "unknown", 0: "raw_ushort", 1: "zlib_ushort", 2: "occi_ushort"}
_FRAME_HEAD = struct.Struct("<16f4Q")


@dataclasses.dataclass
class SensScene:
    sensor_name: bytes
    intrinsic_color: np.ndarray        # [4,4] float32
    extrinsic_color: np.ndarray
    intrinsic_depth: np.ndarray
    extrinsic_depth: np.ndarray
    color_compression: str
    depth_compression: str
    color_hw: tuple
    depth_hw: tuple
    depth_shift: float
    n_frames_total: int
    frame_index: List[int]             # indices into the stream of the frames kept (0, skip, 2*skip, ...)
    camera_to_world: np.ndarray        # [F,4,4] float32
    timestamps: np.ndarray             # [F,2] uint64 (colour, depth)
    depth: np.ndarray                  # [F,DH,DW] uint16
    color_jpeg: Optional[List[bytes]]  # undecoded payloads (None if not requested)
    export_position: Optional[List[int]] = None   # position of each kept frame among the frames upstream exports (names)
    depth_device: object = None        # [F,DH,DW] int16 device tensor (the uint16 values) when the frames were inflated on the GPU

    @staticmethod
    def index_to_str(index: int) -> str:
        return str(index).zfill(5)     # SENS:133-135


def read_sens(path: str, frame_skip: int = 1, want_color: bool = False, keep_every: int = 1, n_threads: int = 0,
              native: Optional[bool] = None, want_depth: bool = True, depth_to_device=None) -> SensScene:
    """Parse a .sens file keeping every ``frame_skip``-th frame (SENS:106-116) -- the frames upstream exports -- and, of
    those, only every ``keep_every``-th (UPD:20-68 keeps every 5th exported frame; the others need not be inflated at
    all).  One pass over the memory-mapped file collects the frame headers; the kept depth payloads are then inflated
    straight from the mapping into one [F, DH, DW] uint16 array by the library's copy threads
    (``mspa_inflate_blocks_host``; ``native=False`` or a missing library falls back to ``zlib`` frame by frame -- same
    bytes, this is file parsing, not the compute path).  ``want_depth=False`` reads headers and poses only (what the
    scene-info update needs): no payload is touched and ``depth`` comes back with zero frames.

    ``depth_to_device`` (a torch device): the kept frames' zlib payloads are copied, still compressed, into one pinned buffer,
    cross PCIe once and are inflated ON THE DEVICE, one wave per frame (``mspa_inflate_blocks_device``: the per-frame
    ``zlib.decompress`` of extract_posed_images.py:49-57) -- ``depth_device`` holds the [F, DH, DW] frames, ``depth`` stays empty.
    A frame the device declines (damaged stream, wrong size, checksum) is inflated by zlib on the host and uploaded on its own;
    if zlib rejects it too the error is zlib's."""
    import mmap
    with open(path, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        if size < 4:
            raise ValueError(f"{path}: truncated header")
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    try:
        view = memoryview(mm)
        pos = 0

        def take(n):
            nonlocal pos
            if pos + n > size:
                raise ValueError(f"{path}: truncated header")
            out = view[pos:pos + n]
            pos += n
            return out

        (version,) = struct.unpack("<I", take(4))
        if version != VERSION:
            raise AssertionError(f"unsupported .sens version {version}")          # SENS:71
        (strlen,) = struct.unpack("<Q", take(8))
        name = bytes(take(strlen))
        mats = [np.frombuffer(take(64), dtype="<f4").reshape(4, 4).copy() for _ in range(4)]
        c_comp, d_comp = struct.unpack("<ii", take(8))
        cw, ch, dw, dh = struct.unpack("<IIII", take(16))
        (shift,) = struct.unpack("<f", take(4))
        (n_frames,) = struct.unpack("<Q", take(8))
        color_compression, depth_compression = COLOR_COMPRESSION[c_comp], DEPTH_COMPRESSION[d_comp]
        if depth_compression not in ("zlib_ushort", "raw_ushort"):
            raise AssertionError(f"depth compression {depth_compression} not supported")   # SENS:51
        keep, positions, heads, d_off, d_len, c_off, c_len = [], [], [], [], [], [], []
        for i in range(n_frames):
            if pos + _FRAME_HEAD.size > size:
                raise ValueError(f"{path}: truncated at frame {i}")
            vals = _FRAME_HEAD.unpack_from(view, pos)
            pos += _FRAME_HEAD.size
            c_bytes, d_bytes = vals[18], vals[19]
            if pos + c_bytes + d_bytes > size:
                raise ValueError(f"{path}: truncated at frame {i}")
            if i % frame_skip == 0 and (i // frame_skip) % keep_every == 0:
                keep.append(i)
                positions.append(i // frame_skip)
                heads.append(vals)
                c_off.append(pos)
                c_len.append(c_bytes)
                d_off.append(pos + c_bytes)
                d_len.append(d_bytes)
            pos += c_bytes + d_bytes
        F = len(keep)
        poses = np.empty((F, 4, 4), dtype=np.float32)
        stamps = np.empty((F, 2), dtype=np.uint64)
        for k, vals in enumerate(heads):
            poses[k] = np.asarray(vals[:16], dtype=np.float32).reshape(4, 4)
            stamps[k] = (vals[16], vals[17])
        on_device = depth_to_device is not None and want_depth and depth_compression == "zlib_ushort" and F > 0
        depth = np.empty((F if (want_depth and not on_device) else 0, dh, dw), dtype=np.uint16)
        depth_device = None
        frame_bytes = dh * dw * 2
        jpeg: Optional[List[bytes]] = [bytes(view[o:o + n]) for o, n in zip(c_off, c_len)] if want_color else None
        if not want_depth:
            pass
        elif on_device:
            import torch
            from . import engine, ingest
            offs = np.zeros(F, dtype=np.int64)
            total = 0
            for k in range(F):
                offs[k] = total
                total += (d_len[k] + 15) // 16 * 16 + 16
            stage = ingest.PINNED_POOL.take(total)
            host = stage.numpy()
            whole = np.frombuffer(view, dtype=np.uint8)
            for k in range(F):
                host[offs[k]:offs[k] + d_len[k]] = whole[d_off[k]:d_off[k] + d_len[k]]
            del whole
            dev = torch.device(depth_to_device)
            src = stage[:total].to(dev, non_blocking=True)
            raw, status = engine.inflate_blocks_device(src, torch.from_numpy(offs).to(dev), torch.tensor(d_len, dtype=torch.int64, device=dev),
                                                       frame_bytes)
            frames = raw[:, :frame_bytes].contiguous().view(torch.int16).view(F, dh, dw) if raw.shape[1] != frame_bytes else \
                raw.view(torch.int16).view(F, dh, dw)
            bad = np.nonzero(status.cpu().numpy())[0]            # (synchronises: the staging buffer can go back afterwards)
            ingest.PINNED_POOL.give(stage)
            for k in bad:
                got = zlib.decompress(view[d_off[k]:d_off[k] + d_len[k]])
                if len(got) != frame_bytes:
                    raise ValueError(f"{path}: frame {keep[k]} inflates to {len(got)} bytes, expected {frame_bytes}")
                frames[int(k)] = torch.from_numpy(np.frombuffer(got, dtype="<u2").reshape(dh, dw).view(np.int16).copy()).to(dev)
            depth_device = frames
        elif depth_compression == "raw_ushort":
            for k in range(F):
                if d_len[k] != frame_bytes:
                    raise ValueError(f"{path}: frame {keep[k]} holds {d_len[k]} depth bytes, expected {frame_bytes}")
                depth[k] = np.frombuffer(view[d_off[k]:d_off[k] + frame_bytes], dtype="<u2").reshape(dh, dw)
        elif F:
            if native is None or native:
                try:
                    from . import _lib
                    lib = _lib.load()
                except Exception:
                    if native:
                        raise
                    lib = None
            else:
                lib = None
            if lib is not None:
                import ctypes
                whole = np.frombuffer(view, dtype=np.uint8)                 # zero-copy view of the mapping
                base = whole.ctypes.data
                ptrs = (ctypes.c_void_p * F)(*[base + o for o in d_off])
                lens = (ctypes.c_int64 * F)(*d_len)
                threads = n_threads if n_threads > 0 else min(64, os.cpu_count() or 1)
                _lib.check(lib.mspa_inflate_blocks_host(ptrs, lens, F, frame_bytes, depth.ctypes.data, threads))
                del whole
            else:
                for k in range(F):
                    raw = zlib.decompress(view[d_off[k]:d_off[k] + d_len[k]])
                    if len(raw) != frame_bytes:
                        raise ValueError(f"{path}: frame {keep[k]} inflates to {len(raw)} bytes, expected {frame_bytes}")
                    depth[k] = np.frombuffer(raw, dtype="<u2").reshape(dh, dw)
        view.release()
    finally:
        try:
            mm.close()
        except BufferError:                   # a view escaped (error path): the mapping goes with the garbage collector
            pass
    return SensScene(name, mats[0], mats[1], mats[2], mats[3], color_compression, depth_compression, (ch, cw), (dh, dw),
                     float(shift), int(n_frames), keep, poses, stamps, depth, jpeg, positions, depth_device)


def text_roundtrip(matrix: np.ndarray) -> np.ndarray:
    """What a float32 matrix becomes after ``np.savetxt(fmt="%f")`` (SENS:138-142) and ``float(token)`` (UPD:32-35):
    float64 values rounded to six decimals; +-inf survive as +-inf."""
    m = np.asarray(matrix)
    return np.array([[float("%f" % v) for v in row] for row in m], dtype=np.float64)


def matrix_text(matrix: np.ndarray) -> str:
    """The text upstream writes for a pose / intrinsic matrix (one ``%f`` row per line, SENS:138-142)."""
    buf = io.StringIO()
    for line in np.asarray(matrix):
        np.savetxt(buf, line[np.newaxis], fmt="%f")
    return buf.getvalue()


def _kept(sens: SensScene, image_frame_skip: int):
    """(row in the arrays, exported position) of the frames UPD:20-68 keeps: every ``image_frame_skip``-th exported frame."""
    pos = sens.export_position if sens.export_position is not None else list(range(len(sens.frame_index)))
    out = [(k, e) for k, e in enumerate(pos) if e % image_frame_skip == 0]
    if pos and len(out) != len(range(0, pos[-1] + 1, image_frame_skip)):
        raise ValueError("the stream was read with a keep_every that drops frames image_frame_skip needs")
    return out


def scene_info_entries(scene_id: str, sens: SensScene, image_frame_skip: int = 5) -> dict:
    """num_posed_images / images_info / intrinsic_matrix of one scene as UPD:20-68 builds them from the exported
    folder: frames are named by their position among the exported ones, every ``image_frame_skip``-th is kept."""
    images = {}
    for k, e in _kept(sens, image_frame_skip):
        image_id = SensScene.index_to_str(e)
        images[image_id] = {"image_path": f"posed_images/{scene_id}/{image_id}.jpg",
                            "depth_image_path": f"posed_images/{scene_id}/{image_id}.png",
                            "extrinsic_matrix": text_roundtrip(sens.camera_to_world[k])}
    return {"num_posed_images": len(images), "images_info": images,
            "intrinsic_matrix": text_roundtrip(sens.intrinsic_color)}


def depth_frames(sens: SensScene, image_frame_skip: int = 5) -> Dict[str, np.ndarray]:
    """{image_id: uint16 depth frame} for the frames ``scene_info_entries`` keeps (views, no copies)."""
    return {SensScene.index_to_str(e): sens.depth[k] for k, e in _kept(sens, image_frame_skip)}


def export_posed_images(sens: SensScene, output_path: str, with_depth_png: bool = True):
    """The folder extract_posed_images.process_scene leaves behind (SENS:161-178): intrinsic.txt, {idx}.txt,
    {idx}.png (16-bit) and -- when the colour payloads were read -- {idx}.jpg.  The JPEG payload is written as
    stored in the stream instead of being decoded and re-encoded."""
    os.makedirs(output_path, exist_ok=True)
    with open(os.path.join(output_path, "intrinsic.txt"), "w") as f:
        f.write(matrix_text(sens.intrinsic_color))
    pos = sens.export_position if sens.export_position is not None else list(range(len(sens.frame_index)))
    for k, e in enumerate(pos):
        stem = os.path.join(output_path, SensScene.index_to_str(e))
        with open(stem + ".txt", "w") as f:
            f.write(matrix_text(sens.camera_to_world[k]))
        if sens.color_jpeg is not None:
            with open(stem + ".jpg", "wb") as f:
                f.write(sens.color_jpeg[k])
        if with_depth_png:
            from PIL import Image
            Image.fromarray(sens.depth[k]).save(stem + ".png")


def write_sens(path: str, intrinsic_color: np.ndarray, camera_to_world: Sequence[np.ndarray], depth: Sequence[np.ndarray],
               color_hw=(968, 1296), color_payloads: Optional[Sequence[bytes]] = None, sensor_name: bytes = b"synthetic",
               depth_shift: float = 1000.0, depth_compression: int = 1):
    """Write a version-4 stream (for tests and synthetic scenes); the inverse of ``read_sens``."""
    depth = [np.ascontiguousarray(d, dtype="<u2") for d in depth]
    dh, dw = depth[0].shape
    eye = np.eye(4, dtype="<f4")
    with open(path, "wb") as f:
        f.write(struct.pack("<I", VERSION))
        f.write(struct.pack("<Q", len(sensor_name)))
        f.write(sensor_name)
        f.write(np.asarray(intrinsic_color, dtype="<f4").tobytes())
        f.write(eye.tobytes())
        f.write(np.asarray(intrinsic_color, dtype="<f4").tobytes())
        f.write(eye.tobytes())
        f.write(struct.pack("<ii", 2, depth_compression))
        f.write(struct.pack("<IIII", color_hw[1], color_hw[0], dw, dh))
        f.write(struct.pack("<f", depth_shift))
        f.write(struct.pack("<Q", len(depth)))
        for k, d in enumerate(depth):
            colour = color_payloads[k] if color_payloads is not None else b""
            payload = zlib.compress(d.tobytes()) if depth_compression == 1 else d.tobytes()
            f.write(np.asarray(camera_to_world[k], dtype="<f4").tobytes())
            f.write(struct.pack("<QQQQ", 1000 * k, 1000 * k + 1, len(colour), len(payload)))
            f.write(colour)
            f.write(payload)
