"""mspa.sens (.sens reader, scene-info entries, export) against what the reference's SensorData and
update_info_file_with_images.py made of the same synthetic stream (tests/golden/sens.npz, oracle/gen_golden.py)."""
import json
import os
import struct

import numpy as np
import pytest

from golden_util import GOLDEN_DIR
from mspa import sens as S


@pytest.fixture(scope="module")
def gold(tmp_path_factory):
    z = np.load(os.path.join(GOLDEN_DIR, "sens.npz"), allow_pickle=False)
    path = str(tmp_path_factory.mktemp("sens") / "scene5151_00.sens")
    with open(path, "wb") as f:
        f.write(z["sens_bytes"].tobytes())
    return z, path


@pytest.mark.parametrize("skip", [1, 2])
def test_reader_matches_reference(gold, skip, tmp_path):
    z, path = gold
    sc = S.read_sens(path, frame_skip=skip, want_color=True)
    cw, ch, dw, dh, n = (int(v) for v in z[f"skip{skip}_header"])
    assert sc.color_hw == (ch, cw) and sc.depth_hw == (dh, dw) and len(sc.frame_index) == n
    assert sc.frame_index == list(range(0, sc.n_frames_total, skip))
    assert sc.depth_compression == "zlib_ushort" and sc.color_compression == "jpeg" and sc.sensor_name == b"synthetic"
    assert sc.camera_to_world.dtype == np.float32
    assert np.array_equal(sc.camera_to_world, z[f"skip{skip}_c2w"], equal_nan=True)       # -inf poses included
    assert skip != 1 or np.isinf(sc.camera_to_world).any()
    assert sc.depth.dtype == np.uint16 and np.array_equal(sc.depth, z[f"skip{skip}_depth"])
    assert [bytes(c) for c in sc.color_jpeg] == [bytes(c) for c in z[f"skip{skip}_color"]]
    # the exported folder: same text, byte for byte; the JPEG payload passes through untouched
    out = str(tmp_path / "posed")
    try:
        import PIL  # noqa: F401
        S.export_posed_images(sc, out)
    except ImportError:
        S.export_posed_images(sc, out, with_depth_png=False)
    texts = json.loads(str(z[f"skip{skip}_text_json"]))
    for name, text in texts.items():
        assert open(os.path.join(out, name)).read() == text, name
    assert open(os.path.join(out, "00001.jpg"), "rb").read() == sc.color_jpeg[1]
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(os.path.join(out, "00002.png"))), sc.depth[2])
    except ImportError:
        pass
    # scene-info entries == the parse of that text (six-decimal float64), every 5th exported frame
    info = S.scene_info_entries("scene5151_00", sc, image_frame_skip=5)
    assert info["intrinsic_matrix"].dtype == np.float64
    assert np.array_equal(info["intrinsic_matrix"], z[f"skip{skip}_info_K"])
    E = np.stack([v["extrinsic_matrix"] for v in info["images_info"].values()])
    assert np.array_equal(E, z[f"skip{skip}_info_E"]) and info["num_posed_images"] == len(E)
    assert list(info["images_info"]) == [f"{k:05d}" for k in range(0, n, 5)]
    first = info["images_info"]["00000"]
    assert first["image_path"] == "posed_images/scene5151_00/00000.jpg"
    assert first["depth_image_path"] == "posed_images/scene5151_00/00000.png"
    frames = S.depth_frames(sc, 5)
    assert list(frames) == list(info["images_info"]) and np.array_equal(frames["00005"], sc.depth[5])


def test_text_roundtrip_values():
    m = np.array([[1.0000001, -0.1234565, np.inf, -np.inf], [0.0, -0.0, 1e-7, 123456.789]], dtype=np.float32)
    r = S.text_roundtrip(m)
    assert r.dtype == np.float64
    want = [[float("%f" % v) for v in row] for row in m]
    assert r.tolist() == want and r[0, 2] == np.inf and r[0, 3] == -np.inf
    assert S.matrix_text(np.eye(2, dtype=np.float32)) == "1.000000 0.000000\n0.000000 1.000000\n"


def test_write_read_roundtrip_and_errors(tmp_path):
    rng = np.random.default_rng(3)
    depth = [rng.integers(0, 65536, (6, 8), dtype=np.uint16) for _ in range(5)]
    poses = [rng.normal(size=(4, 4)).astype(np.float32) for _ in range(5)]
    K = np.eye(4, dtype=np.float32)
    for comp in (1, 0):                                      # zlib_ushort, raw_ushort
        path = str(tmp_path / f"s{comp}.sens")
        S.write_sens(path, K, poses, depth, color_hw=(12, 16), depth_compression=comp)
        sc = S.read_sens(path, frame_skip=2)
        assert sc.color_jpeg is None and sc.frame_index == [0, 2, 4]
        assert np.array_equal(sc.depth, np.stack(depth)[::2]) and np.array_equal(sc.camera_to_world, np.stack(poses)[::2])
        assert sc.timestamps[1].tolist() == [2000, 2001]
    raw = open(path, "rb").read()
    bad = str(tmp_path / "bad.sens")
    with open(bad, "wb") as f:
        f.write(struct.pack("<I", 3) + raw[4:])
    with pytest.raises(AssertionError):
        S.read_sens(bad)
    with open(bad, "wb") as f:
        f.write(raw[:len(raw) - 40])
    with pytest.raises((ValueError, Exception)):
        S.read_sens(bad)


@pytest.mark.gpu
def test_sens_to_resident_scene(gold):
    """.sens -> scene-info entries + depth block -> SceneOnDevice -> pair table, no image files in between."""
    from mspa.scene import SceneOnDevice
    z, path = gold
    sc = S.read_sens(path, frame_skip=1)
    info = S.scene_info_entries("scene5151_00", sc, image_frame_skip=2)
    frames = S.depth_frames(sc, 2)
    E = {k: v["extrinsic_matrix"] for k, v in info["images_info"].items()}
    pts = np.random.default_rng(0).uniform(0, 6, (500, 3))
    scene = SceneOnDevice(info["intrinsic_matrix"], np.eye(4), E, frames, sc.color_hw, pts, "cuda")
    assert scene.ids == [k for k, e in E.items() if np.isfinite(e).all()] and len(scene.ids) >= 3
    table = scene.frames_relations()
    assert len(table) == len(scene.ids) * (len(scene.ids) - 1) // 2


def test_extract_posed_images_mirror(gold, tmp_path, monkeypatch):
    """The mirrored scripts: SensorData surface, posed_images/<scene>/ layout == the reference's texts, and the scene-info
    update from the folder == from the .sens stream == the reference's parse."""
    import importlib
    import pickle
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-spatialmllm_amd")
    for name in [m for m in sys.modules if m == "spatial_engine" or m.startswith("spatial_engine.")]:
        if not (getattr(sys.modules[name], "__file__", None) or "").startswith(pkg):
            del sys.modules[name]
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    EPI = importlib.import_module("spatial_engine.utils.scannet_utils.extract_posed_images")
    UPD = importlib.import_module("spatial_engine.utils.scannet_utils.update_info_file_with_images")
    z, path = gold
    scans = tmp_path / "scans" / "scene5151_00"
    scans.mkdir(parents=True)
    (scans / "scene5151_00.sens").write_bytes(z["sens_bytes"].tobytes())
    data = EPI.SensorData(str(scans / "scene5151_00.sens"), 1)
    assert (data.color_width, data.color_height, data.depth_width, data.depth_height, len(data.frames)) == \
        tuple(int(v) for v in z["skip1_header"])
    assert data.depth_compression_type == "zlib_ushort" and np.array_equal(data.frames[3].camera_to_world, z["skip1_c2w"][3])
    assert np.frombuffer(data.frames[2].decompress_depth("zlib_ushort"), dtype=np.uint16).reshape(24, 32).tolist() == \
        z["skip1_depth"][2].tolist()
    monkeypatch.chdir(tmp_path)
    try:
        import PIL  # noqa: F401
    except ImportError:
        pytest.skip("Pillow is needed for the depth PNGs")
    EPI.process_directory("scans", 1, nproc=1)
    texts = json.loads(str(z["skip1_text_json"]))
    folder = tmp_path / "posed_images" / "scene5151_00"
    for name, text in texts.items():
        assert (folder / name).read_text() == text, name
    assert (folder / "00004.jpg").read_bytes() == bytes(z["skip1_color"][4])
    info_file = tmp_path / "infos.pkl"
    with open(info_file, "wb") as f:
        pickle.dump({"scene5151_00": {"num_objects": 0}}, f)
    out1 = UPD.update_info_file(str(info_file), str(tmp_path / "posed_images"), 5)
    with open(out1, "rb") as f:
        from_folder = pickle.load(f)["scene5151_00"]
    out2 = UPD.update_info_file(str(info_file), None, 5, sens_root=str(tmp_path / "scans"))
    with open(out2, "rb") as f:
        from_sens = pickle.load(f)["scene5151_00"]
    assert out1.endswith("infos_i_D5.pkl") and from_folder["num_posed_images"] == from_sens["num_posed_images"] == 3
    for src in (from_folder, from_sens):
        assert np.array_equal(src["intrinsic_matrix"], z["skip1_info_K"])
        assert np.array_equal(np.stack([v["extrinsic_matrix"] for v in src["images_info"].values()]), z["skip1_info_E"])
        assert list(src["images_info"]) == ["00000", "00005", "00010"] and src["num_objects"] == 0


def test_native_inflate_keep_every_and_corrupt_payload(tmp_path):
    """The two-pass reader: native multi-threaded inflate == zlib frame by frame; ``keep_every`` inflates only the frames
    UPD:20-68 keeps and names them by their exported position; a corrupt payload is refused with the frame named."""
    from mspa import _lib
    rng = np.random.default_rng(9)
    n = 23
    depth = [(rng.integers(0, 40, (12, 16)) + 100 * k).astype(np.uint16) for k in range(n)]
    poses = [rng.normal(size=(4, 4)).astype(np.float32) for _ in range(n)]
    path = str(tmp_path / "s.sens")
    S.write_sens(path, np.eye(4, dtype=np.float32), poses, depth, color_hw=(12, 16), color_payloads=[b"jpg%d" % k for k in range(n)])
    a = S.read_sens(path, frame_skip=2, native=True, n_threads=5, want_color=True)
    b = S.read_sens(path, frame_skip=2, native=False, want_color=True)
    assert np.array_equal(a.depth, b.depth) and np.array_equal(a.depth, np.stack(depth)[::2])
    assert a.color_jpeg == b.color_jpeg == [b"jpg%d" % k for k in range(0, n, 2)]
    assert a.export_position == list(range(12))
    full = S.scene_info_entries("scene0000_00", a, image_frame_skip=5)
    thin = S.read_sens(path, frame_skip=2, keep_every=5, native=True)
    assert thin.frame_index == [0, 10, 20] and thin.export_position == [0, 5, 10]
    got = S.scene_info_entries("scene0000_00", thin, image_frame_skip=5)
    assert list(got["images_info"]) == list(full["images_info"]) == ["00000", "00005", "00010"]
    for k in got["images_info"]:
        assert np.array_equal(got["images_info"][k]["extrinsic_matrix"], full["images_info"][k]["extrinsic_matrix"])
    fa, ft = S.depth_frames(a, 5), S.depth_frames(thin, 5)
    assert list(fa) == list(ft) and all(np.array_equal(fa[k], ft[k]) for k in fa)
    with pytest.raises(ValueError):
        S.scene_info_entries("s", S.read_sens(path, frame_skip=2, keep_every=2), image_frame_skip=5)
    # corrupt one payload: flip bytes in the middle of frame 4's zlib stream
    raw = bytearray(open(path, "rb").read())
    probe = S.read_sens(path)                                    # offsets via a second parse of the headers
    blob = zlib_payload_offset(raw, 4)
    raw[blob + 6:blob + 12] = b"\xff" * 6
    bad = str(tmp_path / "bad.sens")
    open(bad, "wb").write(raw)
    with pytest.raises(_lib.MspaError, match="block 4"):
        S.read_sens(bad, native=True)
    with pytest.raises(Exception):
        S.read_sens(bad, native=False)
    assert probe.n_frames_total == n


def zlib_payload_offset(raw, frame):
    """Byte offset of frame ``frame``'s depth payload in a stream written by write_sens."""
    pos = 4
    (strlen,) = struct.unpack_from("<Q", raw, pos)
    pos += 8 + strlen + 4 * 64 + 8 + 16 + 4
    (n,) = struct.unpack_from("<Q", raw, pos)
    pos += 8
    for i in range(n):
        vals = struct.unpack_from("<16f4Q", raw, pos)
        pos += struct.calcsize("<16f4Q")
        if i == frame:
            return pos + vals[18]
        pos += vals[18] + vals[19]
    raise IndexError(frame)


def test_headers_only_read(tmp_path):
    """want_depth=False: poses, timestamps and names as usual, no payload touched (a corrupt payload does not matter)."""
    rng = np.random.default_rng(4)
    depth = [rng.integers(0, 60000, (6, 8), dtype=np.uint16) for _ in range(11)]
    poses = [rng.normal(size=(4, 4)).astype(np.float32) for _ in range(11)]
    path = str(tmp_path / "h.sens")
    S.write_sens(path, np.eye(4, dtype=np.float32), poses, depth, color_hw=(12, 16))
    raw = bytearray(open(path, "rb").read())
    off = zlib_payload_offset(raw, 5)
    raw[off + 4:off + 9] = b"\x00" * 5
    open(path, "wb").write(raw)
    full_names = list(S.scene_info_entries("s", S.read_sens(path, want_depth=False), 5)["images_info"])
    thin = S.read_sens(path, keep_every=5, want_depth=False)
    assert thin.depth.shape == (0, 6, 8) and thin.frame_index == [0, 5, 10]
    info = S.scene_info_entries("s", thin, 5)
    assert list(info["images_info"]) == full_names == ["00000", "00005", "00010"]
    assert np.array_equal(info["images_info"]["00005"]["extrinsic_matrix"], S.text_roundtrip(poses[5]))
