"""TEST INFRASTRUCTURE (build container only): import the *unmodified* reference under stubs.

The reference (facebookresearch/Multi-SpatialMLLM, mounted read-only at /root/reference) needs
``cv2``, ``mmengine`` and ``open3d``, none of which exist in this image.  This module pre-seeds
``sys.modules`` with minimal stand-ins -- an in-memory ``cv2.imread`` keyed by path, an
``mmengine.load`` that returns a synthetic scene-info dict, a no-op ``TimeCounter`` -- and then
imports the reference modules themselves.  It is used for two things only:

* ``oracle/gen_golden.py`` -- emit the golden vectors committed under ``tests/golden/``;
* ``tests/test_oracle_vs_reference.py`` -- check ``oracle/np_oracle.py`` function by function.

/root/reference does not exist on the GPU box; everything here is skipped there.  Nothing under
the product package may import this file.
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import types
from typing import Dict

import numpy as np

REFERENCE_ROOT = os.environ.get("MSPA_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "spatial_engine"))


class _ImageStore:
    """Path -> ndarray table served by the stub ``cv2.imread``."""

    def __init__(self):
        self.images: Dict[str, np.ndarray] = {}
        self.written: Dict[str, np.ndarray] = {}

    def clear(self):
        self.images.clear()
        self.written.clear()


STORE = _ImageStore()
_INFOS: Dict[str, dict] = {}   # info_path -> infos dict served by the stub mmengine.load


def _install_stubs():
    if "cv2" in sys.modules and getattr(sys.modules["cv2"], "_mspa_stub", False):
        return
    cv2 = types.ModuleType("cv2")
    cv2._mspa_stub = True
    cv2.IMREAD_UNCHANGED = -1
    cv2.COLOR_BGR2RGB = 4
    cv2.FONT_HERSHEY_SIMPLEX = 0
    cv2.LINE_AA = 16
    cv2.CV_64F = 6

    def imread(path, flags=1):
        img = STORE.images.get(path)
        if img is None:
            return None
        if path.endswith(".png") and flags != -1 and img.ndim == 2:
            # cv2.imread without IMREAD_UNCHANGED turns a 16-bit PNG into 8-bit BGR; only .shape is used
            return np.zeros(img.shape + (3,), dtype=np.uint8)
        return img

    def cvtColor(img, code):
        return img[..., ::-1]   # BGR <-> RGB

    def imwrite(path, img):
        STORE.written[path] = img
        return True

    def _noop(img, *a, **k):
        return img

    cv2.imread, cv2.cvtColor, cv2.imwrite = imread, cvtColor, imwrite
    cv2.circle = cv2.putText = cv2.line = cv2.rectangle = _noop
    cv2.getTextSize = lambda *a, **k: ((10, 10), 2)
    sys.modules["cv2"] = cv2

    mmengine = types.ModuleType("mmengine")

    def load(path, *a, **k):
        if path in _INFOS:
            return _INFOS[path]
        raise FileNotFoundError(path)

    mmengine.load = load
    mmengine.dump = lambda obj, path, *a, **k: _INFOS.__setitem__(path, obj)
    mmengine.mkdir_or_exist = lambda p: os.makedirs(p, exist_ok=True)
    mmengine.list_from_file = lambda p: [ln.rstrip("\n") for ln in open(p)]
    utils = types.ModuleType("mmengine.utils")
    dl_utils = types.ModuleType("mmengine.utils.dl_utils")

    class TimeCounter:
        def __init__(self, *a, **k):
            pass

        def __call__(self, fn):
            return fn

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    dl_utils.TimeCounter = TimeCounter
    utils.dl_utils = dl_utils
    mmengine.utils = utils
    sys.modules["mmengine"] = mmengine
    sys.modules["mmengine.utils"] = utils
    sys.modules["mmengine.utils.dl_utils"] = dl_utils
    sys.modules["open3d"] = types.ModuleType("open3d")
    mmengine.exists = os.path.exists
    # imageio is only touched by extract_posed_images.py: payloads pass through, written arrays are captured
    imageio = types.ModuleType("imageio")
    v2 = types.ModuleType("imageio.v2")
    v2.imread = lambda data, *a, **k: np.frombuffer(data, dtype=np.uint8)
    v2.imwrite = lambda path, arr, *a, **k: STORE.written.__setitem__(path, np.array(arr))
    imageio.v2 = v2
    sys.modules["imageio"] = imageio
    sys.modules["imageio.v2"] = v2


def import_reference():
    """Return a namespace holding the reference modules on the hot path."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    # the façade package in this repo is also called ``spatial_engine``: make sure the reference
    # wins inside this process and that a previously imported façade is not reused.
    for name in [m for m in sys.modules if m == "spatial_engine" or m.startswith("spatial_engine.")]:
        del sys.modules[name]
    for p in (os.path.join(REFERENCE_ROOT, "spatial_engine", "camera_movement"), REFERENCE_ROOT):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    # The reference tree has no __init__.py files (namespace package); a regular package of the same
    # name anywhere on sys.path would shadow it, so the façade's root is taken off the path while the
    # reference modules are imported (they stay in sys.modules under their own names afterwards).
    facade_roots = [q for q in sys.path
                    if os.path.isfile(os.path.join(q or ".", "spatial_engine", "__init__.py"))]
    saved_path = list(sys.path)
    sys.path[:] = [q for q in sys.path if q not in facade_roots]
    try:
        return _import_reference_modules()
    finally:
        sys.path[:] = saved_path


def _import_reference_modules():
    ns = types.SimpleNamespace()
    ns.IH = importlib.import_module("spatial_engine.utils.scannet_utils.handler.info_handler")
    ns.OPS = importlib.import_module("spatial_engine.utils.scannet_utils.handler.ops")
    ns.CFR = importlib.import_module("spatial_engine.camera_movement.calculate_frames_relations")
    ns.MVI = importlib.import_module("spatial_engine.utils.scannet_utils.make_visibility_info")
    ns.CME = importlib.import_module("spatial_engine.camera_movement.camera_movement_engine_train_val")
    ns.OM_C = importlib.import_module("spatial_engine.object_movement.single_object_movement_engine_coord")
    ns.VC_C = importlib.import_module(
        "spatial_engine.visual_correspondence.visual_correspondence_qa_engine_coor_2_coor")
    ns.DE_C = importlib.import_module("spatial_engine.depth_perception.depth_estimation_coor_engine")
    ns.DC_C = importlib.import_module("spatial_engine.depth_perception.depth_comparison_coor_engine")
    ns.DE_D = importlib.import_module("spatial_engine.depth_perception.depth_estimation_dot_engine")
    ns.DC_D = importlib.import_module("spatial_engine.depth_perception.depth_comparison_dot_engine")
    ns.VC_D = importlib.import_module(
        "spatial_engine.visual_correspondence.visual_correspondence_qa_engine_dot_2_multichoice")
    ns.OM_D = importlib.import_module("spatial_engine.object_movement.single_object_movement_engine_dot")
    assert ns.IH.__file__.startswith(REFERENCE_ROOT), ns.IH.__file__
    return ns


def import_sens(ns):
    """extract_posed_images.py (the .sens reader) under the imageio stub."""
    ns.SENS = importlib.import_module("spatial_engine.utils.scannet_utils.extract_posed_images")
    assert ns.SENS.__file__.startswith(REFERENCE_ROOT), ns.SENS.__file__
    return ns


def import_object_perception(ns):
    """Add the object_perception scripts to the namespace (their imports reseed ``random``: COV seeds 0,
    OPE seeds 1 -- callers reseed before every use)."""
    ns.COVIS = importlib.import_module("spatial_engine.object_perception.compute_object_visibility")
    ns.COV = importlib.import_module("spatial_engine.object_perception.single_object_coverage_finder")
    ns.OPE = importlib.import_module("spatial_engine.object_perception.single_object_perception_engine")
    assert ns.COV.__file__.startswith(REFERENCE_ROOT), ns.COV.__file__
    return ns


def make_handler(ns, scenes, root=None):
    """Build a reference ``SceneInfoHandler`` over synthetic scenes (mspa.synth.SynthScene)."""
    root = root or tempfile.mkdtemp(prefix="mspa_ref_")
    posed = os.path.join(root, "posed_images")
    inst = os.path.join(root, "scannet_instance_data")
    infos = {}
    for sc in scenes:
        infos[sc.scene_id] = sc.info_dict()
        os.makedirs(os.path.join(inst, sc.scene_id), exist_ok=True)
        np.save(os.path.join(inst, sc.scene_id, "aligned_points.npy"), sc.points)
        H, W = sc.color_hw
        for image_id in sc.image_ids:
            col = sc.color.get(image_id)
            if col is None:
                col = np.zeros((H, W, 3), dtype=np.uint8)
            STORE.images[os.path.join(posed, sc.scene_id, f"{image_id}.jpg")] = col[..., ::-1]  # BGR on disk
            STORE.images[os.path.join(posed, sc.scene_id, f"{image_id}.png")] = sc.depth[image_id]
    info_path = os.path.join(root, "infos.pkl")
    _INFOS[info_path] = infos
    handler = ns.IH.SceneInfoHandler(info_path, posed_images_root=posed, instance_data_root=inst)
    handler._mspa_info_path, handler._mspa_root = info_path, root
    return handler


def register_pickle(path, obj):
    """Make ``mmengine.load(path)`` (stub) return ``obj`` -- e.g. a visibility-info dict."""
    _INFOS[path] = obj
