"""The drop-in ``spatial_engine`` façade (reference signatures, HIP underneath) against the frozen
reference outputs in tests/golden/."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

from golden_util import GoldenScene, close_f64, same_f64

pytestmark = pytest.mark.gpu
PKG_ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-spatialmllm_amd")


def f64_ok(a, b):
    return same_f64(a, b) or close_f64(a, b, rtol=1e-12, scale=1e-6)


def facade():
    """Import the façade package (never the reference tree) even if something shadowed it earlier."""
    for name in [m for m in sys.modules if m == "spatial_engine" or m.startswith("spatial_engine.")]:
        if not getattr(sys.modules[name], "__file__", "").startswith(PKG_ROOT):
            del sys.modules[name]
    if sys.path[0] != PKG_ROOT:
        sys.path.insert(0, PKG_ROOT)
    ns = type("NS", (), {})()
    ns.IH = importlib.import_module("spatial_engine.utils.scannet_utils.handler.info_handler")
    ns.OPS = importlib.import_module("spatial_engine.utils.scannet_utils.handler.ops")
    ns.IMG = importlib.import_module("spatial_engine.utils.scannet_utils.handler._images")
    ns.CFR = importlib.import_module("spatial_engine.camera_movement.calculate_frames_relations")
    ns.MVI = importlib.import_module("spatial_engine.utils.scannet_utils.make_visibility_info")
    assert ns.IH.__file__.startswith(PKG_ROOT)
    return ns


@pytest.fixture(scope="module", params=["scene_ident", "scene_scaled"])
def setup(request, tmp_path_factory):
    ns = facade()
    g = GoldenScene(request.param)
    root = tmp_path_factory.mktemp(request.param)
    sid = "scene_golden_00"
    posed, inst = str(root / "posed_images"), str(root / "scannet_instance_data")
    os.makedirs(os.path.join(inst, sid))
    np.save(os.path.join(inst, sid, "aligned_points.npy"), g.points)
    H, W = g.color_hw
    for i in g.image_ids:
        col = g.color.get(i)
        ns.IMG.register(os.path.join(posed, sid, f"{i}.jpg"), col if col is not None else np.zeros((H, W, 3), np.uint8))
        ns.IMG.register(os.path.join(posed, sid, f"{i}.png"), g.depth[i])
    infos = {sid: {"num_posed_images": len(g.image_ids), "intrinsic_matrix": g.K, "axis_align_matrix": g.A,
                   "num_objects": 0, "images_info": {i: {"extrinsic_matrix": g.E[i]} for i in g.image_ids}}}
    h = ns.IH.SceneInfoHandler(infos, posed_images_root=posed, instance_data_root=inst)
    return ns, g, h, sid


def test_handler_projection_and_visibility(setup):
    ns, g, h, sid = setup
    assert h.get_all_extrinsic_valid_image_ids(sid) == g.valid_image_ids
    assert h.get_image_shape(sid) == g.color_hw and h.get_depth_image_shape(sid, g.valid_image_ids[0]) == g.depth_hw
    pts = h.get_scene_points_align(sid)[:, :3]
    for k, image_id in enumerate(g.valid_image_ids):
        uv, d = h.project_3d_point_to_image(sid, image_id, pts)
        assert uv.dtype == np.float64 and uv.shape == (len(pts), 2) and d.shape == (len(pts),)
        assert f64_ok(uv, g["ref_uv"][k]) and f64_ok(d, g["ref_depth"][k])
        vis = h.check_point_visibility(sid, image_id, g["ref_uv"][k], g["ref_depth"][k])
        assert vis.dtype == bool and np.array_equal(vis, g["ref_vis"][k])
        inb = h.check_point_in_image_boundary(sid, g["ref_uv"][k])
        byd = h.check_point_visibility_by_depth(sid, image_id, g["ref_uv"][k], g["ref_depth"][k])
        assert np.array_equal(inb & byd, g["ref_vis"][k])
        u = g["ref_uv"][k]
        assert np.array_equal(inb, (u[:, 0] >= 0) & (u[:, 0] < g.color_hw[1]) & (u[:, 1] >= 0) & (u[:, 1] < g.color_hw[0]))
    # single vertex (a6), (3,) input, both check_visible settings
    image_id = g.valid_image_ids[0]
    vis0 = g["ref_vis"][0]
    seen, unseen = int(np.where(vis0)[0][0]), int(np.where(~vis0)[0][0])
    uv, d = h.get_point_2d_coordinates_in_image(sid, image_id, seen, check_visible=True, return_depth=True)
    assert uv.shape == (1, 2) and f64_ok(uv[0], g["ref_uv"][0][seen]) and f64_ok(d[0], g["ref_depth"][0][seen])
    assert len(h.get_point_2d_coordinates_in_image(sid, image_id, unseen, check_visible=True)) == 0
    assert h.get_point_2d_coordinates_in_image(sid, image_id, unseen).shape == (1, 2)
    # free function
    uv, d = ns.IH.project_points(np.hstack([pts, np.ones((len(pts), 1))]), g.K, g.A @ g.E[image_id])
    assert f64_ok(uv, g["ref_uv"][0]) and f64_ok(d, g["ref_depth"][0])
    # any homogeneous coordinate is accepted, as upstream (IH:46-72): (2x, 2y, 2z, 2) is the same Euclidean point,
    # same pixel, twice the homogeneous depth (tests/test_gpu_boundary.py holds the general case against the oracle)
    uv2, d2 = ns.IH.project_points(np.hstack([2.0 * pts, np.full((len(pts), 1), 2.0)]), g.K, g.A @ g.E[image_id])
    assert np.allclose(uv2, uv, rtol=1e-12, equal_nan=True) and np.allclose(d2, 2.0 * d, rtol=1e-12, equal_nan=True)
    with pytest.raises(ValueError):
        ns.IH.project_points(pts, g.K, g.E[image_id])                    # [N, 3]: not homogeneous rows


def test_project_mask_to_3d(setup):
    ns, g, h, sid = setup
    fid = str(g["a7_frame"])
    color = g.color.get(fid)
    out = ns.OPS.project_mask_to_3d(g.depth[fid], g.K, g.E[fid], g["a7_mask"], g.A, color)
    assert out.dtype == np.float64 and out.shape == g["ref_a7"].shape and f64_ok(out, g["ref_a7"])
    out = ns.OPS.project_mask_to_3d(g.depth[fid], g.K, g.E[fid], g["a7_mask"])
    assert f64_ok(out, g["ref_a7_noalign"])
    with pytest.raises(AttributeError):
        ns.OPS.project_mask_to_3d(g.depth[fid], g.K, g.E[fid])
    if color is not None:    # path-based wrapper, mask=None => all pixels, with colour
        id1 = str(g["pair_ids"][0][0])
        out = h.project_image_to_3d_with_mask(sid, id1, None, with_color=True)
        assert f64_ok(out, g["pair0_xyzrgb"])


def test_scene_scripts(setup, tmp_path):
    ns, g, h, sid = setup
    warn = str(tmp_path / "warn.txt")
    s, table = ns.CFR.process_scene(sid, h, warn)
    keys = [tuple(str(x) for x in k) for k in g["cfr_pairs"]]
    assert s == sid and list(table.keys()) == keys
    got = np.array([[table[k][f] for f in ("overlap", "distance", "yaw", "pitch")] for k in keys])
    assert same_f64(got[:, 0], g["cfr_values"][:, 0]) and same_f64(got[:, 2:], g["cfr_values"][:, 2:])
    assert f64_ok(got[:, 1], g["cfr_values"][:, 1])
    s, vis = ns.MVI.process_scene(sid, h, warn)
    ref = g.json("mvi_json")
    assert vis["image_to_points"] == ref["image_to_points"]
    assert {str(k): v for k, v in vis["point_to_images"].items()} == ref["point_to_images"]
    assert all(isinstance(v, int) for v in next(iter(vis["image_to_points"].values()))[:3])
    # correspondence primitive == np.intersect1d of the reference's lists (VC_C:303)
    scene = h.scene_on_device(sid)
    a, b = keys[0]
    common = scene.common_visible_points(a, b)
    assert np.array_equal(common, np.intersect1d(ref["image_to_points"][a], ref["image_to_points"][b]))
    if len(common):
        uv1, _ = scene.point_2d_in_image(a, common[:4])
        k = g.valid_image_ids.index(a)
        assert f64_ok(uv1, g["ref_uv"][k][common[:4]])


def test_overlap_and_angles():
    ns = facade()
    rng = np.random.default_rng(0)
    a, b = rng.random(1000) < 0.3, rng.random(1000) < 0.3
    got = ns.CFR.calculate_camera_overlap({"x": a, "y": b}, "x", "y")
    assert same_f64(got, np.sum(a & b) / np.sum(a | b) * 100)
    z = np.zeros(77, dtype=bool)
    assert np.isnan(ns.CFR.calculate_camera_overlap({"x": z, "y": z}, "x", "y"))
    E = np.eye(4)
    E[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    zz = E[:3, 2]
    yaw, pitch = ns.CFR.extract_yaw_pitch(E)
    assert same_f64(yaw, np.degrees(np.arctan2(zz[1], zz[0])))
    assert same_f64(pitch, np.degrees(np.arcsin(zz[2] / np.linalg.norm(zz))))


def test_visibility_info_handler_roundtrip(setup, tmp_path):
    ns, g, h, sid = setup
    import pickle
    ref = g.json("mvi_json")
    try:
        import pyarrow  # noqa: F401  (absent on some GPU boxes; the pkl branch below still runs)
        have_parquet = True
    except ImportError:
        have_parquet = False
    if have_parquet:
        import pandas as pd
        rows = [(f"{sid}:image_to_points:{k}", json.dumps(v)) for k, v in ref["image_to_points"].items()]
        rows += [(f"{sid}:point_to_images:{k}", json.dumps(v)) for k, v in list(ref["point_to_images"].items())[:50]]
        path = str(tmp_path / "vis.parquet")
        pd.DataFrame(rows, columns=["key", "values"]).to_parquet(path, index=False)
        vh = ns.IH.VisibilityInfoHandler(path)
        k0 = next(iter(ref["image_to_points"]))
        assert vh.get_image_to_points_info(sid, k0) == ref["image_to_points"][k0]
        assert vh.get_point_to_images_info(sid, 3) == ref["point_to_images"]["3"]
        with pytest.raises(ValueError):
            vh.get_image_to_points_info(sid, "99999")
    ppath = str(tmp_path / "vis.pkl")
    with open(ppath, "wb") as f:
        pickle.dump({sid: {"image_to_points": ref["image_to_points"],
                           "point_to_images": {int(k): v for k, v in ref["point_to_images"].items()}}}, f)
    vp = ns.IH.VisibilityInfoHandler(ppath)
    assert vp.get_point_to_images_info(sid, 3) == ref["point_to_images"]["3"]
    with pytest.raises(ValueError):
        ns.IH.VisibilityInfoHandler(str(tmp_path / "vis.txt"))


def test_head_wrappers(setup, tmp_path):
    """Reference-signature record builders of the façade against the frozen reference answers."""
    import random
    ns, g, h, sid = setup
    CME = importlib.import_module("spatial_engine.camera_movement.camera_movement_engine_train_val")
    VC = importlib.import_module("spatial_engine.visual_correspondence.visual_correspondence_qa_engine_coor_2_coor")
    keys = [tuple(str(x) for x in k) for k in g["cfr_pairs"]]
    for n, ((id1, id2), vals, (yaw, pitch), ans) in enumerate(zip(keys, g["cfr_values"], g["cme_yaw_pitch"],
                                                                 g["cme_answers_json"])):
        row = {"scene_id": sid, "image_id1": id1, "image_id2": id2, "overlap": 20.0, "distance": float(vals[1]),
               "yaw": float(yaw), "pitch": float(pitch)}
        random.seed(1000 + n)                      # the seed oracle/gen_golden.py used for this row
        rec = CME.build_training_sample(h, row, n, "total_distance")
        ref = json.loads(str(ans))
        dv_ref, dv = ref.pop("displacement_vector"), rec["answer_values"].pop("displacement_vector")
        assert rec["answer_values"] == ref and f64_ok(dv, dv_ref)
        assert rec["gt_value"] == ref["total_distance"] and rec["id"] == n and len(rec["image"]) == 2
        assert rec["height_list"] == [g.color_hw[0]] * 2
    # visual correspondence through the façade == record stage fed with the reference's frozen projections
    from mspa import heads
    from mspa import templates as T
    mvi = g.json("mvi_json")
    vis_dict = {sid: {"image_to_points": mvi["image_to_points"]}}
    row = {"scene_id": sid, "image_id1": keys[0][0], "image_id2": keys[0][1]}
    random.seed(4)
    rec = VC.build_training_sample(h, row, 7, vis_dict, str(tmp_path / "w.txt"))
    common = np.intersect1d(mvi["image_to_points"][keys[0][0]], mvi["image_to_points"][keys[0][1]])
    random.seed(4)
    draw = heads.visual_correspondence_draws([row], [len(common)], T.VISUAL_CORRESPONDENCE)[0]
    if draw is None:
        assert rec is None
    else:
        first, second = (row["image_id2"], row["image_id1"]) if draw["swap"] else (row["image_id1"], row["image_id2"])
        v = int(common[draw["positions"][0]])
        k1, k2 = g.valid_image_ids.index(first), g.valid_image_ids.index(second)
        want = heads.visual_correspondence_record(row, 7, draw, g["ref_uv"][k1][[v]], g["ref_uv"][k2][[v]], g.color_hw)
        assert rec == want
    assert VC.build_training_sample(h, dict(row, scene_id="nope"), 0, vis_dict, str(tmp_path / "w.txt")) is None


def test_object_movement_wrapper():
    import random
    facade()
    OM = importlib.import_module("spatial_engine.object_movement.single_object_movement_engine_coord")
    from golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "tracks.npz"))
    eng = OM.TwoFrameVideoQAEngine("tapvid3d_displacement_vector", "adt")
    H, W = (int(v) for v in z["image_hw"])
    pairs = [{"frame1": int(a), "frame2": int(b), "point_index": int(p)} for a, b, p in z["pairs"]]
    recs = eng.format_training_samples(pairs, z["fx_fy_cx_cy"], "golden_tracks", z["ref_world"], z["tracks_XYZ"], H, W,
                                       z["extrinsics_w2c"])
    kept = [json.loads(str(r)) for r, k in zip(z["records_json"], z["kept"]) if k]
    assert len(recs) == len(kept)
    for r, ref in zip(recs, kept):
        assert tuple(r["p1"]) == tuple(ref["p1"]) and tuple(r["p2"]) == tuple(ref["p2"])
        assert r["point_moving"] == ref["point_moving"] and r["cam_moving"] == ref["cam_moving"]
        assert f64_ok(r["gt_value"], ref["gt_value"])
    assert eng.project_point(np.array([0.1, 0.2, -2.0]), z["fx_fy_cx_cy"], H, W) is None
    p = eng.project_point(np.array([0.1, 0.2, 2.0]), z["fx_fy_cx_cy"], H, W)
    fx, fy, cx, cy = z["fx_fy_cx_cy"]
    assert same_f64(p, [((fx * 0.1 / (2.0 + 1e-8)) + cx) / W, ((fy * 0.2 / (2.0 + 1e-8)) + cy) / H])


def test_torch_tensors_stay_on_the_device(setup):
    """the same entry points accept torch ROCm tensors and then return device tensors with the same values."""
    import torch
    ns, g, h, sid = setup
    image_id = g.valid_image_ids[0]
    pts = h.get_scene_points_align(sid)[:, :3]
    uv_np, d_np = h.project_3d_point_to_image(sid, image_id, pts)
    uv_t, d_t = h.project_3d_point_to_image(sid, image_id, torch.from_numpy(pts).cuda())
    assert isinstance(uv_t, torch.Tensor) and uv_t.is_cuda and d_t.is_cuda and uv_t.dtype == torch.float64
    assert np.array_equal(uv_t.cpu().numpy(), uv_np, equal_nan=True) and np.array_equal(d_t.cpu().numpy(), d_np, equal_nan=True)
    vis_np = h.check_point_visibility(sid, image_id, uv_np, d_np)
    vis_t = h.check_point_visibility(sid, image_id, uv_t, d_t)
    assert vis_t.dtype == torch.bool and vis_t.is_cuda and np.array_equal(vis_t.cpu().numpy(), vis_np)
    assert np.array_equal(h.check_point_in_image_boundary(sid, uv_t).cpu().numpy(), h.check_point_in_image_boundary(sid, uv_np))
    assert np.array_equal(h.check_point_visibility_by_depth(sid, image_id, uv_t, d_t).cpu().numpy(),
                          h.check_point_visibility_by_depth(sid, image_id, uv_np, d_np))
    hom = torch.cat([torch.from_numpy(pts).cuda(), torch.ones((len(pts), 1), dtype=torch.float64, device="cuda")], 1)
    uv2, _ = ns.IH.project_points(hom, g.K, h.get_extrinsic_matrix_align(sid, image_id))
    assert uv2.is_cuda and np.array_equal(uv2.cpu().numpy(), uv_np, equal_nan=True)
    uv3, _ = ns.IH.project_points(hom * 2.0, g.K, h.get_extrinsic_matrix_align(sid, image_id))   # w = 2: same pixels
    assert uv3.is_cuda and torch.allclose(uv3, uv2, rtol=1e-12, atol=0, equal_nan=True)


def test_object_visibility(setup):
    """compute_object_visibility.process_scene: masked popcount on the GPU == Python set intersections."""
    ns, g, h, sid = setup
    COV = importlib.import_module("spatial_engine.object_perception.compute_object_visibility")
    mvi = g.json("mvi_json")
    n = g.points.shape[0]
    rng = np.random.default_rng(1)
    inst = rng.integers(0, 6, n)                      # 0 = unlabelled, objects 0..4 -> mask values 1..5
    inst[rng.random(n) < 0.3] = 0
    np.save(os.path.join(h.instance_data_root, sid, "instance_mask.npy"), inst)
    h.infos[sid]["num_objects"] = 6                    # object 5 has no points, object 1 is a wall
    for o in range(6):
        h.infos[sid][o] = {"raw_category": "wall" if o == 1 else f"thing{o}"}
    vis_dict = {f"{sid}:image_to_points:{k}": json.dumps(v) for k, v in mvi["image_to_points"].items()}
    missing = g.valid_image_ids[-1]
    del vis_dict[f"{sid}:image_to_points:{missing}"]
    s, result, warnings = COV.process_scene(sid, h, vis_dict)
    # straightforward restatement of the reference loop (COVIS:103-150) with Python sets
    want = {"object_to_images": {}, "image_to_objects": {}}
    for o in range(6):
        if o == 1:
            continue
        pts = set(np.where(inst == o + 1)[0].tolist())
        if not pts:
            continue
        thr = max(1, int(0.05 * len(pts)))
        for image_id in g.valid_image_ids:
            if image_id == missing:
                continue
            c = len(set(mvi["image_to_points"][image_id]) & pts)
            if c >= thr:
                v = (c / len(pts)) * 100.0
                want["object_to_images"].setdefault(o, []).append({"image_id": image_id, "intersection_count": c, "visibility": v})
                want["image_to_objects"].setdefault(image_id, []).append({"object_id": o, "intersection_count": c, "visibility": v})
    assert s == sid and result == want and len(result["object_to_images"]) >= 3
    assert any("has no point indices" in w for w in warnings) and any("not found in visibility dict" in w for w in warnings)
    # split level: parquet index in, object_visibility.pkl + warning.txt out
    try:
        import pyarrow  # noqa: F401
        import pandas as pd
        import pickle
        import tempfile
        tmp = tempfile.mkdtemp(prefix="mspa_covis_")
        pq_path, info_path = os.path.join(tmp, "vis.parquet"), os.path.join(tmp, "infos.pkl")
        pd.DataFrame({"key": list(vis_dict), "values": list(vis_dict.values())}).to_parquet(pq_path)
        with open(info_path, "wb") as f:
            pickle.dump(h.infos, f)
        orig_init = ns.IH.SceneInfoHandler.__init__

        def patched(self, info_path_, *a, **k):
            orig_init(self, info_path_, posed_images_root=h.posed_images_root, instance_data_root=h.instance_data_root)
        ns.IH.SceneInfoHandler.__init__ = patched
        try:
            COV.process_split("val", info_path, pq_path, os.path.join(tmp, "out"))
        finally:
            ns.IH.SceneInfoHandler.__init__ = orig_init
        with open(os.path.join(tmp, "out", "object_visibility.pkl"), "rb") as f:
            assert pickle.load(f) == {sid: want}
        assert "has no point indices" in open(os.path.join(tmp, "out", "warning.txt")).read()
    except ImportError:
        pass
    # resident form straight from K1's bitsets
    scene = h.scene_on_device(sid)
    res2 = scene.object_visibility({o: np.where(inst == o + 1)[0] for o in range(5) if o != 1})
    full = {"object_to_images": {}, "image_to_objects": {}}
    for o in range(5):
        if o == 1:
            continue
        pts = set(np.where(inst == o + 1)[0].tolist())
        thr = max(1, int(0.05 * len(pts)))
        for image_id in g.valid_image_ids:
            c = len(set(mvi["image_to_points"][image_id]) & pts)
            if c >= thr:
                v = (c / len(pts)) * 100.0
                full["object_to_images"].setdefault(o, []).append({"image_id": image_id, "intersection_count": c, "visibility": v})
                full["image_to_objects"].setdefault(image_id, []).append({"object_id": o, "intersection_count": c, "visibility": v})
    assert res2 == full


def test_run_split_scripts(setup, tmp_path):
    """The two scene-precompute scripts end to end through their run_split entry points."""
    import pickle
    ns, g, h, sid = setup
    info_path = str(tmp_path / "infos.pkl")
    with open(info_path, "wb") as f:
        pickle.dump(h.infos, f)
    IH = ns.IH
    orig_init = IH.SceneInfoHandler.__init__

    def patched(self, info_path_, *a, **k):     # the scripts construct the handler with default roots
        orig_init(self, info_path_, posed_images_root=h.posed_images_root, instance_data_root=h.instance_data_root)
    IH.SceneInfoHandler.__init__ = patched
    try:
        vis = ns.MVI.run_split(info_path, str(tmp_path / "out" / "vis.pkl"), str(tmp_path / "w.txt"))
        ref = g.json("mvi_json")
        assert vis[sid]["image_to_points"] == ref["image_to_points"]
        with open(str(tmp_path / "out" / "vis.pkl"), "rb") as f:
            assert pickle.load(f)[sid]["image_to_points"] == ref["image_to_points"]
        vh = IH.VisibilityInfoHandler(str(tmp_path / "out" / "vis.pkl"))
        k0 = next(iter(ref["image_to_points"]))
        assert vh.get_image_to_points_info(sid, k0) == ref["image_to_points"][k0]
        try:
            import pyarrow  # noqa: F401
        except ImportError:
            return
        import pandas as pd
        # the visibility index streamed to parquet (one row group per scene), read back through the handler
        kept = ns.MVI.run_split(info_path, str(tmp_path / "out" / "vis.parquet"), str(tmp_path / "w.txt"), keep=False)
        assert kept == {}
        vp = IH.VisibilityInfoHandler(str(tmp_path / "out" / "vis.parquet"))
        for image_id, want_list in ref["image_to_points"].items():
            assert vp.get_image_to_points_info(sid, image_id) == want_list
        some = next(k for k, v in ref["point_to_images"].items() if v)
        assert vp.get_point_to_images_info(sid, int(some)) == ref["point_to_images"][some]
        table = ns.CFR.run_split(info_path, str(tmp_path / "out" / "pairs.parquet"), str(tmp_path / "w.txt"))
        df = pd.read_parquet(str(tmp_path / "out" / "pairs.parquet"))
        keys = [tuple(str(x) for x in k) for k in g["cfr_pairs"]]
        assert list(zip(df["image_id1"], df["image_id2"])) == keys and len(table[sid]) == len(keys)
        assert same_f64(df["overlap"].to_numpy(), g["cfr_values"][:, 0])
        nz = pd.read_parquet(str(tmp_path / "out" / "pairs_nonzero.parquet"))
        assert len(nz) == int((g["cfr_values"][:, 0] != 0).sum())
    finally:
        IH.SceneInfoHandler.__init__ = orig_init


def test_rigid_body_segmentation():
    """K7 + SciPy linkage == the reference's groups (tests/golden/tracks.npz) and its loss matrix."""
    facade()
    OM = importlib.import_module("spatial_engine.object_movement.single_object_movement_engine_coord")
    from golden_util import GOLDEN_DIR
    from scipy.spatial.distance import pdist, squareform
    import torch
    from mspa import engine
    z = np.load(os.path.join(GOLDEN_DIR, "tracks.npz"))
    pts = z["tracks_XYZ"]
    groups = OM.rigid_body_segmentation(pts)
    assert groups == json.loads(str(z["groups_json"]))
    assert OM.filter_large_groups([[1] * 6, [2] * 5]) == [[1] * 6]
    loss = engine.track_rigidity_loss(torch.from_numpy(np.ascontiguousarray(pts)).cuda()).cpu().numpy()
    ref = np.zeros_like(loss)
    for t in range(1, pts.shape[0]):                       # the reference's accumulation, restated with SciPy
        ch = np.abs(squareform(pdist(pts[t])) - squareform(pdist(pts[t - 1])))
        ref += np.where(ch > 0.01, ch, 0)
    assert np.allclose(loss, ref, rtol=1e-12, atol=1e-15) and np.array_equal(loss, loss.T) and (np.diag(loss) == 0).all()


def _fake_jpeg(h, w):
    """Smallest byte string with a JPEG start-of-frame segment carrying (h, w) -- after an APP0 segment to skip."""
    app0 = b"\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    sof = b"\xff\xc0\x00\x11\x08" + h.to_bytes(2, "big") + w.to_bytes(2, "big") + b"\x03\x01\x11\x00\x02\x11\x01\x03\x11\x01"
    return b"\xff\xd8" + app0 + sof + b"\xff\xd9"


def test_object_movement_scene_level(tmp_path):
    """npz in, records out (OM_C.generate_qa_training_single_scene): K5a world transform, K7 grouping, K5c pair mining,
    K5b records == the host chain with NumPy distances; K5c distances are bit-identical to np.linalg.norm."""
    facade()
    import random
    import torch
    from mspa import engine, heads, synth
    OM = importlib.import_module("spatial_engine.object_movement.single_object_movement_engine_coord")
    tr = synth.make_tracks(33, T=150, P=64, n_groups=4)
    H, W = tr.image_hw
    assert OM.jpeg_size(_fake_jpeg(H, W)) == (H, W)
    path = str(tmp_path / f"{tr.scene_id}.npz")
    np.savez(path, images_jpeg_bytes=np.array([_fake_jpeg(H, W)] * tr.tracks_XYZ.shape[0], dtype=object),
             tracks_XYZ=tr.tracks_XYZ, visibility=tr.visibility, fx_fy_cx_cy=tr.fx_fy_cx_cy, extrinsics_w2c=tr.extrinsics_w2c)
    # K5c against NumPy, bit for bit
    T_, P_ = tr.tracks_XYZ.shape[:2]
    c2w = np.linalg.inv(tr.extrinsics_w2c)
    world_np = np.einsum("nij,nkj->nki", c2w, np.concatenate([tr.tracks_XYZ, np.ones((T_, P_, 1))], axis=2))[..., :3]
    world = engine.track_to_world(torch.from_numpy(np.ascontiguousarray(tr.tracks_XYZ)).cuda(),
                                  torch.from_numpy(c2w.reshape(T_, 16)).cuda(), tr.fx_fy_cx_cy, (H, W), ("world",))["world"]
    world_host = world.cpu().numpy()
    pts = [3, 40, 41, 7]
    frames = [np.where(tr.visibility[:, p])[0] for p in pts]
    frames[3] = frames[3][:1]                                   # a point with a single visible frame: no pairs
    got = engine.track_pair_distances(world, pts, frames)
    for p, fr, d in zip(pts, frames, got):
        ii, jj = np.triu_indices(len(fr), 1)
        want = np.linalg.norm(world_host[fr[jj], p] - world_host[fr[ii], p], axis=1)
        assert d.shape == want.shape and np.array_equal(d, want)
    assert len(got[3]) == 0 and len(got[0]) > 100
    with pytest.raises(ValueError):
        engine.track_pair_distances(world, [P_], [np.array([0, 1])])
    # the whole scene through the façade
    eng = OM.TwoFrameVideoQAEngine("tapvid3d_displacement_vector", "adt")
    random.seed(23)
    recs = eng.generate_qa_training_single_scene(path, 6, 5, str(tmp_path / "img"), True, 0.5)
    end_state = random.getstate()
    assert sorted(os.listdir(tmp_path / "img" / tr.scene_id))[:2] == ["00000.jpg", "00001.jpg"]
    assert open(tmp_path / "img" / tr.scene_id / "00003.jpg", "rb").read() == _fake_jpeg(H, W)
    groups = OM.filter_large_groups(OM.rigid_body_segmentation(tr.tracks_XYZ), min_size=5)

    def distance_fn(points, frs):
        return [np.linalg.norm(world_host[fr[np.triu_indices(len(fr), 1)[1]], p] - world_host[fr[np.triu_indices(len(fr), 1)[0]], p],
                               axis=1) for p, fr in zip(points, frs)]
    random.seed(23)
    pairs = heads.object_movement_mine_pairs(tr.visibility, groups, distance_fn, 6, 5, True, 0.5)
    want = heads.object_movement_records(tr.scene_id, tr.tracks_XYZ, tr.extrinsics_w2c, tr.fx_fy_cx_cy, (H, W), pairs,
                                         "tapvid3d_displacement_vector", eng.templates, random)
    assert recs == want and len(recs) > 20 and random.getstate() == end_state
    # dataset level: JSONL + eval form
    (tmp_path / "src").mkdir()
    os.rename(path, tmp_path / "src" / f"{tr.scene_id}.npz")
    out = str(tmp_path / "val.jsonl")
    random.seed(24)
    eng.generate_qa_eval_data([tr.scene_id], str(tmp_path / "src"), str(tmp_path), out, str(tmp_path / "img"), 1, 1, False,
                              max_samples=10)
    lines = [json.loads(line) for line in open(out)]
    assert 0 < len(lines) <= 10 and all("text" in r for r in lines)


def test_scene_prefetcher_matches_direct_upload():
    """mspa.upload.ScenePrefetcher (pinned staging, copy stream, recycled slots) hands out scenes whose products equal those of
    the plain SceneOnDevice constructor -- threaded and unthreaded, scenes of different sizes through the same two slots."""
    from mspa import synth, upload
    from mspa.scene import SceneOnDevice
    scs = [synth.make_scene(3100 + k, n_points=2000 + 700 * k, n_frames=3 + 2 * k, color_hw=(48, 64), depth_hw=(48, 64),
                            invalid_pose_frac=0.2 if k == 1 else 0.0, with_color=False) for k in range(4)]
    want = []
    for sc in scs:
        d = SceneOnDevice(sc.K, sc.A, sc.E, sc.depth, sc.color_hw, sc.points, "cuda")
        want.append((d.ids, d.frames_relations_arrays(), d.visibility_csr()))
    for threaded in (False, True):
        got = []
        for scene in upload.ScenePrefetcher(scs, "cuda", threaded=threaded):
            got.append((scene.ids, scene.frames_relations_arrays(), scene.visibility_csr()))
        assert len(got) == len(want)
        for (ids, rel, csr), (ids0, rel0, csr0) in zip(got, want):
            assert ids == ids0
            for k in rel0:
                assert np.array_equal(rel[k], rel0[k], equal_nan=True), k
            assert np.array_equal(csr.i2p_indices, csr0.i2p_indices) and np.array_equal(csr.p2i_offsets, csr0.p2i_offsets)
    # a scene whose pose is not affine fails in the staging thread: the error reaches the consumer, the slots go back to the
    # pool intact (the depth helper is waited for) and the next prefetcher works
    import copy
    bad = copy.copy(scs[2])
    bad.E = {k: v.copy() for k, v in scs[2].E.items()}
    bad.E[scs[2].valid_image_ids[1]][3, 1] = 1e-3
    with pytest.raises(ValueError, match="affine|last row|E"):
        for _scene in upload.ScenePrefetcher([scs[0], bad, scs[3]], "cuda"):
            pass
    again = [s.ids for s in upload.ScenePrefetcher([scs[0], scs[3]], "cuda")]
    assert again == [want[0][0], want[3][0]]


@pytest.mark.gpu
def test_host_pose_prep_bitwise_on_this_host():
    """The same bit-identity check as tests/test_host_cpu.py, repeated on the GPU box's host CPU: NumPy picks its ufunc
    loops (SVML / AVX-512) per machine, and the vectorised preparation must equal the per-frame form on each."""
    from test_host_cpu import check_host_pose_prep_bitwise
    check_host_pose_prep_bitwise()
