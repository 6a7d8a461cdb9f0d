"""oracle/np_oracle.py against the imported, unmodified reference (build container only).

Every comparison is bit-for-bit: both sides run the same NumPy/BLAS in the same process, so the
restatement must reproduce the reference's float64 results exactly, not approximately.
"""
import random

import numpy as np
import pytest

from oracle import np_oracle as O
from oracle import ref_harness as RH
from mspa import synth

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not RH.reference_available(), reason="/root/reference not mounted")]


@pytest.fixture(scope="module")
def ref():
    return RH.import_reference()


@pytest.fixture(scope="module", params=[((480, 640), (480, 640)), ((968, 1296), (480, 640))],
                ids=["c640x480", "c1296x968"])
def scene_and_handler(ref, request):
    color_hw, depth_hw = request.param
    sc = synth.make_scene(1000, n_points=6000, n_frames=7, color_hw=color_hw, depth_hw=depth_hw,
                          invalid_pose_frac=0.15, with_color=(color_hw == (480, 640)))
    return sc, RH.make_handler(ref, [sc])


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.int64)


def same_f64(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def test_project_points_and_visibility(ref, scene_and_handler):
    sc, h = scene_and_handler
    sid = sc.scene_id
    assert h.get_all_extrinsic_valid_image_ids(sid) == O.valid_image_ids(sc.E) == sc.valid_image_ids
    assert len(sc.valid_image_ids) < len(sc.image_ids)
    pts = sc.points[:, :3]
    for image_id in sc.valid_image_ids:
        uv_r, d_r = h.project_3d_point_to_image(sid, image_id, pts)
        m_r = h.check_point_visibility(sid, image_id, uv_r, d_r)
        m_o, uv_o, d_o = O.vertex_visibility(pts, sc.K, O.aligned_extrinsic(sc.A, sc.E[image_id]),
                                             sc.depth[image_id], sc.color_hw)
        assert same_f64(uv_r, uv_o) and same_f64(d_r, d_o)
        assert np.array_equal(m_r, m_o)
        assert np.array_equal(h.check_point_in_image_boundary(sid, uv_r),
                              O.check_point_in_image_boundary(uv_o, sc.color_hw))
        assert np.array_equal(h.check_point_visibility_by_depth(sid, image_id, uv_r, d_r),
                              O.check_point_visibility_by_depth(uv_o, d_o, sc.depth[image_id], sc.color_hw))
    # single vertex form (a6)
    uv_r, d_r = h.get_point_2d_coordinates_in_image(sid, sc.valid_image_ids[0], 17, align=True,
                                                    check_visible=False, return_depth=True)
    uv_o, d_o = O.point_2d_in_image(sc.points[17], sc.K, O.aligned_extrinsic(sc.A, sc.E[sc.valid_image_ids[0]]),
                                    sc.depth[sc.valid_image_ids[0]], sc.color_hw, check_visible=False)
    assert same_f64(uv_r, uv_o) and same_f64(d_r, d_o)


def test_project_mask_to_3d(ref, scene_and_handler):
    sc, h = scene_and_handler
    image_id = sc.valid_image_ids[1]
    rng = np.random.default_rng(3)
    H, W = sc.color_hw
    mask = rng.random((H, W)) < 0.05
    color = sc.color.get(image_id)
    r = ref.OPS.project_mask_to_3d(sc.depth[image_id], sc.K, sc.E[image_id], mask, sc.A, color)
    o = O.project_mask_to_3d(sc.depth[image_id], sc.K, sc.E[image_id], mask, sc.A, color)
    assert same_f64(r, o)
    r = ref.OPS.project_mask_to_3d(sc.depth[image_id], sc.K, sc.E[image_id], mask)
    o = O.project_mask_to_3d(sc.depth[image_id], sc.K, sc.E[image_id], mask)
    assert same_f64(r, o)
    if color is not None:
        r = h.project_image_to_3d_with_mask(sc.scene_id, image_id, None, with_color=True)
        o = O.project_mask_to_3d(sc.depth[image_id], sc.K, sc.E[image_id], None, sc.A, color)
        assert same_f64(r, o)
    with pytest.raises(AttributeError):
        O.project_mask_to_3d(sc.depth[image_id], sc.K, sc.E[image_id])
    with pytest.raises(AttributeError):
        ref.OPS.project_mask_to_3d(sc.depth[image_id], sc.K, sc.E[image_id])


def test_frame_pair_composite(ref, scene_and_handler):
    sc, h = scene_and_handler
    if not sc.color:
        pytest.skip("composite needs a colour image for mask=None")
    id1, id2 = sc.valid_image_ids[0], sc.valid_image_ids[2]
    pts = h.project_image_to_3d_with_mask(sc.scene_id, id1, None, with_color=True)
    uv, d = h.project_3d_point_to_image(sc.scene_id, id2, pts[:, :3])
    vis = h.check_point_visibility(sc.scene_id, id2, uv, d)
    o = O.frame_pair(sc.depth[id1], sc.depth[id2], sc.K, sc.E[id1], sc.E[id2], sc.A, sc.color_hw, sc.color[id1])
    v = o["valid"]
    assert same_f64(o["xyz"][v], pts[:, :3]) and same_f64(o["uv2"][v], uv) and same_f64(o["depth2"][v], d)
    assert np.array_equal(o["vis"][v], vis) and not o["vis"][~v].any()
    assert np.array_equal(o["rgb"][v].astype(np.float64), pts[:, 3:6])


def test_frames_relations_and_visibility_index(ref, scene_and_handler, tmp_path):
    sc, h = scene_and_handler
    sid, table_r = ref.CFR.process_scene(sc.scene_id, h, str(tmp_path / "w.txt"))
    table_o = O.frames_relations_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
    assert list(table_r.keys()) == list(table_o.keys())
    for k in table_r:
        for f in ("overlap", "distance", "yaw", "pitch"):
            assert same_f64(table_r[k][f], table_o[k][f]), (k, f)
    sid, vis_r = ref.MVI.process_scene(sc.scene_id, h, str(tmp_path / "w2.txt"))
    vis_o = O.visibility_index_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
    assert vis_r == vis_o


def test_overlap_nan_and_yaw_pitch(ref):
    a = np.zeros(50, dtype=bool)
    with np.errstate(all="ignore"):
        r = ref.CFR.calculate_camera_overlap({"a": a, "b": a}, "a", "b")
    assert np.isnan(r) and np.isnan(O.calculate_camera_overlap(a, a))
    rng = np.random.default_rng(0)
    for _ in range(20):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        E = np.eye(4)
        E[:3, :3] = q
        assert same_f64(ref.CFR.extract_yaw_pitch(E), O.extract_yaw_pitch(E))


def test_relative_pose(ref, scene_and_handler, tmp_path):
    sc, h = scene_and_handler
    table = O.frames_relations_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
    for n, ((id1, id2), row) in enumerate(table.items()):
        row_d = {"scene_id": sc.scene_id, "image_id1": id1, "image_id2": id2, "overlap": 20.0,
                 "yaw": row["yaw"] + (300 if n % 3 == 0 else 0), "pitch": row["pitch"], "distance": row["distance"]}
        random.seed(n)
        swap = random.random() < 0.5
        random.seed(n)
        r = ref.CME.build_training_sample(h, row_d, n, "displacement_vector")
        E1, E2 = O.aligned_extrinsic(sc.A, sc.E[id1]), O.aligned_extrinsic(sc.A, sc.E[id2])
        o = O.relative_pose_answer_values(E1, E2, row_d["yaw"], row_d["pitch"], swap)
        assert r["answer_values"] == o


def test_tracks(ref):
    tr = synth.make_tracks(7, T=20, P=40)
    eng = ref.OM_C.TwoFrameVideoQAEngine("tapvid3d_total_distance", "adt")
    world_o = O.tracks_cam_to_world(tr.tracks_XYZ, tr.extrinsics_w2c)
    c2w = np.linalg.inv(tr.extrinsics_w2c)
    hom = np.concatenate([tr.tracks_XYZ, np.ones(tr.tracks_XYZ.shape[:2] + (1,))], axis=2)
    assert same_f64(world_o, np.einsum('nij,nkj->nki', c2w, hom)[..., :3])
    H, W = tr.image_hw
    pairs = [{"frame1": f1, "frame2": f2, "point_index": p}
             for f1, f2, p in [(0, 5, 3), (2, 19, 30), (7, 7, 1), (4, 9, 39), (10, 3, 22)]]
    random.seed(0)
    recs = eng.format_training_samples(pairs, tr.fx_fy_cx_cy, tr.scene_id, world_o, tr.tracks_XYZ, H, W,
                                       tr.extrinsics_w2c)
    got = [O.object_displacement(world_o, tr.tracks_XYZ, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                 p["frame1"], p["frame2"], p["point_index"]) for p in pairs]
    got = [g for g in got if g is not None]
    assert len(got) == len(recs)
    for r, g in zip(recs, got):
        assert tuple(r["p1"]) == g["p1"] and tuple(r["p2"]) == g["p2"]
        assert r["gt_value"] == g["gt_total_distance"]
        assert r["point_moving"] == g["point_moving"] and r["cam_moving"] == g["cam_moving"]
    for pt in [np.array([0.1, 0.2, 2.0]), np.array([0.1, 0.2, -2.0]), np.array([9.0, 0.0, 1.0]), np.zeros(3)]:
        assert eng.project_point(pt, tr.fx_fy_cx_cy, H, W) == O.project_point(pt, tr.fx_fy_cx_cy, H, W)


def test_general_homogeneous_points_and_depth_scale(ref, scene_and_handler):
    """The two open parameters of the interface (missing 4): project_points on [N, 4] rows with w != 1
    (IH:46-72) and a handler built with another depth_value_scale (IH:76 -> IH:368)."""
    sc, h = scene_and_handler
    sid = sc.scene_id
    rng = np.random.default_rng(11)
    i = sc.valid_image_ids[0]
    E = O.aligned_extrinsic(sc.A, sc.E[i])
    w = rng.uniform(0.25, 4.0, sc.points.shape[0])
    w[::5] = 1.0
    w[1::131] = 0.0
    pts = np.hstack([sc.points[:, :3] * w[:, None], w[:, None]])
    with np.errstate(all="ignore"):
        uv_r, d_r = ref.IH.project_points(pts, sc.K, E)
        uv_o, d_o = O.project_points(pts, sc.K, E)
    assert same_f64(uv_r, uv_o) and same_f64(d_r, d_o)
    h.depth_value_scale = 0.0005                                          # what SceneInfoHandler(depth_value_scale=...) stores
    try:
        uv, d = h.project_3d_point_to_image(sid, i, sc.points[:, :3])
        m_r = h.check_point_visibility(sid, i, uv, d)
        assert np.array_equal(m_r, O.check_point_visibility(uv, d, sc.depth[i], sc.color_hw, 0.0005))
        assert not np.array_equal(m_r, O.check_point_visibility(uv, d, sc.depth[i], sc.color_hw))
    finally:
        h.depth_value_scale = 0.001
