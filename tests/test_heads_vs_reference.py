"""Record stage of the task heads against the imported reference (build container only).

The record stage is pure Python: handed the reference's own template tables, the same ``random`` seed
and numerics from the oracle, it must reproduce the reference's records exactly -- ids, image lists,
conversation text, integer formatting, answer fields.  (The GPU numeric stages are checked on the
MI355X in tests/test_gpu_heads.py.)
"""
import os
import random

import numpy as np
import pytest

from mspa import heads, synth
from mspa import templates as T
from oracle import np_oracle as O
from oracle import ref_harness as RH

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not RH.reference_available(), reason="/root/reference not mounted")]


@pytest.fixture(scope="module")
def ref():
    return RH.import_reference()


@pytest.fixture(scope="module")
def world(ref):
    sc = synth.make_scene(3001, n_points=3000, n_frames=7, color_hw=(96, 128), depth_hw=(96, 128),
                          invalid_pose_frac=0.15, with_color=False)
    h = RH.make_handler(ref, [sc])
    h.get_image_size = h.get_image_shape          # upstream defect: VC_C calls a missing name
    table = O.frames_relations_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
    vis = O.visibility_index_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
    rows = [{"scene_id": sc.scene_id, "image_id1": a, "image_id2": b, "overlap": max(float(v["overlap"]), 1.0),
             "distance": float(v["distance"]), "yaw": float(v["yaw"]), "pitch": float(v["pitch"])}
            for (a, b), v in table.items()]
    return sc, h, rows, vis


def test_camera_movement_records(ref, world):
    sc, h, rows, _ = world
    tpl = T.TemplateSet.from_module(ref.CME)
    for qt in T.CAMERA_MOVEMENT_TYPES:
        assert qt in tpl.questions and qt in tpl.answers
    for n, row in enumerate(rows):
        qt = T.CAMERA_MOVEMENT_TYPES[n % len(T.CAMERA_MOVEMENT_TYPES)]
        row = dict(row, yaw=row["yaw"] + (300.0 if n % 4 == 0 else 0.0))
        E1 = sc.A @ sc.E[row["image_id1"]]
        E2 = sc.A @ sc.E[row["image_id2"]]
        t12 = (np.linalg.inv(E1) @ E2)[:3, 3]
        t21 = (np.linalg.inv(E2) @ E1)[:3, 3]
        random.seed(50 + n)
        want = ref.CME.build_training_sample(h, row, n, qt)
        random.seed(50 + n)
        got = heads.camera_movement_record(row, n, qt, t12, t21, sc.color_hw, tpl)
        assert got == want
    ev = heads.to_eval_sample(dict(got))
    assert "conversations" not in ev and ev["text"] == want["conversations"][0]["value"]


def test_visual_correspondence_records(ref, world, tmp_path):
    sc, h, rows, vis = world
    tpl = T.TemplateSet.from_module(ref.VC_C, ["default"])
    vis_dict = {sc.scene_id: vis}
    warn = str(tmp_path / "w.txt")
    random.seed(7)
    want = [ref.VC_C.build_training_sample(h, row, n, vis_dict, warn) for n, row in enumerate(rows)]
    n_common = [len(np.intersect1d(vis["image_to_points"][r["image_id1"]], vis["image_to_points"][r["image_id2"]]))
                for r in rows]
    random.seed(7)
    draws = heads.visual_correspondence_draws(rows, n_common, tpl)
    assert any(d is None for d in draws) == any(w is None for w in want)
    for n, (row, dr, w) in enumerate(zip(rows, draws, want)):
        if dr is None:
            assert w is None
            continue
        first, second = (row["image_id2"], row["image_id1"]) if dr["swap"] else (row["image_id1"], row["image_id2"])
        common = np.intersect1d(vis["image_to_points"][first], vis["image_to_points"][second])
        verts = [int(common[j]) for j in dr["positions"]]
        uv1 = np.stack([O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[first], sc.depth[first], sc.color_hw)[0][0]
                        for v in verts])
        uv2 = np.stack([O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[second], sc.depth[second], sc.color_hw)[0][0]
                        for v in verts])
        got = heads.visual_correspondence_record(row, n, dr, uv1, uv2, sc.color_hw, tpl)
        assert got == w


def test_visual_correspondence_stale_index(ref, world, tmp_path):
    """A vertex of the common set that fails the visibility re-check (stale index): upstream draws no template for it and
    drops the row; the batched dataset function ends with the same records and the same ``random`` stream."""
    sc, h, rows, vis = world
    tpl = T.TemplateSet.from_module(ref.VC_C, ["default"])
    vis_dict = {sc.scene_id: vis}
    warn = str(tmp_path / "w.txt")
    random.seed(7)
    base = [ref.VC_C.build_training_sample(h, row, n, vis_dict, warn) for n, row in enumerate(rows)]
    target = [n for n, b in enumerate(base) if b is not None][2]
    bad_vertex = None
    plain_get = h.get_point_2d_coordinates_in_image
    state = {"row": -1}

    def flaky(scene_id, image_id, point_id, **kw):
        out = plain_get(scene_id, image_id, point_id, **kw)
        if state["row"] == target and state.setdefault("img", image_id) == image_id:
            state["vertex"] = point_id
            return out[:0]
        return out
    h.get_point_2d_coordinates_in_image = flaky
    random.seed(7)
    want = []
    for n, row in enumerate(rows):
        state["row"] = n
        want.append(ref.VC_C.build_training_sample(h, row, n, vis_dict, warn))
    end_ref = random.getstate()
    h.get_point_2d_coordinates_in_image = plain_get
    assert want[target] is None and "vertex" in state
    bad_vertex = state["vertex"]

    class Flaky(_OracleCorrespondenceBackend):
        def project(self, scene_id, jobs):
            return [(v, a, b, o1 and v != bad_vertex, o2) for (v, a, b, o1, o2) in super().project(scene_id, jobs)]
    warned = []
    random.seed(7)
    got = heads.visual_correspondence_dataset(rows, Flaky(sc, vis), None, tpl, on_warn=warned.append)
    assert got == want and random.getstate() == end_ref
    assert any("is not visible in image" in w for w in warned) and any("No conversation" in w for w in warned)


def test_depth_estimation_records(ref, world, tmp_path):
    sc, h, rows, vis = world
    vis_path = os.path.join(h._mspa_root, "vis.pkl")
    RH.register_pickle(vis_path, {sc.scene_id: vis})
    eng = ref.DE_C.DepthEstimationCoorQAEngine(h._mspa_info_path, visibility_info_path=vis_path,
                                               warning_file=str(tmp_path / "w.txt"))
    eng.scene_info.posed_images_root = h.posed_images_root
    eng.scene_info.instance_data_root = h.instance_data_root
    eng.max_samples = 4
    tpl = T.TemplateSet(list(eng.task_description), {"default": list(eng.templates["questions"])},
                        {"default": list(eng.templates["answers"])})
    random.seed(11)
    want = eng.generate_qa_training_single_scene(sc.scene_id)
    ids = O.valid_image_ids(sc.E)
    n_visible = {k: len(vis["image_to_points"][k]) for k in ids}
    random.seed(11)
    draws = heads.depth_estimation_draws(ids, n_visible, 4, tpl)
    got = []
    for dr in draws:
        for j, pick in zip(dr["positions"], dr["picks"]):
            v = vis["image_to_points"][dr["image_id"]][j]
            uv, d = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[dr["image_id"]], sc.depth[dr["image_id"]],
                                        sc.color_hw)
            got.append(heads.depth_estimation_record(sc.scene_id, dr["image_id"], v, uv[0], float(d[0]), pick,
                                                     sc.color_hw, tpl))
    assert got == want and len(got) == 4


@pytest.mark.parametrize("quantum", [0.0, 0.5])
def test_depth_comparison_records(ref, world, tmp_path, quantum):
    """DC_C.generate_qa_training_single_scene; with ``quantum`` both sides round depths to 0.5 m so that many pairs
    tie and are skipped mid-stream -- the rewind of the speculative draw loop is what is under test."""
    sc, h, rows, vis = world
    vis_path = os.path.join(h._mspa_root, "vis.pkl")
    RH.register_pickle(vis_path, {sc.scene_id: vis})
    eng = ref.DC_C.DepthComparisonCoorQAEngine(h._mspa_info_path, visibility_info_path=vis_path,
                                               warning_file=str(tmp_path / "w.txt"))
    eng.scene_info.posed_images_root = h.posed_images_root
    eng.scene_info.instance_data_root = h.instance_data_root
    eng.max_samples = 60                       # more than the scene has images: drawn with replacement (DC_C:239-241)
    if quantum:
        plain = eng.scene_info.get_point_2d_coordinates_in_image

        def quantised(*a, **k):
            uv, d = plain(*a, **k)
            return uv, np.round(d / quantum) * quantum
        eng.scene_info.get_point_2d_coordinates_in_image = quantised
    tpl = T.TemplateSet(list(eng.task_description),
                        {"closer": list(eng.templates["closer_questions"]), "farther": list(eng.templates["farther_questions"])},
                        {"closer": list(eng.templates["closer_answers"]), "farther": list(eng.templates["farther_answers"])})
    random.seed(13)
    want = eng.generate_qa_training_single_scene(sc.scene_id)
    state_ref = random.getstate()
    ids = O.valid_image_ids(sc.E)
    n_visible = {k: len(vis["image_to_points"][k]) for k in ids}

    def numeric_fn(samples):
        out = []
        for image_id, j in samples:
            v = vis["image_to_points"][image_id][j]
            uv, d = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[image_id], sc.depth[image_id], sc.color_hw)
            d = float(d[0]) if not quantum else float(np.round(d[0] / quantum) * quantum)
            out.append((v, uv[0], d))
        return out
    skipped = []
    random.seed(13)
    got = heads.depth_comparison_records(sc.scene_id, ids, n_visible, numeric_fn, sc.color_hw, 60, tpl,
                                         on_skip=lambda *a: skipped.append(a))
    assert got == want and random.getstate() == state_ref
    assert len(got) + len(skipped) == 60 and (len(skipped) >= 2 if quantum else True)


def _oracle_numeric_fn(sc, vis, quantum=0.0):
    def fn(samples):
        out = []
        for image_id, j in samples:
            v = vis["image_to_points"][image_id][j]
            uv, d = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[image_id], sc.depth[image_id], sc.color_hw)
            d = float(d[0]) if not quantum else float(np.round(d[0] / quantum) * quantum)
            out.append((v, uv[0], d))
        return out
    return fn


def test_depth_estimation_dot_records(ref, world, tmp_path):
    """DE_D: the coordinate head plus a disc colour drawn before the templates; annotated image names."""
    sc, h, rows, vis = world
    vis_path = os.path.join(h._mspa_root, "vis.pkl")
    RH.register_pickle(vis_path, {sc.scene_id: vis})
    eng = ref.DE_D.DepthEstimationDotQAEngine(h._mspa_info_path, visibility_info_path=vis_path, image_output_dir=str(tmp_path / "img"),
                                              warning_file=str(tmp_path / "w.txt"))
    eng.scene_info.posed_images_root = h.posed_images_root
    eng.scene_info.instance_data_root = h.instance_data_root
    eng.max_samples = 5
    tpl = T.TemplateSet(list(eng.task_description), {"default": list(eng.templates["questions"])},
                        {"default": list(eng.templates["answers"])})
    random.seed(19)
    want = eng.generate_qa_training_single_scene(sc.scene_id)
    state_ref = random.getstate()
    ids = O.valid_image_ids(sc.E)
    n_visible = {k: len(vis["image_to_points"][k]) for k in ids}
    marks = []
    random.seed(19)
    got = heads.depth_estimation_records_fn(sc.scene_id, ids, n_visible, _oracle_numeric_fn(sc, vis), sc.color_hw, 5, tpl,
                                            dot=True, on_mark=lambda *a: marks.append(a))
    assert got == want and len(got) == 5 and random.getstate() == state_ref
    assert len(marks) == 5 and all(len(m[4]) == 3 for m in marks)
    assert got[0]["image"][0].endswith("_annotated.jpg") and got[0]["question_type"] == "depth_estimation_dot"


@pytest.mark.parametrize("quantum", [0.0, 0.5])
def test_depth_comparison_dot_records(ref, world, tmp_path, quantum):
    sc, h, rows, vis = world
    vis_path = os.path.join(h._mspa_root, "vis.pkl")
    RH.register_pickle(vis_path, {sc.scene_id: vis})
    eng = ref.DC_D.DepthComparisonDotQAEngine(h._mspa_info_path, visibility_info_path=vis_path, image_output_dir=str(tmp_path / "img"),
                                              warning_file=str(tmp_path / "w.txt"))
    eng.scene_info.posed_images_root = h.posed_images_root
    eng.scene_info.instance_data_root = h.instance_data_root
    eng.max_samples = 50
    if quantum:
        plain = eng.scene_info.get_point_2d_coordinates_in_image

        def quantised(*a, **k):
            uv, d = plain(*a, **k)
            return uv, np.round(d / quantum) * quantum
        eng.scene_info.get_point_2d_coordinates_in_image = quantised
    tpl = T.TemplateSet(list(eng.task_description),
                        {"closer": list(eng.templates["closer_questions"]), "farther": list(eng.templates["farther_questions"])},
                        {"closer": list(eng.templates["closer_answers"]), "farther": list(eng.templates["farther_answers"])})
    random.seed(29)
    want = eng.generate_qa_training_single_scene(sc.scene_id)
    state_ref = random.getstate()
    ids = O.valid_image_ids(sc.E)
    n_visible = {k: len(vis["image_to_points"][k]) for k in ids}
    marks, skipped = [], []
    random.seed(29)
    got = heads.depth_comparison_records(sc.scene_id, ids, n_visible, _oracle_numeric_fn(sc, vis, quantum), sc.color_hw, 50, tpl,
                                         on_skip=lambda *a: skipped.append(a), dot=True, on_mark=lambda *a: marks.append(a))
    assert got == want and random.getstate() == state_ref
    assert len(got) == 50 and len(marks) == 50 and (len(skipped) >= 2 if quantum else True)   # tied pairs are re-drawn (DC_D:263-309)
    assert got[0]["gt_value"] in ("A", "B") and got[0]["question_type"] == "depth_comparison_annotated"


def test_object_movement_records(ref):
    tr = synth.make_tracks(21, T=18, P=30)
    H, W = tr.image_hw
    world = O.tracks_cam_to_world(tr.tracks_XYZ, tr.extrinsics_w2c)
    rng = np.random.default_rng(2)
    pairs = [{"frame1": int(a), "frame2": int(b), "point_index": int(p)}
             for a, b, p in zip(rng.integers(0, 18, 40), rng.integers(0, 18, 40), rng.integers(0, 30, 40))]
    for qt in T.OBJECT_MOVEMENT_TYPES:
        eng = ref.OM_C.TwoFrameVideoQAEngine(qt, "adt")
        tpl = T.TemplateSet.from_module(ref.OM_C)
        random.seed(5)
        want = eng.format_training_samples(pairs, tr.fx_fy_cx_cy, tr.scene_id, world, tr.tracks_XYZ, H, W,
                                           tr.extrinsics_w2c)
        random.seed(5)
        got = []
        for s in pairs:
            o = O.object_displacement(world, tr.tracks_XYZ, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                      s["frame1"], s["frame2"], s["point_index"])
            p1n = O.project_point(tr.tracks_XYZ[s["frame1"], s["point_index"]], tr.fx_fy_cx_cy, H, W)
            p2n = O.project_point(tr.tracks_XYZ[s["frame2"], s["point_index"]], tr.fx_fy_cx_cy, H, W)
            if o is None:
                num = {"p1n": p1n, "p2n": p2n}
            else:
                dist = np.linalg.norm(world[s["frame2"], s["point_index"]] - world[s["frame1"], s["point_index"]])
                num = {"distance": float(dist) if o["point_moving"] else 0, "vector": o["gt_vector"],
                       "point_moving": bool(o["point_moving"]), "cam_moving": bool(o["cam_moving"]),
                       "p1n": p1n, "p2n": p2n}
            r = heads.object_movement_record(tr.scene_id, s["frame1"], s["frame2"], s["point_index"], qt, num,
                                             tr.image_hw, tpl)
            if r is not None:
                got.append(r)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g == w


def test_object_movement_dot_records(ref, tmp_path):
    """OM_D.format_training_samples: the colour of the disc is drawn after the templates and only for annotated frames
    that do not exist yet, so repeated (frame1, point) samples shorten the stream."""
    tr = synth.make_tracks(21, T=18, P=30)
    H, W = tr.image_hw
    world = O.tracks_cam_to_world(tr.tracks_XYZ, tr.extrinsics_w2c)
    rng = np.random.default_rng(2)
    pairs = [{"frame1": int(a), "frame2": int(b), "point_index": int(p)}
             for a, b, p in zip(rng.integers(0, 18, 40), rng.integers(0, 18, 40), rng.integers(0, 30, 40))]
    pairs += [dict(pairs[3]), dict(pairs[7], frame2=1)]                  # repeats of an annotated (frame1, point)
    base = tmp_path / "base" / tr.scene_id
    base.mkdir(parents=True)
    for f in range(18):
        (base / f"{f:05d}.jpg").write_bytes(b"jpeg")
        RH.STORE.images[str(base / f"{f:05d}.jpg")] = np.zeros((H, W, 3), np.uint8)
    import cv2
    real_imwrite = cv2.imwrite
    cv2.imwrite = lambda path, img: open(path, "wb").write(b"x") or True   # the existence test needs real files
    try:
        for qt in T.OBJECT_MOVEMENT_TYPES:
            eng = ref.OM_D.TwoFrameVideoQAEngineDot(qt, "adt")
            eng.image_width, eng.image_height = W, H                      # upstream reads self.image_width without setting it
            tpl = T.TemplateSet.from_module(ref.OM_D)
            out_dir = tmp_path / f"out_{qt}"
            random.seed(5)
            want = eng.format_training_samples(pairs, tr.fx_fy_cx_cy, tr.scene_id, world, tr.tracks_XYZ, H, W,
                                               tr.extrinsics_w2c, str(tmp_path / "base"), str(out_dir))
            state_ref = random.getstate()
            done, marks = set(), []
            random.seed(5)
            got = []
            for s in pairs:
                o = O.object_displacement(world, tr.tracks_XYZ, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                          s["frame1"], s["frame2"], s["point_index"])
                p1n = O.project_point(tr.tracks_XYZ[s["frame1"], s["point_index"]], tr.fx_fy_cx_cy, H, W)
                p2n = O.project_point(tr.tracks_XYZ[s["frame2"], s["point_index"]], tr.fx_fy_cx_cy, H, W)
                if o is None:
                    num = {"p1n": p1n, "p2n": p2n}
                else:
                    dist = np.linalg.norm(world[s["frame2"], s["point_index"]] - world[s["frame1"], s["point_index"]])
                    num = {"distance": float(dist) if o["point_moving"] else 0, "vector": o["gt_vector"],
                           "point_moving": bool(o["point_moving"]), "cam_moving": bool(o["cam_moving"]), "p1n": p1n, "p2n": p2n}

                def needs(name):
                    fresh = name not in done
                    done.add(name)
                    return fresh
                r = heads.object_movement_record(tr.scene_id, s["frame1"], s["frame2"], s["point_index"], qt, num, tr.image_hw, tpl,
                                                 dot=True, needs_annotation=needs, on_mark=lambda *a: marks.append(a))
                if r is not None:
                    got.append(r)
            assert got == want and len(got) > 10 and random.getstate() == state_ref
            assert sum(m[4] is None for m in marks) >= 1 and got[0]["id"].endswith("_ann")
    finally:
        cv2.imwrite = real_imwrite


class _OracleCorrespondenceBackend:
    """The three numerics of heads.GpuCorrespondenceBackend from the oracle (CPU)."""

    def __init__(self, sc, vis):
        self.sc, self.vis = sc, vis

    def image_hw(self, scene_id):
        return self.sc.color_hw if scene_id == self.sc.scene_id else None

    def common_counts(self, scene_id, pairs):
        if scene_id != self.sc.scene_id:
            return None
        i2p = self.vis["image_to_points"]
        return [len(np.intersect1d(i2p.get(a, []), i2p.get(b, []))) for a, b in pairs]

    def project(self, scene_id, jobs):
        sc, i2p, out = self.sc, self.vis["image_to_points"], []
        for a, b, pos in jobs:
            v = int(np.intersect1d(i2p[a], i2p[b])[pos])
            uv1, _ = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[a], sc.depth[a], sc.color_hw)
            uv2, _ = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[b], sc.depth[b], sc.color_hw)
            out.append((v, uv1[0], uv2[0], True, True))
        return out


def test_visual_correspondence_dot_records(ref, world, tmp_path):
    """VC_D.build_training_sample row after row == the batched dataset function (same draws, same records, same stream at
    the end), including a row whose random distractor is forced onto the correct pixel."""
    sc, h, rows, vis = world
    H, W = sc.color_hw
    h.image_width, h.image_height = W, H                    # upstream reads these handler attributes without defining them
    ref.VC_D.USE_PICKLE = True
    tpl = T.TemplateSet.from_module(ref.VC_D, ["default"])
    vis_dict = {sc.scene_id: vis}
    warn = str(tmp_path / "w.txt")
    all_rows = rows + [dict(rows[0], scene_id="scene_unknown_00")] + rows[:5]
    img_dir = str(tmp_path / "images")
    random.seed(31)
    want = [ref.VC_D.build_training_sample(h, row, n, vis_dict, warn, image_output_dir=img_dir) for n, row in enumerate(all_rows)]
    state_ref = random.getstate()
    marks = []
    random.seed(31)
    got = heads.visual_correspondence_dot_dataset(all_rows, _OracleCorrespondenceBackend(sc, vis), tpl,
                                                  on_mark=lambda *a: marks.append(a))
    assert len(got) == len(want) and sum(g is not None for g in got) >= 4
    for g, w in zip(got, want):
        assert g == w
    assert random.getstate() == state_ref and len(marks) == sum(g is not None for g in got)

    # a vertex that fails the visibility re-check (stale index): upstream warns and returns before the colour draws
    target = [n for n, g in enumerate(got) if g is not None][1]
    plain_get = h.get_point_2d_coordinates_in_image
    calls = {"row": -1}

    def flaky(scene_id, image_id, point_id, **kw):
        out = plain_get(scene_id, image_id, point_id, **kw)
        return out[:0] if calls["row"] == target and calls.setdefault("hit", image_id) == image_id else out
    h.get_point_2d_coordinates_in_image = flaky
    random.seed(31)
    want3 = []
    for n, row in enumerate(all_rows):
        calls["row"] = n
        want3.append(ref.VC_D.build_training_sample(h, row, n, vis_dict, warn, image_output_dir=img_dir))
    state3 = random.getstate()
    h.get_point_2d_coordinates_in_image = plain_get
    assert want3[target] is None

    class Flaky(_OracleCorrespondenceBackend):
        def project(self, scene_id, jobs):
            res = super().project(scene_id, jobs)
            return [(v, a, b, o1 and v != bad_vertex, o2) for (v, a, b, o1, o2) in res]
    bad_vertex = int(got[target]["id"].split("_p")[1])
    warned = []
    random.seed(31)
    got3 = heads.visual_correspondence_dot_dataset(all_rows, Flaky(sc, vis), tpl, on_warn=warned.append)
    # (the oracle backend flags that vertex wherever it is drawn; with seed 31 it is drawn for the target row only)
    assert got3 == want3 and random.getstate() == state3 and any("is not visible in image" in w for w in warned)

    # force a clash: a generator whose first distractor of row 2 is the correct pixel of that row
    class Rigged(random.Random):
        """Overrides two randint results at one position of the stream -- and again whenever the generator is back at that
        very position, as it is when the draws are replayed from a checkpoint."""
        armed = None

        def __init__(self, seed):
            super().__init__(seed)
            self.fixed = {}

        def randint(self, a, b):
            key = hash(self.getstate())
            v = super().randint(a, b)
            if key in self.fixed:
                return self.fixed[key]
            if self.armed and a == 0 and b in (W - 10, H - 10):
                self.fixed[key] = self.armed.pop(0)
                return self.fixed[key]
            return v
    # the rigged runs use seed 77: take the correct pixel of the target row from an unrigged run with that seed (the draws
    # in front of the distractors -- swap, vertex, disc colour -- are the same with and without the rig)
    dry = heads.visual_correspondence_dot_dataset(all_rows, _OracleCorrespondenceBackend(sc, vis), tpl, random.Random(77))
    live = [n for n, g in enumerate(dry) if g is not None and g["p2_list"][0][0] <= W - 10 and g["p2_list"][0][1] <= H - 10][1]
    cx, cy = dry[live]["p2_list"][0]
    forced_rows = []
    plain_draws = heads._vc_dot_row_draws

    def spy(n_common, known, image_hw, templates, rng, correct_point=None):
        if correct_point is not None:
            forced_rows.append(correct_point)
        return plain_draws(n_common, known, image_hw, templates, rng, correct_point)
    heads._vc_dot_row_draws = spy
    if True:
        def run(fn):
            rng = Rigged(77)
            state = {"row": -1}
            orig = rng.random

            def coin():                                      # the swap coin opens every row: arm the rig at row `live`
                state["row"] += 1
                if state["row"] == live:
                    rng.armed = [cx, cy]
                return orig()
            rng.random = coin
            return fn(rng), rng.getstate()
        real = ref.VC_D.random
        try:
            def ref_run(rng):
                ref.VC_D.random = rng
                ref.DE_D.random = rng                        # generate_distinct_colors lives in VC_D itself; harmless
                return [ref.VC_D.build_training_sample(h, row, n, vis_dict, warn, image_output_dir=img_dir)
                        for n, row in enumerate(all_rows)]
            want2, s_ref = run(ref_run)
        finally:
            ref.VC_D.random = real
        got2, s_got = run(lambda rng: heads.visual_correspondence_dot_dataset(all_rows, _OracleCorrespondenceBackend(sc, vis), tpl, rng))
        heads._vc_dot_row_draws = plain_draws
        assert got2 == want2 and got2[live]["p2_list"][0] not in got2[live]["p2_list"][1:]
        assert forced_rows and forced_rows[0] == (cx, cy)          # the rejection path really ran


def test_default_templates_have_the_reference_keys(ref):
    tpl = T.TemplateSet.from_module(ref.CME)
    assert set(T.CAMERA_MOVEMENT.questions) == set(tpl.questions) and set(T.CAMERA_MOVEMENT.answers) == set(tpl.answers)
    om = T.TemplateSet.from_module(ref.OM_C)
    assert set(T.OBJECT_MOVEMENT.questions) <= set(om.questions)
    # every default template formats with the reference's answer fields
    av = heads.camera_movement_answer_values([0.1, -0.2, 0.3], 12.0, -3.0)
    for qt, lst in T.CAMERA_MOVEMENT.answers.items():
        for s in lst:
            s.format(**av)


@pytest.mark.parametrize("npoints,npairs,augment,ratio", [(15, 30, True, 0.05), (1, 1, False, 1.0), (4, 1e8, True, 1.0)])
def test_object_movement_pair_mining(ref, tmp_path, npoints, npairs, augment, ratio):
    """Frame-pair mining of OM_C.generate_qa_training_single_scene (groups -> points -> static + binned moving pairs ->
    swap augmentation): same pairs in the same order, same ``random`` stream afterwards."""
    import sys
    tr = synth.make_tracks(33, T=150, P=64, n_groups=4)
    H, W = tr.image_hw
    sys.modules["cv2"].imdecode = lambda arr, flags=None: np.zeros((H, W, 3), np.uint8)
    path = str(tmp_path / f"{tr.scene_id}.npz")
    jpeg = np.array([b"\xff\xd8fake"] * tr.tracks_XYZ.shape[0], dtype=object)
    np.savez(path, images_jpeg_bytes=jpeg, tracks_XYZ=tr.tracks_XYZ, visibility=tr.visibility, fx_fy_cx_cy=tr.fx_fy_cx_cy,
             extrinsics_w2c=tr.extrinsics_w2c, queries_xyt=np.zeros((tr.tracks_XYZ.shape[1], 3)))
    eng = ref.OM_C.TwoFrameVideoQAEngine("tapvid3d_total_distance", "adt")
    captured = {}

    def capture(sample_pairs, **kw):
        captured["pairs"] = [dict(s) for s in sample_pairs]
        captured["world"] = kw["points_pos_world"]
        return []
    eng.format_training_samples = capture
    random.seed(17)
    eng.generate_qa_training_single_scene(path, npoints, npairs, str(tmp_path / "img"), augment, ratio)
    state_ref = random.getstate()
    world = captured["world"]
    groups = ref.OM_C.filter_large_groups(ref.OM_C.rigid_body_segmentation(tr.tracks_XYZ), min_size=5)
    assert len(groups) >= 2

    def distance_fn(points, frames):
        out = []
        for p, fr in zip(points, frames):
            ii, jj = np.triu_indices(len(fr), 1)
            out.append(np.linalg.norm(world[fr[jj], p] - world[fr[ii], p], axis=1))
        return out
    random.seed(17)
    got = heads.object_movement_mine_pairs(tr.visibility, groups, distance_fn, npoints, npairs, augment, ratio)
    want = captured["pairs"]
    assert len(got) == len(want) and len(got) > 0
    assert [(s["point_index"], int(s["frame1"]), int(s["frame2"])) for s in got] == \
           [(s["point_index"], int(s["frame1"]), int(s["frame2"])) for s in want]
    assert random.getstate() == state_ref
    # the oracle's list-and-loop restatement gives the same pairs
    orc = O.mine_frame_pairs(world, tr.visibility, [list(g) for g in ref.OM_C.filter_large_groups(
        ref.OM_C.rigid_body_segmentation(tr.tracks_XYZ), min_size=5)], npoints, npairs, augment, ratio, random.Random(17))
    assert [(s["point_index"], int(s["frame1"]), int(s["frame2"])) for s in orc] == \
           [(s["point_index"], int(s["frame1"]), int(s["frame2"])) for s in want]
