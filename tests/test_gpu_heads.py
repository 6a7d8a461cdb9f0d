"""Numeric stages of the task heads on the MI355X (K4, K5, K6a, K6b) and the heads end to end.

End-to-end check: the records the GPU heads emit equal the records the pure-Python record stage
builds from ORACLE numerics with the same seed -- and that record stage is pinned to the reference in
tests/test_heads_vs_reference.py, so the chain reference -> record stage -> GPU heads is closed.
"""
import random

import numpy as np
import pytest
import torch

from golden_util import same_f64, close_f64
from mspa import engine, heads, synth
from mspa import templates as T
from mspa.scene import SceneOnDevice
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


# 96x128: the fast small case.  640x480: BASELINE.json configs[0]'s shape -- depth_perception QA (and the other heads' numeric
# stages) on 16 frame pairs of 640x480 RGB-D frames, GPU heads vs oracle numerics.
@pytest.fixture(scope="module", params=[(96, 128), (480, 640)], ids=["96x128", "640x480"])
def world(request):
    hw = request.param
    sc = synth.make_scene(3001, n_points=3000, n_frames=7, color_hw=hw, depth_hw=hw,
                          invalid_pose_frac=0.15, with_color=False)
    scene = SceneOnDevice(sc.K, sc.A, sc.E, sc.depth, sc.color_hw, sc.points, DEV)
    table = O.frames_relations_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
    vis = O.visibility_index_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
    rows = [{"scene_id": sc.scene_id, "image_id1": a, "image_id2": b, "overlap": max(float(v["overlap"]), 1.0),
             "distance": float(v["distance"]), "yaw": float(v["yaw"]), "pitch": float(v["pitch"])}
            for (a, b), v in table.items()]
    if hw == (480, 640):
        rows = rows[:16]                            # configs[0]: 16 frame pairs
        assert len(rows) >= 10
    return sc, scene, rows, vis


def test_select_common_point_and_project_samples(world):
    sc, scene, rows, vis = world
    bits = scene._visibility()["bits"]
    rng = np.random.default_rng(0)
    sel, expect = [], []
    ids = scene.ids
    for _ in range(400):
        a, b = rng.integers(0, len(ids), 2)
        if rng.random() < 0.2:
            b = a                                                       # one image's own visible list
        common = np.intersect1d(vis["image_to_points"][ids[a]], vis["image_to_points"][ids[b]])
        j = int(rng.integers(0, len(common) + 3))                       # sometimes past the end
        sel.append([a, b, j])
        expect.append(int(common[j]) if j < len(common) else -1)
    got = engine.select_common_point(bits, torch.tensor(sel, dtype=torch.int32, device=DEV)).cpu().numpy()
    assert np.array_equal(got, np.array(expect))
    assert (got >= 0).sum() > 50
    # K6b on the hits, both images of the selection
    hit = [(v, s[0]) for v, s in zip(got, sel) if v >= 0] + [(v, s[1]) for v, s in zip(got, sel) if v >= 0]
    samples = torch.tensor(hit, dtype=torch.int32, device=DEV)
    uv, d, ok = engine.project_samples(scene.xyz, scene.cam_mats, scene.depth, scene.image_hw, samples)
    uv, d, ok = uv.cpu().numpy(), d.cpu().numpy(), ok.cpu().numpy().astype(bool)
    assert ok.all()                                                     # taken from the visible lists
    for k in range(0, len(hit), 7):
        v, img = hit[k]
        ruv, rd = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[ids[img]], sc.depth[ids[img]], sc.color_hw,
                                      check_visible=False)
        assert (same_f64(uv[k], ruv[0]) or close_f64(uv[k], ruv[0], rtol=1e-12, scale=1e-6))
        assert close_f64(d[k], rd[0], rtol=1e-12, scale=1e-9)
    # a vertex that is NOT visible reports so
    unseen = [int(np.setdiff1d(np.arange(sc.points.shape[0]), vis["image_to_points"][ids[0]])[0]), 0]
    _, _, ok2 = engine.project_samples(scene.xyz, scene.cam_mats, scene.depth, scene.image_hw,
                                       torch.tensor([unseen], dtype=torch.int32, device=DEV))
    assert not bool(ok2[0])


def test_camera_movement_head(world):
    sc, scene, rows, _ = world
    for qt in ("displacement_vector", "total_distance", "yaw_movement"):
        random.seed(3)
        got = heads.camera_movement_records(scene, rows, qt, sc.color_hw)
        random.seed(3)
        want = []
        for n, row in enumerate(rows):
            E1, E2 = sc.A @ sc.E[row["image_id1"]], sc.A @ sc.E[row["image_id2"]]
            want.append(heads.camera_movement_record(row, n, qt, (np.linalg.inv(E1) @ E2)[:3, 3],
                                                     (np.linalg.inv(E2) @ E1)[:3, 3], sc.color_hw))
        assert len(got) == len(want) == len(rows)
        for g, w in zip(got, want):
            gv, wv = g["answer_values"].pop("displacement_vector"), w["answer_values"].pop("displacement_vector")
            if qt == "displacement_vector":
                g.pop("gt_value"), w.pop("gt_value")
            assert g == w
            assert same_f64(gv, wv) or close_f64(gv, wv, rtol=1e-12, scale=1e-9)


def test_visual_correspondence_head(world):
    sc, scene, rows, vis = world
    random.seed(9)
    got = heads.visual_correspondence_records(scene, rows, sc.color_hw)
    n_common = [len(np.intersect1d(vis["image_to_points"][r["image_id1"]], vis["image_to_points"][r["image_id2"]]))
                for r in rows]
    random.seed(9)
    draws = heads.visual_correspondence_draws(rows, n_common, T.VISUAL_CORRESPONDENCE)
    want = []
    for n, (row, dr) in enumerate(zip(rows, draws)):
        if dr is None:
            continue
        first, second = (row["image_id2"], row["image_id1"]) if dr["swap"] else (row["image_id1"], row["image_id2"])
        common = np.intersect1d(vis["image_to_points"][first], vis["image_to_points"][second])
        verts = [int(common[j]) for j in dr["positions"]]
        uv1 = np.stack([O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[first], sc.depth[first], sc.color_hw)[0][0]
                        for v in verts])
        uv2 = np.stack([O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[second], sc.depth[second], sc.color_hw)[0][0]
                        for v in verts])
        want.append(heads.visual_correspondence_record(row, n, dr, uv1, uv2, sc.color_hw))
    assert len(want) > 3 and sorted(got, key=lambda r: r["id"]) == sorted(want, key=lambda r: r["id"])


def test_depth_estimation_head(world):
    sc, scene, rows, vis = world
    random.seed(13)
    got = heads.depth_estimation_records(scene, sc.scene_id, sc.color_hw, max_samples=5)
    ids = O.valid_image_ids(sc.E)
    random.seed(13)
    draws = heads.depth_estimation_draws(ids, {k: len(vis["image_to_points"][k]) for k in ids}, 5, T.DEPTH_ESTIMATION)
    want = []
    for dr in draws:
        for j, pick in zip(dr["positions"], dr["picks"]):
            v = vis["image_to_points"][dr["image_id"]][j]
            uv, d = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[dr["image_id"]], sc.depth[dr["image_id"]],
                                        sc.color_hw)
            want.append(heads.depth_estimation_record(sc.scene_id, dr["image_id"], v, uv[0], float(d[0]), pick,
                                                      sc.color_hw))
    assert got == want and len(got) == 5


def _oracle_numeric_fn(sc, vis):
    def fn(samples):
        out = []
        for image_id, j in samples:
            v = vis["image_to_points"][image_id][j]
            uv, d = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[image_id], sc.depth[image_id], sc.color_hw)
            out.append((v, uv[0], float(d[0])))
        return out
    return fn


def test_depth_comparison_head(world):
    sc, scene, rows, vis = world
    ids = O.valid_image_ids(sc.E)
    n_visible = {k: len(vis["image_to_points"][k]) for k in ids}
    random.seed(21)
    got = heads.depth_comparison_records_gpu(scene, sc.scene_id, sc.color_hw, max_samples=40)
    end_state = random.getstate()
    random.seed(21)
    want = heads.depth_comparison_records(sc.scene_id, ids, n_visible, _oracle_numeric_fn(sc, vis), sc.color_hw, 40)
    assert got == want and len(got) >= 38 and random.getstate() == end_state
    r = got[0]
    assert r["question_type"] == "depth_comparison_coordinate" and len(r["points_info"]) == 2
    assert {p["letter"] for p in r["points_info"]} == {"A", "B"}


def test_depth_facade_engines(world, tmp_path):
    """spatial_engine.depth_perception.* engines: scene handler + visibility index in, JSONL out."""
    import importlib
    import json
    import pickle
    sc, scene, rows, vis = world
    IH = importlib.import_module("spatial_engine.utils.scannet_utils.handler.info_handler")
    IMG = importlib.import_module("spatial_engine.utils.scannet_utils.handler._images")
    DE = importlib.import_module("spatial_engine.depth_perception.depth_estimation_coor_engine")
    DC = importlib.import_module("spatial_engine.depth_perception.depth_comparison_coor_engine")
    posed, inst = str(tmp_path / "posed_images"), str(tmp_path / "inst")
    import os
    os.makedirs(os.path.join(inst, sc.scene_id))
    np.save(os.path.join(inst, sc.scene_id, "aligned_points.npy"), sc.points)
    H, W = sc.color_hw
    for i in sc.image_ids:
        IMG.register(os.path.join(posed, sc.scene_id, f"{i}.jpg"), np.zeros((H, W, 3), np.uint8))
        IMG.register(os.path.join(posed, sc.scene_id, f"{i}.png"), sc.depth[i])
    info_path, vis_path = str(tmp_path / "infos.pkl"), str(tmp_path / "vis.pkl")
    with open(info_path, "wb") as f:
        pickle.dump({sc.scene_id: sc.info_dict()}, f)
    with open(vis_path, "wb") as f:
        pickle.dump({sc.scene_id: vis}, f)
    ids = O.valid_image_ids(sc.E)
    n_visible = {k: len(vis["image_to_points"][k]) for k in ids}
    for mod, cls, name in ((DE, "DepthEstimationCoorQAEngine", "depth_estimation_coor"),
                           (DC, "DepthComparisonCoorQAEngine", "depth_comparison_coor")):
        eng = getattr(mod, cls)(info_path, all_max_samples=12, visibility_info_path=vis_path, warning_file=str(tmp_path / "w.txt"))
        eng.scene_info.posed_images_root, eng.scene_info.instance_data_root = posed, inst
        random.seed(31)
        eng.generate_qa_training_data(str(tmp_path / "train"))
        recs = [json.loads(line) for line in open(tmp_path / "train" / f"{name}.jsonl")]
        # the same thing from oracle numerics: one scene, max_samples = 12 // 1 + 1 = 13 images (with replacement for DC)
        random.seed(31)
        if name == "depth_estimation_coor":
            want = heads.depth_estimation_records_fn(sc.scene_id, ids, n_visible, _oracle_numeric_fn(sc, vis), sc.color_hw, 13)
        else:
            want = heads.depth_comparison_records(sc.scene_id, ids, n_visible, _oracle_numeric_fn(sc, vis), sc.color_hw, 13)
        if len(want) > 12:
            want = random.sample(want, 12)
        random.shuffle(want)
        assert recs == json.loads(json.dumps(want)) and len(recs) >= 5
        random.seed(32)
        eng.generate_qa_eval_data(str(tmp_path / "val"))
        ev = [json.loads(line) for line in open(tmp_path / "val" / f"{name}.jsonl")]
        assert ev and all("text" in r for r in ev)


def test_object_movement_head():
    tr = synth.make_tracks(21, T=18, P=30)
    world = O.tracks_cam_to_world(tr.tracks_XYZ, tr.extrinsics_w2c)
    rng = np.random.default_rng(2)
    pairs = [{"frame1": int(a), "frame2": int(b), "point_index": int(p)}
             for a, b, p in zip(rng.integers(0, 18, 40), rng.integers(0, 18, 40), rng.integers(0, 30, 40))]
    H, W = tr.image_hw
    for qt in T.OBJECT_MOVEMENT_TYPES:
        random.seed(5)
        got = heads.object_movement_records(tr.scene_id, tr.tracks_XYZ, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                            pairs, qt)
        random.seed(5)
        want = []
        for s in pairs:
            o = O.object_displacement(world, tr.tracks_XYZ, tr.extrinsics_w2c, tr.fx_fy_cx_cy, tr.image_hw,
                                      s["frame1"], s["frame2"], s["point_index"])
            if o is None:
                continue
            dist = np.linalg.norm(world[s["frame2"], s["point_index"]] - world[s["frame1"], s["point_index"]])
            num = {"distance": float(dist) if o["point_moving"] else 0, "vector": o["gt_vector"],
                   "point_moving": bool(o["point_moving"]), "cam_moving": bool(o["cam_moving"]),
                   "p1n": O.project_point(tr.tracks_XYZ[s["frame1"], s["point_index"]], tr.fx_fy_cx_cy, H, W),
                   "p2n": O.project_point(tr.tracks_XYZ[s["frame2"], s["point_index"]], tr.fx_fy_cx_cy, H, W)}
            want.append(heads.object_movement_record(tr.scene_id, s["frame1"], s["frame2"], s["point_index"], qt, num,
                                                     tr.image_hw))
        assert len(got) == len(want) > 10
        for g, w in zip(got, want):
            if "vector" in qt:
                gv, wv = g.pop("gt_value"), w.pop("gt_value")
                assert same_f64(gv, wv) or close_f64(gv, wv, rtol=1e-12, scale=1e-9)
            assert g == w


# ------------------------------------------------------------------------------------------
# end-to-end pipeline, and its invariance to sharding (2 ranks on one GPU, gloo for the collation)
# ------------------------------------------------------------------------------------------
def _pipeline_scenes():
    return [synth.make_scene(7100 + k, n_points=4000, n_frames=8 + 2 * k, color_hw=(96, 128), depth_hw=(96, 128),
                             invalid_pose_frac=0.1, with_color=False) for k in range(3)]


def _pipeline_tracks():
    return [synth.make_tracks(400 + k, T=100, P=64, n_groups=3) for k in range(3)]


def _pipeline_worker(rank, world, port, out_dir):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "multi-spatialmllm_amd"), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch as th
    from mspa import pipeline, shard as S
    th.cuda.set_device(0)
    ctx = S.init_distributed(th.device("cuda", 0), backend="gloo")
    pipeline.run(_pipeline_scenes(), out_dir, ctx, th.device("cuda", 0), seed=3, n_camera=24, n_correspondence=24,
                 depth_images_per_scene=3, tracks=_pipeline_tracks())
    ctx.barrier()
    ctx.close()


def test_pipeline_end_to_end_and_sharding_invariance(tmp_path):
    import json
    import socket
    import torch.multiprocessing as mp
    from mspa import pipeline
    single = str(tmp_path / "single")
    counts = pipeline.run(_pipeline_scenes(), single, None, DEV, seed=3, n_camera=24, n_correspondence=24,
                          depth_images_per_scene=3, tracks=_pipeline_tracks())
    base = {"camera_movement_total_distance", "camera_movement_displacement_vector",
            "visual_correspondence_coor_2_coor", "depth_estimation_coor", "depth_comparison_coor",
            "object_movement_tapvid3d_total_distance", "object_movement_tapvid3d_displacement_vector"}
    perception = set(counts) - base
    assert base <= set(counts) and perception and all(n.startswith("object_perception_") for n in perception)
    assert any(n.startswith("object_perception_height_k1") for n in perception)
    assert counts["depth_estimation_coor"] == 9 and counts["camera_movement_total_distance"] > 0
    assert counts["object_movement_tapvid3d_total_distance"] > 0
    for name in counts:
        recs = [json.loads(ln) for ln in open(f"{single}/{name}.jsonl")]
        assert len(recs) == counts[name]
        for r in recs:      # InternVL multi-image chat schema
            assert {"id", "image", "conversations", "height_list", "width_list", "question_type", "gt_value"} <= set(r)
            assert r["conversations"][0]["from"] == "human" and r["conversations"][1]["from"] == "gpt"
            assert len(r["image"]) == len(r["height_list"]) == len(r["width_list"])
    assert {"load_s", "pair_table_s", "heads_s", "exchange_s", "write_s", "records_bytes"} <= set(pipeline.LAST_TIMINGS)
    assert "rank0_replay_s" not in pipeline.LAST_TIMINGS           # rank 0 rebuilds nothing: records arrive as finished text
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    sharded = str(tmp_path / "sharded")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, sharded)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for name in counts:
        assert open(f"{single}/{name}.jsonl").read() == open(f"{sharded}/{name}.jsonl").read(), name


def test_scene_object_coverage_matches_index_route():
    """Resident route (K1 bitsets -> K2 popcount + K8 extents -> search) == the script route that goes through
    the visibility index lists (what the reference's three object_perception scripts read from disk)."""
    import importlib
    import json
    import random
    from mspa.scene import SceneOnDevice
    sc = synth.make_scene(7300, n_points=5000, n_frames=30, color_hw=(96, 128), depth_hw=(96, 128),
                          invalid_pose_frac=0.05, with_color=False, walk_step=0.5, target_jitter=1.5)
    scene = SceneOnDevice(sc.K, sc.A, sc.E, sc.depth, sc.color_hw, sc.points, DEV)
    idx, bbox, cat = sc.objects()
    assert len(idx) >= 4
    cov, vis = scene.object_coverage(idx, bbox, rng=random.Random(5))
    assert cov and any(res["height"] for res in cov.values())
    COV = importlib.import_module("spatial_engine.object_perception.single_object_coverage_finder")
    index = scene.visibility_index()["image_to_points"]
    sid = sc.scene_id
    vis_dict = {f"{sid}:image_to_points:{k}": json.dumps(v) for k, v in index.items()}

    class H:
        def get_scene_points_align(self, s):
            return sc.points

        def get_object_point_index(self, s, o):
            return idx[o]

        def get_object_height(self, s, o):
            return bbox[o][5]

        def get_object_length(self, s, o):
            return max(bbox[o][3], bbox[o][4])

        def get_object_width(self, s, o):
            return min(bbox[o][3], bbox[o][4])

        def get_object_width_axis_aligned(self, s, o):
            return 0 if bbox[o][3] < bbox[o][4] else 1

    state = random.getstate()
    random.seed(5)
    _, res = COV.process_scene_for_coverage(sid, H(), vis_dict, {sid: vis})
    random.setstate(state)
    assert res == cov


def test_dataset_level_builders(tmp_path):
    """CME / VC_C build_train_dataset + build_val_dataset: one batched GPU pass over rows that span scenes == the per-row
    loop of build_training_sample (which is pinned to the reference) under the same seed, shuffle included."""
    import importlib
    import json
    import os
    import pickle
    pd = pytest.importorskip("pandas")
    pytest.importorskip("pyarrow")
    IH = importlib.import_module("spatial_engine.utils.scannet_utils.handler.info_handler")
    IMG = importlib.import_module("spatial_engine.utils.scannet_utils.handler._images")
    CME = importlib.import_module("spatial_engine.camera_movement.camera_movement_engine_train_val")
    VC = importlib.import_module("spatial_engine.visual_correspondence.visual_correspondence_qa_engine_coor_2_coor")
    scenes = [synth.make_scene(7400 + k, n_points=3000, n_frames=9 + k, color_hw=(96, 128), depth_hw=(96, 128),
                               invalid_pose_frac=0.1, with_color=False) for k in range(2)]
    posed, inst = str(tmp_path / "posed_images"), str(tmp_path / "inst")
    infos, vis_all, table_rows = {}, {}, []
    for sc in scenes:
        os.makedirs(os.path.join(inst, sc.scene_id))
        np.save(os.path.join(inst, sc.scene_id, "aligned_points.npy"), sc.points)
        for i in sc.image_ids:
            IMG.register(os.path.join(posed, sc.scene_id, f"{i}.jpg"), np.zeros(sc.color_hw + (3,), np.uint8))
            IMG.register(os.path.join(posed, sc.scene_id, f"{i}.png"), sc.depth[i])
        infos[sc.scene_id] = sc.info_dict()
        resident = SceneOnDevice(sc.K, sc.A, sc.E, sc.depth, sc.color_hw, sc.points, DEV)
        vis_all[sc.scene_id] = resident.visibility_index()
        for (a, b), v in resident.frames_relations().items():
            table_rows.append({"scene_id": sc.scene_id, "image_id1": a, "image_id2": b, "overlap": float(v["overlap"]),
                               "distance": float(v["distance"]), "yaw": float(v["yaw"]), "pitch": float(v["pitch"])})
    vis_all["scene_not_in_handler"] = {"image_to_points": {}, "point_to_images": {}}
    info_path, vis_path, table_path = str(tmp_path / "infos.pkl"), str(tmp_path / "vis.pkl"), str(tmp_path / "pairs.parquet")
    with open(info_path, "wb") as f:
        pickle.dump(infos, f)
    with open(vis_path, "wb") as f:
        pickle.dump(vis_all, f)
    df = pd.DataFrame(table_rows)
    df.to_parquet(table_path)
    h = IH.SceneInfoHandler(info_path, posed_images_root=posed, instance_data_root=inst)
    lo, hi = 1, 60
    assert ((df["overlap"] >= lo) & (df["overlap"] <= hi)).sum() >= 10
    out = str(tmp_path / "out")
    os.makedirs(out)

    # ---- camera movement ----
    random.seed(40); np.random.seed(40)
    CME.build_train_dataset(table_path, out, h, "displacement_vector", 30, lo, hi, 1)
    got = [json.loads(line) for line in open(os.path.join(out, "displacement_vector_train.jsonl"))]
    random.seed(40); np.random.seed(40)
    sampled = CME.sample_dataframe(pd.read_parquet(table_path), all_overlap_samples=30, non_overlap_samples=0, overlap_min=lo,
                                   overlap_max=hi, interval=1)
    want = [CME.build_training_sample(h, sampled.iloc[k], k, "displacement_vector") for k in range(len(sampled))]
    random.shuffle(want)
    assert got == json.loads(json.dumps(want)) and len(got) >= 10 and len({r["image"][0].split("/")[0] for r in got}) == 2
    random.seed(41); np.random.seed(41)
    CME.build_val_dataset(table_path, out, h, "yaw_movement", 12, lo, hi, 1)
    val = [json.loads(line) for line in open(os.path.join(out, "yaw_movement_val.jsonl"))]
    assert val and all("text" in r and "conversations" not in r for r in val)

    # ---- visual correspondence ----
    warn = str(tmp_path / "warn.txt")
    open(warn, "w").close()
    random.seed(42); np.random.seed(42)
    VC.build_train_dataset(table_path, out, h, 30, lo, hi, 1, vis_path, warn)
    got = [json.loads(line) for line in open(os.path.join(out, "train_visual_correspondence_coor_2_coor.jsonl"))]
    random.seed(42); np.random.seed(42)
    sampled = VC.sample_dataframe(pd.read_parquet(table_path), all_overlap_samples=30, non_overlap_samples=0, overlap_min=lo,
                                  overlap_max=hi, interval=1)
    want = [VC.build_training_sample(h, sampled.iloc[k], k, vis_all, warn) for k in range(len(sampled))]
    want = [w for w in want if w]
    random.shuffle(want)
    assert got == json.loads(json.dumps(want)) and len(got) >= 10
    random.seed(43); np.random.seed(43)
    VC.build_val_dataset(table_path, out, h, 8, lo, hi, 1, vis_path, warn)
    val = [json.loads(line) for line in open(os.path.join(out, "val_visual_correspondence_coor_2_coor.jsonl"))]
    assert val and all("text" in r for r in val)


# ------------------------------------------------------------------------------------------
# "dot" variants: records from the GPU numerics == records from oracle numerics; annotation jobs; real JPEGs via Pillow
# ------------------------------------------------------------------------------------------
def _write_scene_files(tmp_path, scenes, with_real_jpeg):
    import importlib
    import os
    import pickle
    IMG = importlib.import_module("spatial_engine.utils.scannet_utils.handler._images")
    posed, inst = str(tmp_path / "posed_images"), str(tmp_path / "inst")
    infos = {}
    for sc in scenes:
        os.makedirs(os.path.join(inst, sc.scene_id), exist_ok=True)
        os.makedirs(os.path.join(posed, sc.scene_id), exist_ok=True)
        np.save(os.path.join(inst, sc.scene_id, "aligned_points.npy"), sc.points)
        for i in sc.image_ids:
            jpg = os.path.join(posed, sc.scene_id, f"{i}.jpg")
            IMG.register(jpg, np.zeros(sc.color_hw + (3,), np.uint8))
            IMG.register(os.path.join(posed, sc.scene_id, f"{i}.png"), sc.depth[i])
            if with_real_jpeg:
                from PIL import Image
                Image.fromarray(np.full(sc.color_hw + (3,), 90, np.uint8)).save(jpg)
        infos[sc.scene_id] = sc.info_dict()
    info_path = str(tmp_path / "infos.pkl")
    with open(info_path, "wb") as f:
        pickle.dump(infos, f)
    return info_path, posed, inst


def test_depth_dot_engines(world, tmp_path):
    import importlib
    import json
    import os
    import pickle
    from mspa.annotate import RecordingAnnotator
    sc, scene, rows, vis = world
    try:
        import PIL  # noqa: F401
        real = True
    except ImportError:
        real = False
    info_path, posed, inst = _write_scene_files(tmp_path, [sc], real)
    vis_path = str(tmp_path / "vis.pkl")
    with open(vis_path, "wb") as f:
        pickle.dump({sc.scene_id: vis}, f)
    DE = importlib.import_module("spatial_engine.depth_perception.depth_estimation_dot_engine")
    DC = importlib.import_module("spatial_engine.depth_perception.depth_comparison_dot_engine")
    ids = O.valid_image_ids(sc.E)
    n_visible = {k: len(vis["image_to_points"][k]) for k in ids}
    for mod, cls, name in ((DE, "DepthEstimationDotQAEngine", "depth_estimation_dot"), (DC, "DepthComparisonDotQAEngine", "depth_comparison_dot")):
        img_dir = str(tmp_path / f"img_{name}")
        eng = getattr(mod, cls)(info_path, all_max_samples=6, image_output_dir=img_dir, visibility_info_path=vis_path,
                                warning_file=str(tmp_path / "w.txt"))
        eng.scene_info.posed_images_root, eng.scene_info.instance_data_root = posed, inst
        rec = RecordingAnnotator()
        eng.annotator = rec
        random.seed(51)
        eng.generate_qa_training_data(str(tmp_path / f"train_{name}"))
        got = [json.loads(line) for line in open(tmp_path / f"train_{name}" / f"{name}.jsonl")]
        random.seed(51)
        if name == "depth_estimation_dot":
            want = heads.depth_estimation_records_fn(sc.scene_id, ids, n_visible, _oracle_numeric_fn(sc, vis), sc.color_hw, 7,
                                                     T.DEPTH_ESTIMATION_DOT, dot=True)
        else:
            want = heads.depth_comparison_records(sc.scene_id, ids, n_visible, _oracle_numeric_fn(sc, vis), sc.color_hw, 7,
                                                  T.DEPTH_COMPARISON_DOT, dot=True)
        if len(want) > 6:
            want = random.sample(want, 6)
        random.shuffle(want)
        assert got == json.loads(json.dumps(want)) and len(got) >= 5
        assert len(rec.jobs) >= len(got) and all(j[0] == "annotate" and j[2].endswith("_annotated.jpg") for j in rec.jobs)
        marks = rec.jobs[0][3]
        assert len(marks) == (1 if name == "depth_estimation_dot" else 2) and marks[0][2] == 10
        if real:                                           # the real thing once: a JPEG with the disc in it
            from PIL import Image
            eng.annotator = None
            random.seed(52)
            recs = eng.generate_qa_training_single_scene(sc.scene_id)
            path = os.path.join(img_dir, recs[0]["image"][0])
            im = np.asarray(Image.open(path))
            assert im.shape[:2] == sc.color_hw and (np.abs(im.astype(int) - 90).max() > 20)


def test_correspondence_dot_facade(tmp_path, monkeypatch):
    import importlib
    import json
    import os
    import pickle
    pd = pytest.importorskip("pandas")
    pytest.importorskip("pyarrow")
    from mspa.annotate import RecordingAnnotator
    IH = importlib.import_module("spatial_engine.utils.scannet_utils.handler.info_handler")
    VD = importlib.import_module("spatial_engine.visual_correspondence.visual_correspondence_qa_engine_dot_2_multichoice")
    scenes = [synth.make_scene(7500 + k, n_points=3000, n_frames=9, color_hw=(96, 128), depth_hw=(96, 128),
                               invalid_pose_frac=0.1, with_color=False) for k in range(2)]
    info_path, posed, inst = _write_scene_files(tmp_path, scenes, False)
    vis_all, table_rows = {}, []
    for sc in scenes:
        resident = SceneOnDevice(sc.K, sc.A, sc.E, sc.depth, sc.color_hw, sc.points, DEV)
        vis_all[sc.scene_id] = resident.visibility_index()
        for (a, b), v in resident.frames_relations().items():
            table_rows.append({"scene_id": sc.scene_id, "image_id1": a, "image_id2": b, "overlap": float(v["overlap"]),
                               "distance": float(v["distance"]), "yaw": float(v["yaw"]), "pitch": float(v["pitch"])})
    vis_path, table_path = str(tmp_path / "vis.pkl"), str(tmp_path / "pairs.parquet")
    with open(vis_path, "wb") as f:
        pickle.dump(vis_all, f)
    pd.DataFrame(table_rows).to_parquet(table_path)
    h = IH.SceneInfoHandler(info_path, posed_images_root=posed, instance_data_root=inst)
    out = str(tmp_path / "out")
    os.makedirs(out)
    warn = str(tmp_path / "w.txt")
    open(warn, "w").close()
    rec = RecordingAnnotator()
    monkeypatch.setattr(VD, "ANNOTATOR", rec)         # restored afterwards: the module outlives this test
    random.seed(61); np.random.seed(61)
    VD.build_train_dataset(table_path, out, h, 24, 1, 60, 1, vis_path, warn)
    got = [json.loads(line) for line in open(os.path.join(out, "train_visual_correspondence_dot_2_multichoice.jsonl"))]
    # the same from oracle numerics
    by_id = {sc.scene_id: sc for sc in scenes}

    class Backend:
        def image_hw(self, s):
            return by_id[s].color_hw

        def common_counts(self, s, pairs):
            i2p = vis_all[s]["image_to_points"]
            return [len(np.intersect1d(i2p.get(a, []), i2p.get(b, []))) for a, b in pairs]

        def project(self, s, jobs):
            sc, i2p, res = by_id[s], vis_all[s]["image_to_points"], []
            for a, b, pos in jobs:
                v = int(np.intersect1d(i2p[a], i2p[b])[pos])
                uv1, _ = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[a], sc.depth[a], sc.color_hw)
                uv2, _ = O.point_2d_in_image(sc.points[v], sc.K, sc.A @ sc.E[b], sc.depth[b], sc.color_hw)
                res.append((v, uv1[0], uv2[0], True, True))
            return res
    random.seed(61); np.random.seed(61)
    sampled = VD.sample_dataframe(pd.read_parquet(table_path), all_overlap_samples=24, non_overlap_samples=0, overlap_min=1,
                                  overlap_max=60, interval=1)
    want = [w for w in heads.visual_correspondence_dot_dataset([sampled.iloc[k] for k in range(len(sampled))], Backend(),
                                                               T.VISUAL_CORRESPONDENCE_DOT) if w]
    random.shuffle(want)
    assert got == json.loads(json.dumps(want)) and len(got) >= 8
    assert len(rec.jobs) == 2 * len(got) and len(rec.jobs[1][3]) == 4 and {m[4] for m in rec.jobs[1][3]} == {"A", "B", "C", "D"}
    r = got[0]
    assert r["gt_value"] in "ABCD" and len(r["p2_list"]) == 4 and r["question_type"] == "visual_correspondence_multiple_choice"
    # single-row entry point keeps the caller's index in the names
    random.seed(62)
    one = VD.build_training_sample(h, sampled.iloc[0], 37, vis_all, warn, image_output_dir=str(tmp_path / "dbg"))
    assert one is None or (one["id"].startswith("37_p") and one["image"][0].split(os.sep)[-1].startswith("37_point"))


def test_object_movement_dot_facade(tmp_path):
    import importlib
    import os
    from mspa.annotate import RecordingAnnotator
    OMD = importlib.import_module("spatial_engine.object_movement.single_object_movement_engine_dot")
    tr = synth.make_tracks(33, T=150, P=64, n_groups=4)
    H, W = tr.image_hw
    sof = b"\xff\xd8\xff\xc0\x00\x11\x08" + H.to_bytes(2, "big") + W.to_bytes(2, "big") + b"\x03\x01\x11\x00\x02\x11\x01\x03\x11\x01\xff\xd9"
    (tmp_path / "src").mkdir()
    path = str(tmp_path / "src" / f"{tr.scene_id}.npz")
    np.savez(path, images_jpeg_bytes=np.array([sof] * tr.tracks_XYZ.shape[0], dtype=object), tracks_XYZ=tr.tracks_XYZ,
             visibility=tr.visibility, fx_fy_cx_cy=tr.fx_fy_cx_cy, extrinsics_w2c=tr.extrinsics_w2c)
    eng = OMD.TwoFrameVideoQAEngineDot("tapvid3d_total_distance", "adt")

    class Touching(RecordingAnnotator):                     # the existence test of later samples needs the files to appear
        def annotate(self, src, dst, marks):
            super().annotate(src, dst, marks)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            open(dst, "wb").close()

        def copy(self, src, dst):
            super().copy(src, dst)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            open(dst, "wb").close()
    eng.annotator = Touching()
    random.seed(71)
    recs = eng.generate_qa_training_single_scene(path, str(tmp_path / "base"), 6, 5, str(tmp_path / "img"), True, 0.5)
    assert len(recs) > 20 and all(r["id"].endswith("_ann") and r["image"][0].endswith("_annotated.jpg") for r in recs)
    assert all(isinstance(r["gt_value"], list) for r in recs)          # upstream's == "total_distance" never matches (OM_D:429)
    n_ann = sum(j[0] == "annotate" for j in eng.annotator.jobs)
    assert n_ann == len({r["image"][0] for r in recs}) and n_ann <= len(recs)
    assert all(j[3][0][2] == W // 100 for j in eng.annotator.jobs if j[0] == "annotate")
    out = str(tmp_path / "val.jsonl")
    random.seed(72)
    eng.generate_qa_eval_data([tr.scene_id], str(tmp_path / "src"), str(tmp_path / "base"), str(tmp_path), out, str(tmp_path / "img2"),
                              1, 1, False, max_samples=5)
    assert os.path.exists(out.replace(".jsonl", "_orig.jsonl")) and 0 < sum(1 for _ in open(out)) <= 5
