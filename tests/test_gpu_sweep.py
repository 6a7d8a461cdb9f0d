"""The drop-in entry points on on-disk inputs, one process and two: ``calculate_frames_relations.run_split``,
``make_visibility_info.run_split``, the object-movement dataset builder and ``mspa.pipeline`` over scenes written in the
reference's layout (scene-info pickle + posed_images/*.png + aligned_points.npy; TAPVid .npz files).

Two ranks share the box's one GPU and collate over gloo (``MSPA_DIST_BACKEND=gloo``; RCCL refuses two ranks on one device) --
sharding, loader threads, native PNG ingest, the per-window exchange and rank 0's ordered writers are the real ones.  The files
of the 2-rank run must be byte for byte those of the 1-rank run, and the 1-rank run must match the oracle."""
import hashlib
import json
import os
import pickle
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multi-spatialmllm_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

N_SCENES = 7
INFO = "data/scannet/scannet_instance_data/scenes_info.pkl"
INFO_DEPTH = "data/scannet/scannet_instance_data/scenes_info_depth.pkl"


def _scenes():
    from mspa import synth
    out = []
    for k in range(N_SCENES):
        hw = (480, 640) if k == 3 else (96, 128)
        sc = synth.make_scene(9300 + k, n_points=3000 + 500 * k, n_frames=5 + (k * 3) % 7, color_hw=hw, depth_hw=hw,
                              invalid_pose_frac=0.25 if k == 1 else 0.0, with_color=False, scene_id=f"scene{9300 + k:04d}_00")
        if k == 5:
            for image_id in sc.valid_image_ids[:2]:          # two frames that see nothing: "no in bound points", a NaN overlap
                sc.depth[image_id] = np.zeros_like(sc.depth[image_id])
        out.append(sc)
    return out


def _fake_jpeg(h, w):
    app0 = b"\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    sof = b"\xff\xc0\x00\x11\x08" + h.to_bytes(2, "big") + w.to_bytes(2, "big") + b"\x03\x01\x11\x00\x02\x11\x01\x03\x11\x01"
    return b"\xff\xd8" + app0 + sof + b"\xff\xd9"


def _tracks():
    from mspa import synth
    return [synth.make_tracks(500 + k, T=60 + 20 * k, P=48, n_groups=3) for k in range(5)]


def _write_inputs(root):
    from mspa import synth
    scenes = _scenes()
    paths = synth.write_scannet_layout(scenes, os.path.join(root, "data", "scannet"), jpeg_for_every_image=True)
    # labelled objects (the furniture boxes) for the object-visibility sweep: instance masks + categories in the info record
    with open(paths["info_path"], "rb") as f:
        infos = pickle.load(f)
    for sc in scenes:
        idx, bbox, cat = sc.objects()
        mask = np.zeros(sc.points.shape[0], dtype=np.int64)
        for o, pts in idx.items():
            mask[pts] = o + 1
        np.save(os.path.join(paths["instance_data_root"], sc.scene_id, "instance_mask.npy"), mask)
        infos[sc.scene_id]["num_objects"] = 8
        for o in range(8):
            infos[sc.scene_id][o] = {"raw_category": "wall" if o == 7 else cat.get(o, f"thing{o}")}
            if o in bbox:
                infos[sc.scene_id][o]["aligned_bbox"] = np.append(np.asarray(bbox[o], dtype=np.float64), 0.0)   # + class id
    with open(paths["info_path"], "wb") as f:
        pickle.dump(infos, f)
    # the depth engines draw visible points per sampled image and, like the reference's random.choices / random.sample, raise
    # on an image that sees nothing: they get the split without the scene that has such frames
    with open(INFO_DEPTH if os.path.isabs(INFO_DEPTH) else os.path.join(root, INFO_DEPTH), "wb") as f:
        pickle.dump({k: v for k, v in infos.items() if k != "scene9305_00"}, f)
    os.makedirs(os.path.join(root, "tapvid"), exist_ok=True)
    for tr in _tracks():
        H, W = tr.image_hw
        np.savez(os.path.join(root, "tapvid", f"{tr.scene_id}.npz"),
                 images_jpeg_bytes=np.array([_fake_jpeg(H, W)] * tr.tracks_XYZ.shape[0], dtype=object),
                 tracks_XYZ=tr.tracks_XYZ, visibility=tr.visibility, fx_fy_cx_cy=tr.fx_fy_cx_cy, extrinsics_w2c=tr.extrinsics_w2c)


def _run_everything(out_dir):
    """What a user's job script does; the entry points find RANK / WORLD_SIZE themselves."""
    import random
    import torch
    for name in [m for m in sys.modules if m == "spatial_engine" or m.startswith("spatial_engine.")]:
        if not (getattr(sys.modules[name], "__file__", None) or "").startswith(PKG):     # never the reference's package
            del sys.modules[name]
    if sys.path[0] != PKG:
        sys.path.insert(0, PKG)
    import spatial_engine.camera_movement.calculate_frames_relations as CFR
    import spatial_engine.utils.scannet_utils.make_visibility_info as MVI
    import spatial_engine.object_movement.single_object_movement_engine_coord as OM
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    from mspa import pipeline, shard, sweep
    os.makedirs(out_dir, exist_ok=True)
    timings = sweep.Timings()
    tables = CFR.run_split(INFO, os.path.join(out_dir, "pairs.parquet"), os.path.join(out_dir, "cfr_warn.txt"), num_workers=4,
                           save_interval=3, timings=timings)
    vis = MVI.run_split(INFO, os.path.join(out_dir, "vis.parquet"), os.path.join(out_dir, "mvi_warn.txt"), num_workers=4)
    MVI.run_split(INFO, os.path.join(out_dir, "vis.pkl"), os.path.join(out_dir, "mvi_warn2.txt"), num_workers=2)
    # object visibility from the index just written: scenes dealt over the ranks, each reading only its scenes' row groups
    import spatial_engine.object_perception.compute_object_visibility as COV
    COV.process_split("val", INFO, os.path.join(out_dir, "vis.parquet"), os.path.join(out_dir, "covis"))
    # the camera-movement dataset builders on the pair table just written: every rank draws, each formats its slice of the text
    import spatial_engine.camera_movement.camera_movement_engine_train_val as CME
    handler0 = SceneInfoHandler(INFO)
    random.seed(3)
    np.random.seed(3)
    CME.build_train_dataset(os.path.join(out_dir, "pairs.parquet"), out_dir, handler0, "displacement_vector", 60, 1, 60, 1)
    CME.build_val_dataset(os.path.join(out_dir, "pairs.parquet"), out_dir, handler0, "yaw_movement", 25, 1, 60, 1)
    # ... and the correspondence builder: scenes dealt over the ranks, sizes summed, draws replicated, records as bytes to rank 0
    import spatial_engine.visual_correspondence.visual_correspondence_qa_engine_coor_2_coor as VC
    random.seed(4)
    np.random.seed(4)
    VC.build_train_dataset(os.path.join(out_dir, "pairs.parquet"), out_dir, handler0, 50, 1, 60, 1, os.path.join(out_dir, "vis.pkl"),
                           os.path.join(out_dir, "vc_warn.txt"))
    VC.build_val_dataset(os.path.join(out_dir, "pairs.parquet"), out_dir, handler0, 20, 1, 60, 1, os.path.join(out_dir, "vis.pkl"),
                         os.path.join(out_dir, "vc_warn_val.txt"))
    # ... and the multiple-choice builder: the same sharding, plus the two annotated JPEGs per record drawn by the owner rank
    import spatial_engine.visual_correspondence.visual_correspondence_qa_engine_dot_2_multichoice as VCD
    VCD.ANNOTATOR = None                                     # the default (Pillow) whatever an earlier test installed
    random.seed(6)
    np.random.seed(6)
    os.makedirs(os.path.join(out_dir, "vcd"), exist_ok=True)
    VCD.build_train_dataset(os.path.join(out_dir, "pairs.parquet"), os.path.join(out_dir, "vcd"), handler0, 30, 1, 60, 1,
                            os.path.join(out_dir, "vis.pkl"), os.path.join(out_dir, "vcd_warn.txt"))
    # the two depth-estimation engines: draws for every scene on every rank (they read the index only), scenes dealt over the
    # ranks for the projections, the wording and -- dot engine -- the annotated JPEGs; two calls in a row without reseeding
    import spatial_engine.depth_perception.depth_estimation_coor_engine as DEC
    import spatial_engine.depth_perception.depth_estimation_dot_engine as DED
    # ... and the two comparison engines, whose draws depend on the numerics (a pair of equal millimetre depths is skipped before
    # its templates are drawn): windows of one scene per rank, every start guessed without a skip, wrong guesses redone
    import spatial_engine.depth_perception.depth_comparison_coor_engine as DCC
    import spatial_engine.depth_perception.depth_comparison_dot_engine as DCD
    random.seed(8)
    for mod, cls, sub in ((DEC, "DepthEstimationCoorQAEngine", "de_coor"), (DED, "DepthEstimationDotQAEngine", "de_dot"),
                          (DCC, "DepthComparisonCoorQAEngine", "dc_coor"), (DCD, "DepthComparisonDotQAEngine", "dc_dot")):
        d = os.path.join(out_dir, sub)
        os.makedirs(d, exist_ok=True)
        e = getattr(mod, cls)(INFO_DEPTH, "v1_0", 40, os.path.join(d, "images"), os.path.join(out_dir, "vis.pkl"),
                              warning_file=os.path.join(d, "warn.txt"))
        e.generate_qa_training_data(os.path.join(d, "train"))
        e.all_max_samples = 9
        e.generate_qa_eval_data(os.path.join(d, "val"))
    eng = OM.TwoFrameVideoQAEngine("tapvid3d_total_distance", "adt")
    random.seed(11)
    eng.generate_qa_training_data([t.scene_id for t in _tracks()], "tapvid", out_dir, os.path.join(out_dir, "om_train.jsonl"),
                                  os.path.join(out_dir, f"img_rank{os.environ.get('RANK', '0')}"), 4, 3, True, 0.5, num_workers=3)
    # a second and a third call WITHOUT reseeding (upstream's main makes eight in a row): the writer's sample / shuffle of the
    # first call moved rank 0's generator only -- every rank must start the next call's scenes from the same state
    eng.generate_qa_eval_data([t.scene_id for t in _tracks()], "tapvid", out_dir, os.path.join(out_dir, "om_val.jsonl"),
                              os.path.join(out_dir, f"img_rank{os.environ.get('RANK', '0')}"), 2, 2, True, 0.5, max_samples=7,
                              num_workers=3)
    eng2 = OM.TwoFrameVideoQAEngine("tapvid3d_displacement_vector", "adt")
    eng2.generate_qa_training_data([t.scene_id for t in _tracks()], "tapvid", out_dir, os.path.join(out_dir, "om_train2.jsonl"),
                                   os.path.join(out_dir, f"img_rank{os.environ.get('RANK', '0')}"), 3, 2, True, 0.5, max_samples=40,
                                   num_workers=3)
    ctx = shard.context_from_env()
    handler = SceneInfoHandler(INFO)
    # (not the scene with frames that see nothing: the depth heads draw two visible points per sampled image and, like the
    # reference's random.sample(visible_points, 2), DC_C:259, raise on an image that has none)
    scenes = [pipeline.DiskScene(handler, sid, 3) for sid in handler.get_all_scene_ids() if sid != "scene9305_00"]
    dev = ctx.device if ctx is not None else torch.device("cuda", 0)
    counts = pipeline.run(scenes, os.path.join(out_dir, "pipe"), ctx, dev, seed=5, n_camera=20, n_correspondence=20,
                          depth_images_per_scene=2, tracks=_tracks()[:3])
    pipe_timings = dict(pipeline.LAST_TIMINGS)
    if ctx is None:
        # the same job with rank 0's record collector forced through run files and the external merge, and the track blocks as
        # files priced by size and read by their owner: the same bytes
        track_files = [pipeline.TrackFile(os.path.join("tapvid", f"{t.scene_id}.npz")) for t in _tracks()[:3]]
        spilled = pipeline.run(scenes, os.path.join(out_dir, "pipe_spilled"), ctx, dev, seed=5, n_camera=20, n_correspondence=20,
                               depth_images_per_scene=2, tracks=track_files, spill_bytes=1500)
        assert spilled == counts and pipeline.LAST_TIMINGS["spill_runs"] > 5
        for name in counts:
            assert open(os.path.join(out_dir, "pipe", f"{name}.jsonl"), "rb").read() == \
                open(os.path.join(out_dir, "pipe_spilled", f"{name}.jsonl"), "rb").read(), name
        import shutil
        shutil.rmtree(os.path.join(out_dir, "pipe_spilled"))
    return tables, vis, timings, counts, pipe_timings


def _rank_main(rank, world, port, root, out_dir):
    os.chdir(root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MSPA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mspa import shard
    tables, vis, _, counts, _ = _run_everything(out_dir)
    ctx = shard.context_from_env()
    assert ctx is not None and ctx.world == world and ctx.backend == "gloo" and ctx.device.type == "cuda"
    if rank:
        assert tables == {} and vis == {} and counts == {}
    ctx.barrier()
    ctx.close()


def _digest_tree(out_dir):
    out = {}
    for base, _dirs, files in os.walk(out_dir):
        if "img_rank" in base:
            continue
        for n in files:
            path = os.path.join(base, n)
            out[os.path.relpath(path, out_dir)] = hashlib.sha256(open(path, "rb").read()).hexdigest()
    return out


def test_drop_in_entry_points_one_rank_vs_two_ranks_from_disk(tmp_path, monkeypatch):
    import pandas as pd
    from oracle import np_oracle as O
    root = str(tmp_path)
    _write_inputs(root)
    monkeypatch.chdir(root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    for stub in ("mmengine", "cv2"):                         # oracle/ref_harness.py's stand-ins, if an earlier test imported the reference
        if getattr(sys.modules.get(stub), "__file__", None) is None:
            monkeypatch.delitem(sys.modules, stub, raising=False)
    tables, vis, timings, counts, pipe_t = _run_everything(os.path.join(root, "one"))
    scenes = _scenes()
    # ---- one process vs the oracle -------------------------------------------------------------------------------------
    assert list(tables) == [s.scene_id for s in scenes] == list(vis)
    df = pd.read_parquet(os.path.join(root, "one", "pairs.parquet"))
    want = []
    for sc in scenes:
        for (a, b), v in O.frames_relations_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw).items():
            want.append((sc.scene_id, a, b, v["overlap"], v["distance"], v["yaw"], v["pitch"]))
    assert list(zip(df.scene_id, df.image_id1, df.image_id2)) == [r[:3] for r in want]
    got = df[["overlap", "distance", "yaw", "pitch"]].to_numpy()
    ref = np.array([r[3:] for r in want], dtype=np.float64)
    assert np.array_equal(got[:, 0], ref[:, 0], equal_nan=True)                              # overlaps: bit-exact (integer counts)
    assert np.allclose(got[:, 1:], ref[:, 1:], rtol=1e-12, atol=1e-12)                       # float64 pose quantities
    assert np.isnan(got[:, 0]).sum() == 1
    with open(os.path.join(root, "one", "vis.pkl"), "rb") as f:
        vis_pkl = pickle.load(f)
    vp = pd.read_parquet(os.path.join(root, "one", "vis.parquet"))
    lookup = dict(zip(vp["key"], vp["values"]))
    for sc in scenes:
        w = O.visibility_index_scene(sc.points[:, :3], sc.K, sc.A, sc.E, sc.depth, sc.color_hw)
        assert vis[sc.scene_id] == w == vis_pkl[sc.scene_id]                                  # integer index lists: exact
        for image_id, pts in w["image_to_points"].items():
            assert lookup[f"{sc.scene_id}:image_to_points:{image_id}"] == json.dumps(pts)
    empty = scenes[5].valid_image_ids[0]
    warn = open(os.path.join(root, "one", "cfr_warn.txt")).read()
    assert f"{scenes[5].scene_id}: {empty} has no in bound points\n" in warn and "has something wrong" in warn
    assert timings.n["decode"] == N_SCENES == timings.n["stage"] and timings.s["write"] > 0
    assert counts["camera_movement_total_distance"] > 0 and 0 < counts["depth_estimation_coor"] <= 2 * (N_SCENES - 1)
    assert counts["object_movement_tapvid3d_total_distance"] > 0
    assert pipe_t["records_bytes"] > 1000 and "rank0_replay_s" not in pipe_t
    assert pipe_t["max_resident_scenes"] <= 10 and pipe_t["peak_rss_mb"] > 0 and pipe_t["pair_table_rows"] > 50
    cm = [json.loads(line) for line in open(os.path.join(root, "one", "displacement_vector_train.jsonl"))]
    assert len(cm) > 10 and {"id", "image", "conversations", "answer_values", "gt_value"} <= set(cm[0])
    cmv = [json.loads(line) for line in open(os.path.join(root, "one", "yaw_movement_val.jsonl"))]
    assert len(cmv) > 5 and "text" in cmv[0] and "conversations" not in cmv[0]
    vc = [json.loads(line) for line in open(os.path.join(root, "one", "train_visual_correspondence_coor_2_coor.jsonl"))]
    assert len(vc) > 10 and {"id", "image", "conversations", "p1_list", "p2_list", "gt_value"} <= set(vc[0])
    with open(os.path.join(root, "one", "covis", "object_visibility.pkl"), "rb") as f:
        covis = pickle.load(f)
    assert list(covis) == [s.scene_id for s in scenes]
    sc0 = scenes[0]
    idx0, _, _ = sc0.objects()
    masks0 = O.scene_visibility_masks(sc0.points[:, :3], sc0.K, sc0.A, sc0.E, sc0.depth, sc0.color_hw)
    for o, entries in covis[sc0.scene_id]["object_to_images"].items():                      # counts = |object & image| of the oracle's masks
        for e in entries:
            assert e["intersection_count"] == int(masks0[e["image_id"]][idx0[o]].sum()) >= max(1, int(0.05 * len(idx0[o])))
    assert all(7 not in covis[s]["object_to_images"] for s in covis)                        # the "wall" is not asked about
    assert len(covis[sc0.scene_id]["object_to_images"]) >= 1 and sum(len(v["object_to_images"]) for v in covis.values()) >= 6
    for sub, name in (("de_coor", "depth_estimation_coor"), ("de_dot", "depth_estimation_dot"), ("dc_coor", "depth_comparison_coor"),
                      ("dc_dot", "depth_comparison_dot")):
        tr = [json.loads(line) for line in open(os.path.join(root, "one", sub, "train", f"{name}.jsonl"))]
        ev = [json.loads(line) for line in open(os.path.join(root, "one", sub, "val", f"{name}.jsonl"))]
        assert 20 <= len(tr) <= 40 and 1 <= len(ev) <= 9 and "text" in ev[0] and {"id", "image", "conversations", "gt_value"} <= set(tr[0])
    assert len(os.listdir(os.path.join(root, "one", "de_dot", "images"))) >= 3                # annotated frames, per scene
    vcd = [json.loads(line) for line in open(os.path.join(root, "one", "vcd", "train_visual_correspondence_dot_2_multichoice.jsonl"))]
    assert len(vcd) > 8 and all(os.path.exists(os.path.join(root, "one", "vcd", "images", p)) for r in vcd for p in r["image"])
    om = [json.loads(line) for line in open(os.path.join(root, "one", "om_train.jsonl"))]
    assert len(om) > 5 and {"id", "conversations", "point_moving", "cam_moving"} <= set(om[0])
    # ---- two ranks, one GPU: the same bytes everywhere ----------------------------------------------------------------
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    procs = [mpc.Process(target=_rank_main, args=(r, 2, port, root, os.path.join(root, "two"))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    one, two = _digest_tree(os.path.join(root, "one")), _digest_tree(os.path.join(root, "two"))
    assert set(one) == set(two) and len(one) >= 15
    assert [n for n in one if one[n] != two[n]] == []


def test_handler_host_scene_is_what_the_general_reader_gives(tmp_path, monkeypatch):
    """The native PNG path of ``SceneInfoHandler.host_scene`` against frame-by-frame ``get_depth_image`` (Pillow / OpenCV), and
    the prefetched scene against a directly uploaded one."""
    import torch
    from test_gpu_facade import facade
    SceneInfoHandler = facade().IH.SceneInfoHandler
    root = str(tmp_path)
    _write_inputs(root)
    monkeypatch.chdir(root)
    for stub in ("mmengine", "cv2"):
        if getattr(sys.modules.get(stub), "__file__", None) is None:
            monkeypatch.delitem(sys.modules, stub, raising=False)
    h = SceneInfoHandler(INFO)
    sid = h.get_all_scene_ids()[3]                            # the 640 x 480 scene
    hs = h.host_scene(sid, num_workers=6)
    for image_id in h.get_all_extrinsic_valid_image_ids(sid):
        assert np.array_equal(hs.depth[image_id], h.get_depth_image(sid, image_id))
    direct = h.scene_on_device(sid)
    (pre,) = list(h.prefetched_scenes([sid], num_workers=3))
    torch.cuda.synchronize()
    assert torch.equal(pre.depth, direct.depth) and torch.equal(pre.xyz, direct.xyz) and torch.equal(pre.cam_mats, direct.cam_mats)
    assert pre.frames_relations_arrays()["overlap"].tolist() == direct.frames_relations_arrays()["overlap"].tolist()


def test_device_decode_of_the_sweeps_equals_the_host_decode(tmp_path, monkeypatch):
    """The streaming sweeps decode the depth PNGs on the MI355X (compressed bytes over PCIe, csrc/device_ingest.hip); with
    MSPA_DEPTH_DECODE=host the native host threads do.  Same resident tensors, same parquet bytes -- also for a scene that holds
    a frame the device has to hand to the host reader (an 8-bit PNG) -- and ten scenes in flight at once."""
    import torch
    from PIL import Image
    from test_gpu_facade import facade
    from mspa import sweep
    SceneInfoHandler = facade().IH.SceneInfoHandler
    import spatial_engine.camera_movement.calculate_frames_relations as CFR
    root = str(tmp_path)
    _write_inputs(root)
    monkeypatch.chdir(root)
    for stub in ("mmengine", "cv2"):
        if getattr(sys.modules.get(stub), "__file__", None) is None:
            monkeypatch.delitem(sys.modules, stub, raising=False)
    h = SceneInfoHandler(INFO)
    sids = h.get_all_scene_ids()
    odd = h.get_depth_image_path(sids[2], h.get_all_extrinsic_valid_image_ids(sids[2])[1])
    a = np.array(Image.open(odd))
    Image.fromarray((a >> 4).astype(np.uint8)).save(odd)                   # an 8-bit frame in the middle of a scene
    resident = {}
    for mode in ("device", "host"):
        got = []
        for sc in h.prefetched_scenes(sids * 2, num_workers=3, decode=mode):   # 14 scenes: more than the decode slots
            got.append((sc.depth.clone(), sc.cam_mats.clone()))
            # K4's inverse-pose table is cut out of the resident camera records on the CONSUMER's stream (built on the uploader's
            # thread it raced the copy and held the slot's previous scene: round 6)
            assert torch.equal(sc.pose_tables()[1], sc.cam_mats[:, 0, :]) and sc.pose_tables()[0].shape[0] == len(sc.ids)
        torch.cuda.synchronize()
        resident[mode] = got
    assert len(resident["device"]) == 14
    for (d1, c1), (d2, c2) in zip(resident["device"], resident["host"]):
        assert torch.equal(d1, d2) and torch.equal(c1, c2)
    assert int(resident["device"][2][0][1].max()) < 4096                   # the 8-bit frame came through the host reader
    digests = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("MSPA_DEPTH_DECODE", mode)
        tm = sweep.Timings()
        CFR.run_split(INFO, os.path.join(root, mode, "pairs.parquet"), os.path.join(root, mode, "warn.txt"), num_workers=3, keep=False,
                      timings=tm)
        digests[mode] = _digest_tree(os.path.join(root, mode))
    assert digests["device"] == digests["host"] and len(digests["device"]) == 3


def test_long_lived_streams_are_the_process_own_and_prepared_scenes_are_the_same_scenes(tmp_path, monkeypatch):
    """(1) Decode slots, the sweeps' copy stream and the encoder threads' side streams are streams of the process's own
    (``_lib.own_stream``), one per role however often they are asked for -- ``torch.cuda.Stream()`` walks a pool of 32 and, from a
    process's third sweep on, handed an encoder thread a decode slot's stream.  (2) A scene staged from tables a loader thread
    prepared (``host_scene(prepare=True)``) is bit for bit the scene staged without them."""
    import threading
    import torch
    from test_gpu_facade import facade
    from mspa import _lib, sweep, upload
    SceneInfoHandler = facade().IH.SceneInfoHandler
    a, b = _lib.own_stream("test-role", "cuda"), _lib.own_stream("test-role", "cuda:0")
    assert a is b and a.cuda_stream != _lib.own_stream("test-role-2", "cuda").cuda_stream
    pooled = {torch.cuda.Stream().cuda_stream for _ in range(40)}            # the whole pool, more than once around
    seen = {}

    def ask(name):
        seen[name] = sweep.side_stream("cuda").cuda_stream
    for rep in range(3):                                                      # three "sweeps": same names, same streams
        ts = [threading.Thread(target=ask, args=(f"mspa-encode_{k}",), name=f"mspa-encode_{k}") for k in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        if rep == 0:
            first = dict(seen)
        assert seen == first
    assert len(set(first.values())) == sweep.SIDE_STREAMS and first["mspa-encode_1"] == first[f"mspa-encode_{1 + sweep.SIDE_STREAMS}"]
    own = set(first.values()) | {a.cuda_stream, _lib.own_stream("sweep-copy", "cuda").cuda_stream}
    assert not (own & pooled)
    root = str(tmp_path)
    _write_inputs(root)
    monkeypatch.chdir(root)
    for stub in ("mmengine", "cv2"):
        if getattr(sys.modules.get(stub), "__file__", None) is None:
            monkeypatch.delitem(sys.modules, stub, raising=False)
    h = SceneInfoHandler(INFO)
    sids = h.get_all_scene_ids()
    got = {}
    for prepare in (False, True):
        hosts = [h.host_scene(sid, 3, True, "device", prepare=prepare) for sid in sids]
        assert all((hs.prepared is not None) == prepare for hs in hosts)
        out = []
        for sc in upload.ScenePrefetcher(hosts, "cuda", decode_on_device=True):
            out.append([t.clone() for t in (sc.depth, sc.cam_mats, sc.frame_mats, sc.xyz, sc.pose_tables()[0], sc.pose_tables()[2])]
                       + [sc.frames_relations_arrays()["overlap"].copy()])
        torch.cuda.synchronize()
        got[prepare] = out
        slot_streams = {sl.stream.cuda_stream for sl in upload._SLOT_POOL.get("cuda", []) + upload._SLOT_POOL.get("cuda:0", [])
                        if sl.stream is not None}
        assert slot_streams and not (slot_streams & pooled)
    assert len(got[True]) == len(sids)
    for x, y in zip(got[False], got[True]):
        assert all(torch.equal(p, q) for p, q in zip(x[:-1], y[:-1])) and np.array_equal(x[-1], y[-1], equal_nan=True)


def test_reference_cli_unchanged_under_the_launcher(tmp_path):
    """`python -m spatial_engine.camera_movement.calculate_frames_relations` and `python -m mspa.pipeline --scene-info ...`,
    started the way a user starts a multi-GPU job (`torch.distributed.run`, one process per GPU; here two processes on the one
    GPU, gloo), against the same commands started plain: the scripts' own `main()`s with their own fixed paths, nothing
    handed to them but the environment the launcher sets."""
    import shutil
    import socket
    import subprocess
    from mspa import synth
    root = str(tmp_path)
    scenes = _scenes()
    for run in ("one", "two"):
        base = os.path.join(root, run, "data", "scannet")
        synth.write_scannet_layout(scenes[:4], base, info_name="scenes_train_info_i_D5.pkl", jpeg_for_every_image=True)
        synth.write_scannet_layout(scenes[4:], base, info_name="scenes_val_info_i_D5.pkl", jpeg_for_every_image=True)
        os.makedirs(os.path.join(root, run, "tapvid"), exist_ok=True)
        for tr in _tracks()[:2]:                              # TAPVid-3D sample files for the pipeline's --tapvid-root
            H, W = tr.image_hw
            np.savez(os.path.join(root, run, "tapvid", f"{tr.scene_id}.npz"),
                     images_jpeg_bytes=np.array([_fake_jpeg(H, W)] * tr.tracks_XYZ.shape[0], dtype=object), tracks_XYZ=tr.tracks_XYZ,
                     visibility=tr.visibility, fx_fy_cx_cy=tr.fx_fy_cx_cy, extrinsics_w2c=tr.extrinsics_w2c)
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MSPA_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    plain = [sys.executable]
    launched = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    for run, head in (("one", plain), ("two", launched)):
        cwd = os.path.join(root, run)
        for mod, extra in (("spatial_engine.camera_movement.calculate_frames_relations", []),
                           ("mspa.pipeline", ["--scene-info", "data/scannet/scannet_instance_data/scenes_train_info_i_D5.pkl",
                                              "--tapvid-root", "tapvid", "--out", "pipe_out"])):
            out = subprocess.run(head + ["-m", mod] + extra, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
            assert out.returncode == 0, (run, mod, out.stdout[-1500:], out.stderr[-3000:])
    files = ["training_data/camera_movement/train_camera_info_D5.parquet", "training_data/camera_movement/train_camera_info_D5_nonzero.parquet",
             "evaluation_data/camera_movement/val_camera_info_D5.parquet", "evaluation_data/camera_movement/val_warning_D5.txt"]
    files += [os.path.join("pipe_out", n) for n in sorted(os.listdir(os.path.join(root, "one", "pipe_out")))]
    assert len(files) >= 10
    for n in files:
        a, b = os.path.join(root, "one", n), os.path.join(root, "two", n)
        assert os.path.exists(a) == os.path.exists(b), n
        if os.path.exists(a):
            assert open(a, "rb").read() == open(b, "rb").read(), n
    shutil.rmtree(root, ignore_errors=True)
