"""Both oracles (NumPy restatement, C restatement) against the frozen reference outputs.

Runs anywhere (no /root/reference, no GPU).  Integer / boolean products must match exactly; the
float64 products are required to be bit-identical as well when the local BLAS evaluates 4xK
products in the same FMA order as the machine that produced the goldens (true for every OpenBLAS
FMA kernel we have seen) and, failing that, to agree to 1e-12 relative -- far inside the 1e-5 bar.
"""
import json

import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import np_oracle as O
from golden_util import GoldenScene, close_f64, same_f64

SCENES = ["scene_ident", "scene_scaled"]


def f64_ok(a, b):
    return same_f64(a, b) or close_f64(a, b, rtol=1e-12, scale=1e-3)


@pytest.mark.parametrize("name", SCENES)
def test_vertex_path(name):
    g = GoldenScene(name)
    assert O.valid_image_ids(g.E) == g.valid_image_ids
    pts = g.points[:, :3]
    for n, image_id in enumerate(g.valid_image_ids):
        Ea = O.aligned_extrinsic(g.A, g.E[image_id])
        m, uv, d = O.vertex_visibility(pts, g.K, Ea, g.depth[image_id], g.color_hw)
        assert np.array_equal(m, g["ref_vis"][n])
        assert f64_ok(uv, g["ref_uv"][n]) and f64_ok(d, g["ref_depth"][n])
        mc, uvc, dc = C.vertex_visibility(pts, g.K, Ea, g.depth[image_id], g.color_hw)
        assert np.array_equal(mc, g["ref_vis"][n])
        assert f64_ok(uvc, g["ref_uv"][n]) and f64_ok(dc, g["ref_depth"][n])
        uvp, dp = C.project_points(pts, g.K, Ea)
        assert same_f64(uvp, uvc) and same_f64(dp, dc)


@pytest.mark.parametrize("name", SCENES)
def test_back_projection(name):
    g = GoldenScene(name)
    fid = str(g["a7_frame"])
    color = g.color.get(fid)
    o = O.project_mask_to_3d(g.depth[fid], g.K, g.E[fid], g["a7_mask"], g.A, color)
    assert o.shape == g["ref_a7"].shape and f64_ok(o, g["ref_a7"])
    o = O.project_mask_to_3d(g.depth[fid], g.K, g.E[fid], g["a7_mask"])
    assert f64_ok(o, g["ref_a7_noalign"])


@pytest.mark.parametrize("name", SCENES)
def test_frame_pairs(name):
    g = GoldenScene(name)
    H, W = g.color_hw
    for n, (id1, id2) in enumerate(g["pair_ids"]):
        id1, id2 = str(id1), str(id2)
        col = g.color.get(id1)
        if col is None:
            col = np.zeros((H, W, 3), np.uint8)
        o = O.frame_pair(g.depth[id1], g.depth[id2], g.K, g.E[id1], g.E[id2], g.A, g.color_hw, col)
        c = C.frame_pair(g.depth[id1], g.depth[id2], g.K, g.E[id1], g.E[id2], g.A, g.color_hw)
        ref_xyzrgb, ref_uv, ref_d, ref_vis = (g[f"pair{n}_{k}"] for k in ("xyzrgb", "uv", "depth", "vis"))
        for r in (o, c):
            v = r["valid"]
            assert int(v.sum()) == len(ref_vis) == r["n_valid"]
            assert np.array_equal(r["vis"][v], ref_vis) and not r["vis"][~v].any()
            assert r["n_vis"] == int(ref_vis.sum())
            assert f64_ok(r["xyz"][v], ref_xyzrgb[:, :3])
            assert f64_ok(r["uv2"][v], ref_uv) and f64_ok(r["depth2"][v], ref_d)
        assert np.array_equal(o["rgb"][o["valid"]].astype(np.float64), ref_xyzrgb[:, 3:6])
        assert np.array_equal(o["xi"], c["xi"]) and np.array_equal(o["yi"], c["yi"])
        # identity pair: every valid pixel lands on itself
        if id1 == id2:
            v = o["valid"]
            my, mx = np.divmod(np.arange(H * W), W)
            sx, sy = g.depth_hw[1] / W, g.depth_hw[0] / H
            assert np.array_equal(o["xi"][v], np.clip(np.round(mx[v] * sx).astype(int), 0, g.depth_hw[1] - 1))
            assert np.array_equal(o["yi"][v], np.clip(np.round(my[v] * sy).astype(int), 0, g.depth_hw[0] - 1))


@pytest.mark.parametrize("name", SCENES)
def test_scene_products(name):
    g = GoldenScene(name)
    pts = g.points[:, :3]
    table = O.frames_relations_scene(pts, g.K, g.A, g.E, g.depth, g.color_hw)
    keys = [tuple(str(s) for s in k) for k in g["cfr_pairs"]]
    assert list(table.keys()) == keys
    got = np.array([[table[k][f] for f in ("overlap", "distance", "yaw", "pitch")] for k in keys])
    assert same_f64(got[:, 0], g["cfr_values"][:, 0])            # integer counts -> exact ratio
    assert f64_ok(got[:, 1:], g["cfr_values"][:, 1:])
    vis = O.visibility_index_scene(pts, g.K, g.A, g.E, g.depth, g.color_hw)
    ref = g.json("mvi_json")
    assert vis["image_to_points"] == ref["image_to_points"]
    assert {str(k): v for k, v in vis["point_to_images"].items()} == ref["point_to_images"]
    masks = O.scene_visibility_masks(pts, g.K, g.A, g.E, g.depth, g.color_hw)
    for (a, b), row in zip(keys, g["cfr_values"]):
        ov, inter, uni = C.pair_overlap(masks[a], masks[b])
        assert same_f64(ov, row[0])


@pytest.mark.parametrize("name", SCENES)
def test_relative_pose(name):
    g = GoldenScene(name)
    keys = [tuple(str(s) for s in k) for k in g["cfr_pairs"]]
    for (id1, id2), (yaw, pitch), swap, ans in zip(keys, g["cme_yaw_pitch"], g["cme_swap"], g["cme_answers_json"]):
        ref = json.loads(str(ans))
        E1, E2 = O.aligned_extrinsic(g.A, g.E[id1]), O.aligned_extrinsic(g.A, g.E[id2])
        got = O.relative_pose_answer_values(E1, E2, yaw, pitch, bool(swap))
        dv_ref, dv_got = ref.pop("displacement_vector"), got.pop("displacement_vector")
        assert got == ref
        assert f64_ok(dv_got, dv_ref)
        rel_c = C.matmul4(np.linalg.inv(E2 if swap else E1), E1 if swap else E2)
        assert f64_ok(rel_c[:3, 3], dv_ref)


def test_ties():
    g = GoldenScene("ties")
    pts = g.points[:, :3]
    Ea = O.aligned_extrinsic(g.A, g.E["00000"])
    with np.errstate(all="ignore"):
        m, uv, d = O.vertex_visibility(pts, g.K, Ea, g.depth["00000"], g.color_hw)
    mc, uvc, dc = C.vertex_visibility(pts, g.K, Ea, g.depth["00000"], g.color_hw)
    for mm, uu, dd in ((m, uv, d), (mc, uvc, dc)):
        assert np.array_equal(mm, g["ref_vis"])
        assert same_f64(uu, g["ref_uv"]) and same_f64(dd, g["ref_depth"])     # exact arithmetic: no slack
    assert np.array_equal(O.check_point_in_image_boundary(uv, g.color_hw), g["ref_inb"])
    # the engineered cases really are ties / equalities
    u = g["ref_uv"][:, 0]
    fin = np.isfinite(u)
    assert np.any(np.abs(u[fin] - np.floor(u[fin])) == 0.5)
    assert np.any(g["ref_depth"] == 2.0) and not g["ref_vis"][g["ref_depth"] == 2.0].any()


def test_tracks():
    g = GoldenScene.__new__(GoldenScene)
    import os
    from golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "tracks.npz"))
    world = O.tracks_cam_to_world(z["tracks_XYZ"], z["extrinsics_w2c"])
    assert f64_ok(world, z["ref_world"])
    hw = tuple(int(v) for v in z["image_hw"])
    for (f1, f2, p), kept, rec in zip(z["pairs"], z["kept"], z["records_json"]):
        got = O.object_displacement(world, z["tracks_XYZ"], z["extrinsics_w2c"], z["fx_fy_cx_cy"], hw,
                                    int(f1), int(f2), int(p))
        assert (got is not None) == bool(kept)
        if got is None:
            continue
        ref = json.loads(str(rec))
        assert tuple(ref["p1"]) == got["p1"] and tuple(ref["p2"]) == got["p2"]
        assert ref["point_moving"] == got["point_moving"] and ref["cam_moving"] == got["cam_moving"]
        assert f64_ok(got["gt_vector"], ref["gt_value"])
    d, f1, f2 = O.point_pair_distances(world, z["visibility"], 3)
    vf = np.where(z["visibility"][:, 3])[0]
    assert len(d) == len(vf) * (len(vf) - 1) // 2


def test_relative_pose_256_pairs():
    """the reference's CME answer_values for 256 pairs (both swap branches, wrapped yaw differences)."""
    import os
    from golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "cme256.npz"))
    E = [O.aligned_extrinsic(z["A"], e) for e in z["E"]]
    wrapped = 0
    for (a, b, yaw, pitch, dist), swap, ans in zip(z["rows"], z["swap"], z["answers_json"]):
        ref = json.loads(str(ans))
        got = O.relative_pose_answer_values(E[int(a)], E[int(b)], yaw, pitch, bool(swap))
        dv_ref, dv_got = ref.pop("displacement_vector"), got.pop("displacement_vector")
        assert got == ref and f64_ok(dv_got, dv_ref)
        wrapped += abs(yaw) > 180
    assert wrapped >= 20 and 60 < int(z["swap"].sum()) < 200
