"""The two parameters the reference's interface leaves open and round 2 narrowed (missing 4):
``project_points`` with ANY homogeneous coordinate (IH:46-72 takes an arbitrary [N, 4] array) and
``SceneInfoHandler(depth_value_scale=...)`` with any scale (IH:76, applied at IH:368) -- HIP path vs the NumPy oracle."""
import numpy as np
import pytest
import torch

from mspa import engine, synth, _lib
from oracle import np_oracle as O
from test_gpu_facade import facade

pytestmark = pytest.mark.gpu
DEV = "cuda"


def scene(hw=(96, 128), dhw=None, seed=3030):
    return synth.make_scene(seed, n_points=4096, n_frames=3, color_hw=hw, depth_hw=dhw or hw, invalid_pose_frac=0.0,
                            with_color=False)


def test_project_points_general_homogeneous_coordinate():
    ns = facade()
    sc = scene()
    rng = np.random.default_rng(5)
    i = sc.valid_image_ids[1]
    E = sc.A @ sc.E[i]
    xyz = sc.points[:2048, :3]
    w = rng.uniform(0.25, 4.0, len(xyz))
    w[::7] = 1.0                                   # the value every call site of the reference passes
    w[1::97] = -2.0
    w[2::211] = 0.0                                # a point at infinity: finite (u, v) from the direction alone
    pts = np.hstack([xyz * w[:, None], w[:, None]])          # the same Euclidean points, scaled homogeneous rows
    uv_ref, d_ref = O.project_points(pts, sc.K, E)
    uv, d = ns.IH.project_points(pts, sc.K, E)
    assert uv.dtype == np.float64 and uv.shape == uv_ref.shape
    with np.errstate(invalid="ignore"):
        assert np.allclose(uv, uv_ref, rtol=1e-12, atol=0, equal_nan=True)
        assert np.allclose(d, d_ref, rtol=1e-12, atol=0, equal_nan=True)
    # bit for bit where the reference's own inputs live (w == 1), and equal to the affine kernels' result there
    one = w == 1.0
    uv1, d1 = ns.IH.project_points(np.hstack([xyz[one], np.ones((one.sum(), 1))]), sc.K, E)
    assert np.array_equal(uv[one], uv1) and np.array_equal(d[one], d1)
    assert np.array_equal(uv[one], uv_ref[one]) and np.array_equal(d[one], d_ref[one])
    # depth follows the homogeneous scale (camera_coords[2] of E_inv @ p), as in the reference
    assert np.allclose(d[w > 0] / w[w > 0], O.project_points(np.hstack([xyz, np.ones((len(xyz), 1))]), sc.K, E)[1][w > 0], rtol=1e-9)
    # device tensors in, device tensors out
    uv_t, d_t = ns.IH.project_points(torch.from_numpy(pts).cuda(), sc.K, E)
    assert uv_t.is_cuda and np.array_equal(uv_t.cpu().numpy(), uv, equal_nan=True)
    with pytest.raises(ValueError):
        ns.IH.project_points(pts[:, :3], sc.K, E)


@pytest.mark.parametrize("scale", [0.00025, 0.001, 0.004])
@pytest.mark.parametrize("shape", ["ident", "scaled"])
def test_depth_value_scale_through_k1_k6_and_the_predicates(scale, shape):
    """K1 (bits / mask / count, exact kernel whenever the scale is not the millimetre), K6b and the three predicates with the
    handler's depth_value_scale: masks bit-exact vs the oracle evaluated with the same scale."""
    hw, dhw = ((96, 128), (96, 128)) if shape == "ident" else ((121, 162), (60, 80))
    sc = scene(hw, dhw, seed=3131)
    ids = sc.valid_image_ids
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), DEV)
    Ea = [sc.A @ sc.E[i] for i in ids]
    cam = torch.from_numpy(engine.camera_matrices(sc.K, Ea)).to(DEV)
    xyz = torch.from_numpy(np.ascontiguousarray(sc.points[:, :3])).to(DEV)
    out = engine.vertex_visibility(xyz, cam, depth, hw, ("bits", "mask", "count"), depth_scale=scale)
    torch.cuda.synchronize()
    masks = out["mask"].cpu().numpy().astype(bool)
    n = sc.points.shape[0]
    differs_from_mm = False
    for k, i in enumerate(ids):
        ref, uv, d = O.vertex_visibility(sc.points[:, :3], sc.K, Ea[k], sc.depth[i], hw, scale)
        assert np.array_equal(masks[k], ref), f"image {i}"
        bits = np.unpackbits(out["bits"][k].cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
        assert np.array_equal(bits, ref) and int(out["count"][k]) == int(ref.sum())
        differs_from_mm |= not np.array_equal(ref, O.vertex_visibility(sc.points[:, :3], sc.K, Ea[k], sc.depth[i], hw)[0])
        # the predicates on already-projected points
        got = engine.check_visibility(torch.from_numpy(np.ascontiguousarray(uv)).to(DEV), torch.from_numpy(np.ascontiguousarray(d)).to(DEV), depth[k], hw,
                                      ("by_depth", "visible"), depth_scale=scale)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(got["by_depth"].cpu().numpy().astype(bool), O.check_point_visibility_by_depth(uv, d, sc.depth[i], hw, scale))
        assert np.array_equal(got["visible"].cpu().numpy().astype(bool), ref)
    assert differs_from_mm == (scale != 0.001), "a different scale must change some decision on this scene (and only then)"
    # K6b
    rng = np.random.default_rng(9)
    smp = np.stack([rng.integers(0, n, 500), rng.integers(0, len(ids), 500)], 1).astype(np.int32)
    uv, d, ok = engine.project_samples(xyz, cam, depth, hw, torch.from_numpy(smp).to(DEV), depth_scale=scale)
    ok = ok.cpu().numpy().astype(bool)
    for s, (v, k) in enumerate(smp):
        ref, uv_r, d_r = O.vertex_visibility(sc.points[v:v + 1, :3], sc.K, Ea[k], sc.depth[ids[k]], hw, scale)
        assert ok[s] == bool(ref[0])
    with pytest.raises(_lib.MspaError):
        engine.vertex_visibility(xyz, cam, depth, hw, ("bits",), depth_scale=0.0)


def test_handler_takes_any_depth_value_scale(tmp_path):
    ns = facade()
    sc = scene(seed=3232)
    sid = "scene_scale_00"
    posed, inst = str(tmp_path / "posed_images"), str(tmp_path / "scannet_instance_data")
    import os
    os.makedirs(os.path.join(inst, sid))
    np.save(os.path.join(inst, sid, "aligned_points.npy"), sc.points)
    H, W = sc.color_hw
    for i in sc.image_ids:
        ns.IMG.register(os.path.join(posed, sid, f"{i}.jpg"), np.zeros((H, W, 3), np.uint8))
        ns.IMG.register(os.path.join(posed, sid, f"{i}.png"), sc.depth[i])
    infos = {sid: {"num_posed_images": len(sc.image_ids), "intrinsic_matrix": sc.K, "axis_align_matrix": sc.A,
                   "num_objects": 0, "images_info": {i: {"extrinsic_matrix": sc.E[i]} for i in sc.image_ids}}}
    i = sc.valid_image_ids[0]
    for scale in (0.001, 0.0005):
        h = ns.IH.SceneInfoHandler(infos, posed_images_root=posed, instance_data_root=inst, depth_value_scale=scale)
        assert h.depth_value_scale == scale
        uv, d = h.project_3d_point_to_image(sid, i, sc.points[:, :3])
        ref = O.check_point_visibility(uv, d, sc.depth[i], sc.color_hw, scale)
        assert np.array_equal(h.check_point_visibility(sid, i, uv, d), ref)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(h.check_point_visibility_by_depth(sid, i, uv, d),
                                  O.check_point_visibility_by_depth(uv, d, sc.depth[i], sc.color_hw, scale))
        dev = h.scene_on_device(sid)
        assert dev.depth_scale == scale
        m = dev.vertex_visibility(("mask",))["mask"][dev.index[i]].cpu().numpy().astype(bool)
        assert np.array_equal(m, ref)
    with pytest.raises(ValueError):
        ns.IH.SceneInfoHandler(infos, depth_value_scale=0)
