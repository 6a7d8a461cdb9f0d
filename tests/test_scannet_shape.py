"""The reference at ScanNet's own shapes (colour 1296x968 over depth 640x480): tests/golden/scannet_shape.npz holds the
vertex projections / masks as arrays and the full-frame back-projection + one composite pair as SHA-256 of the reference's
float64 bytes (oracle/gen_golden.py golden_scannet_shape).  The oracle must reproduce them on the CPU; the exact HIP kernel
must reproduce the very same bytes on the GPU."""
import hashlib

import numpy as np
import pytest

from golden_util import GoldenScene, same_f64
from oracle import np_oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def stable_color_image(hw):
    H, W = hw
    return ((np.arange(H * W * 3, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)).astype(np.uint8).reshape(H, W, 3)


@pytest.fixture(scope="module")
def g():
    return GoldenScene("scannet_shape")


def test_oracle_reproduces_reference_bytes(g):
    assert g.color_hw == (968, 1296) and g.depth_hw == (480, 640) and len(g.valid_image_ids) == 2
    pts = g.points[:, :3]
    for k, image_id in enumerate(g.valid_image_ids):
        Ea = O.aligned_extrinsic(g.A, g.E[image_id])
        uv, d = O.project_3d_point_to_image(pts, g.K, Ea)
        assert same_f64(uv, g["ref_uv"][k]) and same_f64(d, g["ref_depth"][k])
        vis = O.check_point_visibility(uv, d, g.depth[image_id], g.color_hw)
        assert np.array_equal(vis, g["ref_vis"][k]) and vis.sum() > 50
    f0, f1 = (str(x) for x in g["pair_ids"][0])
    color = stable_color_image(g.color_hw)
    a7 = O.project_mask_to_3d(g.depth[f0], g.K, g.E[f0], None, g.A, color)
    assert a7.shape == (int(g["a7_rows"]), 6) and same_f64(a7[:64], g["a7_head"])
    assert sha(a7[:, :3]) == str(g["a7_sha_xyz"]) and sha(a7[:, 3:]) == str(g["a7_sha_rgb"]) and sha(a7) == str(g["a7_sha_all"])
    uv, d = O.project_3d_point_to_image(a7[:, :3], g.K, O.aligned_extrinsic(g.A, g.E[f1]))
    vis = O.check_point_visibility(uv, d, g.depth[f1], g.color_hw)
    assert sha(uv) == str(g["pair_sha_uv"]) and sha(d) == str(g["pair_sha_depth"])
    assert sha(vis) == str(g["pair_sha_vis"]) and int(vis.sum()) == int(g["pair_n_vis"])


@pytest.mark.gpu
def test_exact_kernel_reproduces_reference_bytes(g):
    """K3 (exact path, float64 outputs) over the 1296x968 colour grid of a 640x480 depth frame: the valid rows, in row-major
    order, ARE the reference's arrays -- same SHA-256."""
    import torch
    from mspa import engine
    f0, f1 = (str(x) for x in g["pair_ids"][0])
    depth = engine.depth_to_device(np.stack([g.depth[f0], g.depth[f1]]), "cuda")
    mats = torch.from_numpy(engine.frame_matrices(g.K, g.A, [g.E[f0], g.E[f1]])).cuda()
    rgb = torch.from_numpy(np.stack([stable_color_image(g.color_hw)] * 2)).cuda()
    pairs = torch.tensor([[0, 1]], dtype=torch.int32, device="cuda")
    out = engine.alloc_pair_outputs(1, g.color_hw, ("valid_u8", "vis_u8", "vis_bits", "xyz_f64", "uv_f64", "depth_f64", "rgba", "counts"),
                                    "cuda")
    engine.pair_reproject(depth, mats, pairs, g.color_hw, out, rgb=rgb)
    valid = out["valid_u8"][0].cpu().numpy().astype(bool)
    assert int(valid.sum()) == int(g["a7_rows"]) == int(out["counts"][0, 0])
    xyz = out["xyz_f64"][0].cpu().numpy()[valid]
    assert same_f64(xyz[:64], g["a7_head"][:, :3])
    assert sha(xyz) == str(g["a7_sha_xyz"])
    rgba = out["rgba"][0].cpu().numpy()[valid]
    cols = np.stack([rgba & 0xFF, (rgba >> 8) & 0xFF, (rgba >> 16) & 0xFF], axis=1).astype(np.float64)
    assert sha(cols) == str(g["a7_sha_rgb"])
    assert sha(out["uv_f64"][0].cpu().numpy()[valid]) == str(g["pair_sha_uv"])
    assert sha(out["depth_f64"][0].cpu().numpy()[valid]) == str(g["pair_sha_depth"])
    vis = out["vis_u8"][0].cpu().numpy().astype(bool)
    assert sha(vis[valid]) == str(g["pair_sha_vis"]) and int(vis.sum()) == int(g["pair_n_vis"]) == int(out["counts"][0, 1])
    # the fast path at this shape (width 1296 = 20.25 x 64: linear pixel mapping, colour grid != depth grid): same integers
    from mspa import _lib
    ex = engine.alloc_pair_outputs(2, g.color_hw, ("vis_bits", "vis_u8", "valid_u8", "pix_i16", "counts"), "cuda")
    fa = engine.alloc_pair_outputs(2, g.color_hw, ("vis_bits", "vis_u8", "valid_u8", "pix_i16", "counts"), "cuda")
    both = torch.tensor([[0, 1], [1, 0]], dtype=torch.int32, device="cuda")
    engine.pair_reproject(depth, mats, both, g.color_hw, ex, rgb=rgb)
    engine.pair_reproject(depth, mats, both, g.color_hw, fa, rgb=rgb, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    for k in ex:
        assert torch.equal(ex[k], fa[k]), f"fast (linear mapping) differs from exact in {k}"
    assert int(fa["counts"][0, 1]) == int(g["pair_n_vis"])
    # the scene kernels at these shapes
    from mspa.scene import SceneOnDevice
    scene = SceneOnDevice(g.K, g.A, g.E, g.depth, g.color_hw, g.points, "cuda")
    assert scene.ids == g.valid_image_ids
    res = scene.vertex_visibility(("mask", "uv", "depth"))
    assert np.array_equal(res["mask"].cpu().numpy().astype(bool), g["ref_vis"])
    assert same_f64(res["uv"].cpu().numpy(), g["ref_uv"]) and same_f64(res["depth"].cpu().numpy(), g["ref_depth"])
    table = scene.frames_relations()
    for key, want in zip(g["cfr_pairs"], g["cfr_values"]):
        got = table[(str(key[0]), str(key[1]))]
        assert same_f64([got["overlap"], got["distance"], got["yaw"], got["pitch"]], want)
