"""K5a / K5b / K5c / K7 against the oracle at the sizes the bench times and the datasets have: SURVEY.md 8d's TAPVid-3D
block (T = 300 frames x P = 256 tracks, single_object_movement_engine_coord.py:441-454) and one ADT-length block (T = 1 024).
Reference functions: OM_C:446-454 (camera -> world), OM_C:293-315 (project_point), OM_C:476-498 (all-frame-pair
displacement), OM_C:324-376 (object displacement), OM_C:49-92 (rigid_body_segmentation's accumulation)."""
import numpy as np
import pytest
import torch

from golden_util import close_f64, same_f64
from mspa import engine, synth
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def f64_ok(a, b):
    return same_f64(a, b) or close_f64(a, b, rtol=1e-12, scale=1e-9)          # north_star: float 3D quantities within 1e-5 rel


@pytest.mark.parametrize("T,P,seed", [(300, 256, 910), (1024, 256, 911)], ids=["tapvid_300x256", "adt_1024x256"])
def test_track_kernels_at_dataset_sizes(T, P, seed):
    tr = synth.make_tracks(seed, T=T, P=P, n_groups=8)
    H, W = tr.image_hw
    xyz = np.ascontiguousarray(tr.tracks_XYZ)
    # a handful of points pushed behind the camera / outside the image / onto z = -1e-8 (the reference's epsilon)
    xyz[5, 7] = [0.3, 0.2, -1.0]
    xyz[6, 7] = [50.0, 0.0, 1.0]
    xyz[7, 7] = [0.0, 0.0, -1e-8]
    xyz[8, 7] = [0.0, 0.0, 0.0]
    tracks = torch.from_numpy(xyz).to(DEV)
    c2w_np = np.linalg.inv(tr.extrinsics_w2c)                                   # OM_C:448, on the host
    c2w = torch.from_numpy(c2w_np.reshape(T, 16)).to(DEV)
    w2c = torch.from_numpy(np.ascontiguousarray(tr.extrinsics_w2c).reshape(T, 16)).to(DEV)
    # ---- K5a: every (frame, point) ------------------------------------------------------------------------------------
    res = engine.track_to_world(tracks, c2w, tr.fx_fy_cx_cy, (H, W))
    torch.cuda.synchronize()
    world = res["world"].cpu().numpy()
    ref_world = O.tracks_cam_to_world(xyz, tr.extrinsics_w2c)
    assert world.shape == (T, P, 3) and f64_ok(world, ref_world)
    # project_point for ALL T x P points, vectorised restatement of the same expressions (checked against the oracle's scalar
    # function on a subsample below)
    fx, fy, cx, cy = tr.fx_fy_cx_cy
    with np.errstate(divide="ignore", invalid="ignore"):
        un = ((fx * xyz[..., 0] / (xyz[..., 2] + 1e-8)) + cx) / W
        vn = ((fy * xyz[..., 1] / (xyz[..., 2] + 1e-8)) + cy) / H
    ok_ref = (0 <= un) & (un < 1) & (0 <= vn) & (vn < 1) & (xyz[..., 2] > 0)
    ok = res["ok"].cpu().numpy().astype(bool)
    uvn = res["uvn"].cpu().numpy()
    assert np.array_equal(ok, ok_ref) and not ok[5, 7] and not ok[6, 7] and not ok[7, 7] and not ok[8, 7]
    assert same_f64(uvn[ok][:, 0], un[ok]) and same_f64(uvn[ok][:, 1], vn[ok])    # bit-exact: same operation order
    rng = np.random.default_rng(seed)
    for t, p in zip(rng.integers(0, T, 300), rng.integers(0, P, 300)):
        r = O.project_point(xyz[t, p], tr.fx_fy_cx_cy, H, W)
        assert (r is not None) == bool(ok[t, p]) and (r is None or same_f64(uvn[t, p], r))
    # ---- K5c: all frame pairs of 40 points (up to T (T-1) / 2 distances each) --------------------------------------------
    pts = rng.choice(P, 40, replace=False).tolist()
    frames = [np.where(tr.visibility[:, p])[0] for p in pts]
    got = engine.track_pair_distances(res["world"], pts, frames)
    n_total = 0
    for p, d in zip(pts, got):
        want, _f1, _f2 = O.point_pair_distances(world, tr.visibility, p)
        assert d.shape == want.shape and np.array_equal(d, want)               # np.linalg.norm(axis=1) form, bit for bit
        n_total += len(d)
    assert n_total > 40 * (T // 3) ** 2 // 4
    # ---- K5b: 4 000 random (frame1, frame2, point) triples -----------------------------------------------------------------
    trip = np.stack([rng.integers(0, T, 4000), rng.integers(0, T, 4000), rng.integers(0, P, 4000)], 1).astype(np.int32)
    trip[:50, 1] = trip[:50, 0]                                                 # same frame twice: nothing moves
    trip[50:60, 2] = 7
    trip[50:60, 0] = np.arange(0, 10)                                           # through the doctored points
    disp, flags = engine.track_displacement(res["world"], w2c, c2w, torch.from_numpy(trip).to(DEV))
    disp, flags = disp.cpu().numpy(), flags.cpu().numpy()
    n_checked = 0
    for k, (f1, f2, p) in enumerate(trip.tolist()):
        o = O.object_displacement(world, xyz, tr.extrinsics_w2c, tr.fx_fy_cx_cy, (H, W), f1, f2, p)
        assert (o is None) == (not (ok[f1, p] and ok[f2, p]))                   # OM_C:360-362 skip rule from K5a's flags
        dref = np.linalg.norm(world[f2, p] - world[f1, p])
        assert flags[k, 0] == int(not (dref < 0.01))
        if o is None:
            continue
        assert flags[k, 0] == o["point_moving"] and flags[k, 1] == o["cam_moving"]
        assert f64_ok(disp[k, 1:4], o["gt_vector"]) and int(disp[k, 0] * 1000) == o["gt_total_distance"]
        n_checked += 1
    assert n_checked > 2000
    # ---- K7: the T x P^2 accumulation ------------------------------------------------------------------------------------
    loss = engine.track_rigidity_loss(tracks, 0.01).cpu().numpy()
    want = O.rigidity_loss(xyz, 0.01)
    assert np.array_equal(loss, want) and np.array_equal(loss, loss.T) and (np.diag(loss) == 0).all() and (loss > 0).any()
