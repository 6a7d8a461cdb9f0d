"""Object-movement host logic without a GPU: the oracle's restatements against the frozen reference outputs, and the
product's vectorised pair mining against the oracle's list-and-loop form."""
import json
import os
import random

import numpy as np

from golden_util import GOLDEN_DIR
from mspa import heads, synth
from oracle import np_oracle as O


def test_rigidity_loss_oracle_reproduces_reference_groups():
    """O.rigidity_loss + SciPy linkage == the groups the reference's rigid_body_segmentation returned (tracks.npz)."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.spatial.distance import pdist, squareform
    z = np.load(os.path.join(GOLDEN_DIR, "tracks.npz"))
    pts = z["tracks_XYZ"]
    loss = O.rigidity_loss(pts)
    ref = np.zeros_like(loss)
    for t in range(1, pts.shape[0]):                       # upstream's own expression (SciPy pdist), OM_C:66-78
        ch = np.abs(squareform(pdist(pts[t])) - squareform(pdist(pts[t - 1])))
        ref += np.where(ch > 0.01, ch, 0)
    assert np.allclose(loss, ref, rtol=1e-12, atol=1e-15)
    labels = fcluster(linkage(squareform(loss, checks=False), method="average"), 0.1, criterion="distance")
    groups = [np.where(labels == i)[0].tolist() for i in range(1, max(labels) + 1)]
    assert groups == json.loads(str(z["groups_json"]))


def _numpy_distance_fn(world):
    def fn(points, frames):
        out = []
        for p, fr in zip(points, frames):
            ii, jj = np.triu_indices(len(fr), 1)
            out.append(np.linalg.norm(world[fr[jj], p] - world[fr[ii], p], axis=1))
        return out
    return fn


def test_pair_mining_matches_oracle():
    tr = synth.make_tracks(44, T=90, P=40, n_groups=3)
    world = O.tracks_cam_to_world(tr.tracks_XYZ, tr.extrinsics_w2c)
    rng = np.random.default_rng(1)
    base = [sorted(rng.choice(40, size=n, replace=False).tolist()) for n in (12, 9, 7)]
    vis = tr.visibility.copy()
    vis[:, base[0][0]] = False                                   # a point that is never visible
    vis[:, base[1][0]] = False
    vis[5, base[1][0]] = True                                    # ... and one seen in a single frame
    for npoints, npairs, augment, ratio in ((15, 30, True, 0.05), (1, 1, False, 1.0), (5, 1e8, True, 1.0), (3, 2, True, 0.3)):
        want = O.mine_frame_pairs(world, vis, [list(g) for g in base], npoints, npairs, augment, ratio, random.Random(9))
        got_rng = random.Random(9)
        got = heads.object_movement_mine_pairs(vis, [list(g) for g in base], _numpy_distance_fn(world), npoints, npairs,
                                               augment, ratio, got_rng)
        key = lambda s: (int(s["point_index"]), int(s["frame1"]), int(s["frame2"]))   # noqa: E731
        assert [key(s) for s in got] == [key(s) for s in want] and len(got) > 0
