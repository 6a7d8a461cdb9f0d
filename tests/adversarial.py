"""Adversarial pose / depth generators shared by the GPU parity tests and the CPU emulation tools (test infrastructure).

* ``adversarial_pairs``   looking away, coincident, half-pixel shifts, grazing corners (round 1's recipe)
* ``near_plane_case``     camera 2 centred ``delta`` metres behind a back-projected frame-1 point, the point engineered onto
                          rounding ties and integer bounds +- 2e-6 px: the regime where the fast kernels' evaluation error grows
                          like 1 / (camera-2 depth) and round 3's constant guard band failed
* ``translated``          the same scene with world coordinates shifted by ``t``: E' = T E, A' = A inv(T) (aligned coordinates
                          unchanged, the reference's world-space intermediate huge)
* ``near_vertex_cameras`` K1: cameras centred ``delta`` behind scene vertices
"""
import numpy as np

from mspa import synth
from oracle import np_oracle as O


def look_at(eye, tgt):
    return synth._look_at(np.asarray(eye, float), np.asarray(tgt, float))


def render_mm(E_al, Kd, dhw, boxes, rng, noise=4.0, holes=0.07):
    z = synth.render_depth(E_al, Kd, dhw, boxes)
    mm = np.clip(np.rint(z * 1000.0 + rng.normal(0, noise, z.shape)), 0, 65535).astype(np.uint16)
    mm[rng.random(mm.shape) < holes] = 0
    return mm


def depth_intrinsics(K, hw, dhw):
    Kd = K.copy()
    Kd[0] *= dhw[1] / hw[1]
    Kd[1] *= dhw[0] / hw[0]
    return Kd


def adversarial_pairs(rng, n, hw):
    K = synth.intrinsics_for(hw)
    A = np.eye(4)
    E = []
    for k in range(n):
        kind = k % 6
        eye = rng.uniform([1, 1, 1.2], [5, 5, 1.9])
        tgt = synth.ROOM / 2 + rng.normal(0, 1.0, 3) * [1, 1, 0.4]
        if kind == 1:
            tgt = eye + (eye - tgt)                       # looking the other way
        e = look_at(eye, tgt)
        if kind == 2 and E:
            e = E[-1].copy()                              # coincident with the previous camera
        if kind == 3 and E:
            e = E[-1].copy()
            e[:3, 3] += e[:3, 0] * (0.5 / K[0, 0]) * 2.0   # exact half-pixel shift at z = 2
        if kind == 4:
            e[:3, 3] = rng.uniform([0.05, 0.05, 0.1], [0.3, 0.3, 0.4])   # in a corner, grazing the walls
        E.append(synth._roundtrip_f(e))
    return K, A, E


def near_plane_case(rng, hw, deltas, per_delta=5, dhw=None):
    """Frame 0 is a plain view of the room; every further camera sits ``delta`` metres behind a back-projected frame-0 point
    (along its own optical axis, random direction), shifted sideways so that the point lands on a rounding tie, an image
    bound or 2e-6 px off one.  Poses are NOT rounded to six decimals: the regime needs sub-micrometre placement.
    Returns K, A, E, depth (list of uint16 frames on the depth grid), pairs [(0, k)]."""
    dhw = dhw or hw
    H, W = hw
    K = synth.intrinsics_for(hw)
    Kd = depth_intrinsics(K, hw, dhw)
    A = np.eye(4)
    boxes = synth._make_boxes(rng)
    e1 = look_at(rng.uniform([1, 1, 1.2], [5, 5, 1.9]), synth.ROOM / 2)
    E = [e1]
    depth = [render_mm(A @ e1, Kd, dhw, boxes, rng)]
    pts = O.project_mask_to_3d(depth[0], K, e1, np.ones(hw, bool), A, None)
    pairs = []
    for delta in deltas:
        for k in range(per_delta):
            p = pts[int(rng.integers(0, pts.shape[0])), :3]
            e2 = look_at(p, p + rng.normal(0, 1, 3))                     # camera AT the point, random direction
            tu = [W / 2 + 0.5, 0.0, W - 1e-6, 10.5 + 2e-6, 3.0 - 2e-6][k % 5]
            tv = [H / 2 - 0.5, H / 3, 0.0 + 2e-6, H - 2e-6, 7.5][k % 5]
            off = np.array([(tu - K[0, 2]) * delta / K[0, 0], (tv - K[1, 2]) * delta / K[1, 1], delta])
            e2[:3, 3] = p - e2[:3, :3] @ off                               # camera-2 coordinates of p are `off`
            E.append(e2)
            depth.append(render_mm(A @ e2, Kd, dhw, boxes, rng))
            pairs.append((0, len(E) - 1))
    return K, A, E, depth, pairs


def translated(A, E, t):
    """World coordinates shifted by t: the same aligned scene, huge intermediates in the reference's E-then-A chain."""
    T = np.eye(4)
    T[:3, 3] = t
    Ti = np.eye(4)
    Ti[:3, 3] = -np.asarray(t, float)
    return A @ Ti, [T @ e for e in E]


def near_vertex_cameras(rng, points, K, hw, deltas, per_delta=3):
    """K1: aligned camera->world poses centred ``delta`` behind scene vertices, the vertex engineered onto a rounding tie, an
    image bound or 2e-6 px off one (as in near_plane_case)."""
    H, W = hw
    out = []
    for delta in deltas:
        for k in range(per_delta):
            p = points[int(rng.integers(0, points.shape[0])), :3]
            e = look_at(p, p + rng.normal(0, 1, 3))
            tu = [W / 2 + 0.5, 0.0, W - 1e-6, 10.5 + 2e-6, 3.0 - 2e-6][k % 5]
            tv = [H / 2 - 0.5, H / 3, 0.0 + 2e-6, H - 2e-6, 7.5][k % 5]
            off = np.array([(tu - K[0, 2]) * delta / K[0, 0], (tv - K[1, 2]) * delta / K[1, 1], delta])
            e[:3, 3] = p - e[:3, :3] @ off
            out.append(e)
    return out


def fuzz_poses(seed, n, hw):
    """The randomised sweep's poses (tools/fuzz_parity.py, tests/test_gpu_fuzz.py): n // 2 adversarial + a hand-held walk."""
    rng = np.random.default_rng(seed)
    K, A, E = adversarial_pairs(rng, n // 2, hw)
    eye = rng.uniform([1.5, 1.5, 1.3], [4.5, 4.5, 1.8])
    tgt = synth.ROOM / 2 + rng.normal(0, 0.8, 3) * [1, 1, 0.3]
    for _ in range(n - len(E)):                       # neighbouring views with 10-90 % overlap
        eye = np.clip(eye + rng.normal(0, 0.25, 3) * [1, 1, 0.2], [0.4, 0.4, 0.8], [5.6, 5.6, 2.4])
        tgt = tgt + rng.normal(0, 0.4, 3) * [1, 1, 0.3]
        E.append(synth._roundtrip_f(look_at(eye, tgt)))
    return K, A, E, synth._make_boxes(rng)


def fuzz_depth(job):
    """One depth frame of the sweep: rendered room + 4 mm noise + 7 % invalid pixels + a large hole in every fifth frame."""
    seed, k, Ae, Kd, dhw, boxes = job
    rng = np.random.default_rng(seed * 1000 + k)
    mm = render_mm(Ae, Kd, dhw, boxes, rng)
    if k % 5 == 0:                                    # tiles without a single valid sample
        y0, x0 = rng.integers(0, dhw[0] // 2), rng.integers(0, dhw[1] // 2)
        mm[y0:y0 + dhw[0] // 3, x0:x0 + dhw[1] // 3] = 0
    return mm
