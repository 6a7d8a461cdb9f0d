"""Minimal image combinations that cover an object's height / length / width (object_perception).

Reference: spatial_engine/object_perception/single_object_coverage_finder.py (COV).  Upstream carries a
boolean union mask over all scene vertices through a breadth-first search and takes
``max(coords) - min(coords)`` of the masked vertices at every node (COV:56-65,143-146).  The extent of a
union is the min / max of the per-image extents, so here the GPU reduces every (object, image) to six numbers
once (K8, ``engine.object_extents``) and the search itself is a few float comparisons per node.  The search
below keeps the reference's order of visits, its pruning rules and its draws from ``random`` (the 25-image
cap COV:118-119 and the 5000-node cap COV:207-209), so with the same seed it returns the same combinations.
"""
from __future__ import annotations

import random as _random
from typing import Dict, List, Sequence, Tuple

import numpy as np

TOLERANCE = 0.1          # COV:37
MAX_IMAGES = 5           # COV:85 default
MAX_FIRST_LAYER = 25     # COV:118
MAX_LEVEL_NODES = 5000   # COV:207


def covers_dimension(coverage, target, tolerance) -> bool:
    """COV:67-73."""
    if coverage is None:
        return False
    return bool(abs(coverage - target) <= tolerance * target)


def _coverage(lo, hi):
    """max - min of a (possibly empty) set given its extent; None for the empty set (COV:61-62)."""
    if not (lo <= hi):
        return None
    return hi - lo


def minimal_combinations(images: Sequence[str], lo: Sequence, hi: Sequence, target_dim, tolerance=TOLERANCE,
                         max_images: int = MAX_IMAGES, rng=_random) -> Dict[int, List[Tuple[str, ...]]]:
    """{k: [minimal k-image combinations]} for one object and one axis (COV:76-220).

    ``images`` are the candidate image ids in the reference's order (object_to_images order, already reduced
    to those present in the visibility index); ``lo[i]`` / ``hi[i]`` the extent along the axis of the object
    vertices image i sees (lo > hi when it sees none), as NumPy scalars of the scene points' dtype so that
    the subtraction and comparison round exactly as upstream's.
    """
    images = list(images)
    order = list(range(len(images)))
    if len(order) > MAX_FIRST_LAYER:
        order = rng.sample(order, MAX_FIRST_LAYER)        # same draws as random.sample(valid_images, 25)
    names = [images[i] for i in order]
    lo = [lo[i] for i in order]
    hi = [hi[i] for i in order]
    n = len(names)

    def covers(l, h):
        return covers_dimension(_coverage(l, h), target_dim, tolerance)

    # extent of images i..n-1 together (COV:122-127)
    tail_lo, tail_hi = [None] * n, [None] * n
    for i in range(n - 1, -1, -1):
        if i == n - 1:
            tail_lo[i], tail_hi[i] = lo[i], hi[i]
        else:
            tail_lo[i], tail_hi[i] = min(lo[i], tail_lo[i + 1]), max(hi[i], tail_hi[i + 1])

    found: List[int] = []                                  # member bitmasks of the minimal sets so far
    level = [((names[i],), lo[i], hi[i], i, 1 << i) for i in range(n)]
    first_layer: List[tuple] = []
    solutions: Dict[int, List[Tuple[str, ...]]] = {}
    k = 1
    while k <= max_images and level:
        expand, fresh = [], []
        for node in level:
            comb, l, h, last, members = node
            if any((m & members) == m for m in found):     # superset of a known minimal set (COV:132-142)
                continue
            if covers(l, h):
                fresh.append(members)
                solutions.setdefault(k, []).append(tuple(comb))
                continue
            if last < n - 1 and not covers(min(l, tail_lo[last]), max(h, tail_hi[last])):
                continue                                   # even with every later image it cannot match (COV:181-184)
            expand.append(node)
            if k == 1:
                first_layer.append(node)
        found.extend(fresh)
        nxt = []
        if k < max_images:
            for comb, l, h, last, members in expand:
                for c1, l1, h1, last1, m1 in first_layer:
                    if last1 > last:
                        nxt.append((comb + c1, min(l, l1), max(h, h1), last1, members | m1))
        if len(nxt) > MAX_LEVEL_NODES:
            nxt = rng.sample(nxt, MAX_LEVEL_NODES)
        level = nxt
        k += 1
    return solutions


class SceneExtents:
    """K8 results of one scene on the host: extents[obj][image_id] per axis, in the scene points' dtype."""

    def __init__(self, image_ids: Sequence[str], object_ids: Sequence[int], lo: np.ndarray, hi: np.ndarray,
                 count: np.ndarray):
        self.image_index = {img: k for k, img in enumerate(image_ids)}
        self.object_index = {o: k for k, o in enumerate(object_ids)}
        self.lo, self.hi, self.count = lo, hi, count

    def axis(self, object_id, images: Sequence[str], axis: int):
        o = self.object_index[object_id]
        cols = [self.image_index[i] for i in images]
        return [self.lo[o, c, axis] for c in cols], [self.hi[o, c, axis] for c in cols]


def scene_extents(image_bits, image_ids: Sequence[str], scene_pts, object_point_indices: Dict[int, np.ndarray],
                  dtype=None) -> SceneExtents:
    """One K8 launch for every (object, image) of a scene.  ``image_bits`` [F, n_words] int64 on the GPU
    (K1's bitsets or ``scene.pack_index_lists`` of the parquet lists); ``scene_pts`` [V, >=3] float32/64 on the
    host, or the resident [V,3] float64 device tensor (then ``dtype`` names the dtype the points had on disk)."""
    import torch
    from . import engine
    dev = image_bits.device
    if isinstance(scene_pts, torch.Tensor):
        xyz = scene_pts
        dtype = np.dtype(dtype or np.float64)
        pts = xyz
    else:
        pts = np.asarray(scene_pts)[:, :3]
        dtype = np.dtype(dtype) if dtype is not None else (pts.dtype if pts.dtype in (np.float32, np.float64) else np.dtype(np.float64))
        xyz = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float64)).to(dev)     # float32 -> float64 is exact
    objs = list(object_point_indices.items())
    offsets = np.zeros(len(objs) + 1, dtype=np.int64)
    for k, (_, idx) in enumerate(objs):
        offsets[k + 1] = offsets[k] + len(idx)
    verts = (np.concatenate([np.asarray(idx, dtype=np.int64) for _, idx in objs]) if objs else np.zeros(0, np.int64))
    if verts.size and (verts.min() < 0 or verts.max() >= pts.shape[0]):
        raise ValueError("object vertex index outside the scene points")
    if offsets[-1] >= 2 ** 31:
        raise ValueError("too many object vertices for one launch")
    lo, hi, count = engine.object_extents(image_bits, xyz, torch.from_numpy(offsets.astype(np.int32)).to(dev),
                                          torch.from_numpy(verts.astype(np.int32)).to(dev))
    return SceneExtents(image_ids, [o for o, _ in objs], lo.cpu().numpy().astype(dtype), hi.cpu().numpy().astype(dtype),
                        count.cpu().numpy())


def object_coverage(ext: SceneExtents, object_id, visible_images: Sequence[str], height_target, length_target,
                    width_target, width_axis: int, tolerance=TOLERANCE, rng=_random) -> Dict[str, dict]:
    """process_object (COV:222-265): height on axis 2, width on ``width_axis``, length on the other one."""
    images = [i for i in visible_images if i in ext.image_index]
    length_axis = 1 if width_axis == 0 else 0
    out = {}
    for name, axis, target in (("height", 2, height_target), ("length", length_axis, length_target),
                               ("width", width_axis, width_target)):
        lo, hi = ext.axis(object_id, images, axis)
        out[name] = minimal_combinations(images, lo, hi, target, tolerance, rng=rng)
    return out
