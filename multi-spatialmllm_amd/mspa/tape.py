"""Numeric tapes: what lets the pipeline's final exchange carry NUMBERS and leave the text to rank 0.

A task head is a GPU numeric stage (K2 / K4 / K5 / K6 / K7 / K8 launches) feeding a pure-Python record stage that draws
from ``random`` in the reference's order and fills the chat templates.  The two are interleaved -- a draw decides which
vertex gets projected, a projection decides whether a pair is skipped -- so the record stage cannot simply be handed a
table of numbers.  What CAN be separated is WHERE the numbers come from: on the rank that owns a scene the head runs against
the real kernels and a ``Recorder`` notes every array a launch returns; the tape (float64 rows of fixed width 8) is what
crosses the fabric (``shard.collate_records``: counts, then padded rows, RCCL over xGMI); rank 0 runs the very same head
again with a ``Player`` that serves the taped arrays in order, on a ``ReplayScene`` that holds the scene's host-side
metadata and no device data.  Same code, same generator seeds, same numbers: the records are identical by construction, and
no record text -- only a few dozen bytes of numerics per record -- is ever replicated across ranks.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Sequence

import numpy as np
import torch

WIDTH = 8                         # float64 values per collated row
# engine entry points whose RESULTS are numeric-stage outputs (everything else is passed through untouched)
TAPED = ("pair_pose", "pair_overlap", "select_common_point", "project_samples", "overlap_matrix", "object_extents",
         "track_rigidity_loss", "track_to_world", "track_displacement", "track_pair_distances")
_DTYPES = [np.float64, np.float32, np.int64, np.int32, np.int16, np.uint8, np.bool_]
_KEYS = ["world", "uvn", "ok", "bits", "mask", "uv", "depth", "count"]
_T_ARRAY, _T_TENSOR, _T_NONE, _T_TUPLE, _T_LIST, _T_DICT = 0.0, 1.0, 2.0, 3.0, 4.0, 5.0


def _encode(value, out: List[np.ndarray]):
    """Depth-first encoding of a launch result (tensor / ndarray / None / tuple / list / dict of those) as float64 values."""
    if value is None:
        out.append(np.array([_T_NONE]))
    elif isinstance(value, (torch.Tensor, np.ndarray)):
        a = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else value
        code = next(k for k, d in enumerate(_DTYPES) if a.dtype == d)
        if a.dtype == np.int64 and a.size and np.abs(a).max() >= 2 ** 53:
            raise ValueError("tape: int64 value beyond float64's exact range")
        out.append(np.array([_T_TENSOR if isinstance(value, torch.Tensor) else _T_ARRAY, code, a.ndim, *a.shape], dtype=np.float64))
        out.append(a.astype(np.float64).reshape(-1))
    elif isinstance(value, (tuple, list)):
        out.append(np.array([_T_TUPLE if isinstance(value, tuple) else _T_LIST, len(value)], dtype=np.float64))
        for v in value:
            _encode(v, out)
    elif isinstance(value, dict):
        out.append(np.array([_T_DICT, len(value)], dtype=np.float64))
        for k, v in value.items():
            out.append(np.array([_KEYS.index(k)], dtype=np.float64))
            _encode(v, out)
    else:
        raise TypeError(f"tape: cannot encode a {type(value).__name__}")


class _Reader:
    def __init__(self, flat: np.ndarray, device):
        self.flat, self.pos, self.device = flat, 0, device

    def take(self, n):
        v = self.flat[self.pos:self.pos + n]
        if len(v) != n:
            raise RuntimeError("tape exhausted: the replayed head asked for more numbers than the owner recorded")
        self.pos += n
        return v

    def decode(self):
        kind = self.take(1)[0]
        if kind == _T_NONE:
            return None
        if kind in (_T_TENSOR, _T_ARRAY):
            code, ndim = (int(x) for x in self.take(2))
            shape = tuple(int(x) for x in self.take(ndim))
            a = self.take(int(np.prod(shape)) if shape else 1).astype(_DTYPES[code]).reshape(shape)
            return torch.from_numpy(np.ascontiguousarray(a)).to(self.device) if kind == _T_TENSOR else a
        if kind in (_T_TUPLE, _T_LIST):
            items = [self.decode() for _ in range(int(self.take(1)[0]))]
            return tuple(items) if kind == _T_TUPLE else items
        if kind == _T_DICT:
            n = int(self.take(1)[0])
            return {_KEYS[int(self.take(1)[0])]: self.decode() for _ in range(n)}
        raise RuntimeError(f"tape corrupt: kind {kind}")


class Recorder:
    """Stands in for the ``engine`` module on the owning rank: every call goes to the real kernels; the results of the
    numeric-stage entry points are noted."""

    def __init__(self, real):
        self._real, self._chunks = real, []

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name not in TAPED:
            return fn

        def taped(*args, **kwargs):
            r = fn(*args, **kwargs)
            _encode(r, self._chunks)
            return r
        return taped

    def note(self, value):
        _encode(value, self._chunks)

    def rows(self) -> np.ndarray:
        flat = np.concatenate(self._chunks) if self._chunks else np.zeros(0)
        pad = (-len(flat)) % WIDTH
        return np.concatenate([flat, np.zeros(pad)]).reshape(-1, WIDTH)


class Player:
    """Stands in for the ``engine`` module on rank 0: the numeric-stage entry points return the taped results (tensors on
    ``device``), in the order they were recorded; nothing is launched."""

    def __init__(self, real, rows: np.ndarray, device):
        self._real = real
        self._reader = _Reader(np.ascontiguousarray(rows, dtype=np.float64).reshape(-1), device)

    def __getattr__(self, name):
        if name in TAPED:
            return lambda *args, **kwargs: self._reader.decode()
        return getattr(self._real, name)

    def next(self):
        return self._reader.decode()


def frame(key: int, rows: np.ndarray) -> List[np.ndarray]:
    """One unit's tape as it travels through ``shard.collate_records``: a header row (unit key, number of rows) + the rows."""
    header = np.zeros((1, WIDTH))
    header[0, 0], header[0, 1] = key, len(rows)
    return [header, rows]


def unframe(table: np.ndarray) -> Dict[int, np.ndarray]:
    """The collated table (every rank's framed tapes, rank order) -> {unit key: tape rows}."""
    tapes, pos = {}, 0
    while pos < len(table):
        key, n = int(table[pos, 0]), int(table[pos, 1])
        tapes[key] = table[pos + 1:pos + 1 + n]
        pos += 1 + n
    return tapes


@contextlib.contextmanager
def engine_as(proxy):
    """Route ``engine.<kernel>`` calls of the head code (mspa.heads, mspa.scene, mspa.coverage, mspa.pipeline) through
    ``proxy`` for the duration of the block: the package attribute (what ``from . import engine`` inside a function
    resolves to) and the module-level bindings."""
    import sys
    pkg = sys.modules[__name__.rsplit(".", 1)[0]]
    mods = [pkg] + [sys.modules[m] for m in (pkg.__name__ + ".scene", pkg.__name__ + ".coverage", pkg.__name__ + ".pipeline",
                                             pkg.__name__ + ".heads") if m in sys.modules]
    saved = [(m, m.__dict__["engine"]) for m in mods if "engine" in m.__dict__]
    try:
        for m, _ in saved:
            m.engine = proxy
        yield proxy
    finally:
        for m, e in saved:
            m.engine = e


class ReplayScene:
    """What the record stage reads of a resident scene besides kernel results: ids, index, sizes, the device.  Tensors that
    only ever travel INTO kernel launches are placeholders of the right shape (no storage)."""

    def __init__(self, scene_cls, K, A, ids: Sequence[str], image_hw, n_points: int, device, count: torch.Tensor, depth_scale=0.001):
        self.K, self.A = np.asarray(K, np.float64), np.asarray(A, np.float64)
        self.ids = list(ids)
        self.index = {k: n for n, k in enumerate(self.ids)}
        self.image_hw = tuple(int(v) for v in image_hw)
        self.device, self.depth_scale = device, float(depth_scale)
        self.xyz = torch.empty((n_points, 3), dtype=torch.float64, device="meta")
        self.cam_mats = self.depth = self.frame_mats = self.rgb = None
        self._bits = torch.zeros((len(self.ids), 1), dtype=torch.int64, device=device)
        self._count = count
        self._cls = scene_cls

    def _visibility(self) -> Dict[str, torch.Tensor]:
        return {"bits": self._bits, "count": self._count}

    def pose_tables(self):
        return (None, None, None, None)

    def __getattr__(self, name):          # the scene-level pipelines built on kernel results (object_visibility, object_coverage)
        fn = getattr(self._cls, name)
        return fn.__get__(self, self._cls)
