"""The visibility index of a scene (MVI.process_scene, make_visibility_info.py:75-125) as columns.

``image_to_points`` (per image: ascending indices of the vertices it sees) and ``point_to_images`` (per vertex: the
images that see it, in image order) are two compactions of K1's bit matrix -- the second one of its transpose -- done
on the device (K9: mspa_bits_popcount / mspa_bits_expand / mspa_bits_transpose).  What comes back to the host is a
pair of CSR tables; from there

  * ``to_arrow``  builds the (key, values) table every reader of the index expects -- keys
    ``scene:image_to_points:img`` / ``scene:point_to_images:idx``, values the JSON text of the list, byte for byte what
    ``json.dumps`` gives (make_visibility_info.py:38-73) -- with arrow compute kernels, no Python object per row;
  * ``to_dict``   builds the reference's nested dict (for the .pkl output and for callers that want the original type).
"""
from __future__ import annotations

import dataclasses
import json
from typing import Dict, List, Optional

import numpy as np


@dataclasses.dataclass
class VisibilityCSR:
    image_ids: List[str]
    n_points: int
    i2p_offsets: np.ndarray      # [F + 1] int64
    i2p_indices: np.ndarray      # [nnz] int32 vertex indices, ascending within an image
    p2i_offsets: np.ndarray      # [N + 1] int64
    p2i_indices: np.ndarray      # [nnz] int32 image indices (positions in image_ids), ascending within a vertex
    # the lists' JSON text already written on the device (K10, engine.format_lists_device), as arrow string buffers
    # (int32 offsets [rows + 1], uint8 data): what ``to_arrow`` uses instead of the host formatters when present
    i2p_text: Optional[tuple] = None
    p2i_text: Optional[tuple] = None

    def empty_images(self) -> List[str]:
        n = np.diff(self.i2p_offsets)
        return [self.image_ids[k] for k in np.nonzero(n == 0)[0]]

    def _p2i_sorted_ids(self):
        """Image ids per entry of p2i, each vertex's list in sorted() order (MVI:117).  Image order == sorted order whenever
        the ids are sorted (zero-padded frame numbers: always, in ScanNet exports); otherwise re-rank per vertex."""
        ids = self.image_ids
        if ids == sorted(ids):
            return self.p2i_indices
        rank = np.argsort(np.argsort(np.array(ids, dtype=object), kind="stable"), kind="stable")   # image -> rank in sorted()
        r = rank[self.p2i_indices]
        row = np.repeat(np.arange(self.n_points), np.diff(self.p2i_offsets))
        order = np.lexsort((r, row))
        return self.p2i_indices[order]

    def to_dict(self) -> Dict[str, dict]:
        """The reference's nested dict.  131 k small lists and 0.6 M references per scene are created here; the cyclic
        collector is paused meanwhile (none of these objects can form a cycle, and its generation-0 passes otherwise take
        more than half of the time)."""
        import gc
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return self._to_dict()
        finally:
            if was_enabled:
                gc.enable()

    def _to_dict(self) -> Dict[str, dict]:
        off = self.i2p_offsets.tolist()
        flat = self.i2p_indices.tolist()
        image_to_points = {img: flat[off[k]:off[k + 1]] for k, img in enumerate(self.image_ids)}
        ids_obj = np.array(self.image_ids, dtype=object)
        pflat = ids_obj[self._p2i_sorted_ids()].tolist() if len(self.p2i_indices) else []
        poff = self.p2i_offsets.tolist()
        point_to_images = {v: pflat[poff[v]:poff[v + 1]] for v in range(self.n_points)}
        return {"image_to_points": image_to_points, "point_to_images": point_to_images}

    def to_arrow(self, scene_id: str):
        """pyarrow table (key: string, values: string), image_to_points rows first -- the order and the text of
        ``visibility_dict_to_frame``.  The JSON text is written by libmspa's host-side formatters straight into arrow's
        (offsets, data) buffers: no Python object per row, no intermediate string arrays."""
        import pyarrow as pa
        from . import _lib
        lib = _lib.load()

        def string_array(n, text, offsets, nbytes):
            if nbytes < 0:
                _lib.check(int(nbytes))
            return pa.StringArray.from_buffers(n, pa.py_buffer(offsets), pa.py_buffer(text[:nbytes]))

        def ptr(a):
            return a.ctypes.data if a.size else None

        F, N = len(self.image_ids), self.n_points

        def preformatted(n, pair):
            offs, data = pair
            return pa.StringArray.from_buffers(n, pa.py_buffer(offs), pa.py_buffer(data)) if n else pa.array([], type=pa.string())

        if self.i2p_text is not None and self.p2i_text is not None:
            i2p_vals, p2i_vals = preformatted(F, self.i2p_text), preformatted(N, self.p2i_text)
        else:
            i2p_vals, p2i_vals = self._format_on_host(lib, string_array, ptr, pa)
        # keys
        i2p_keys = pa.array([f"{scene_id}:image_to_points:{i}" for i in self.image_ids], type=pa.string())
        prefix = f"{scene_id}:point_to_images:".encode()
        cap = (len(prefix) + 21) * N + 16
        text, offs = np.empty(cap, dtype=np.uint8), np.empty(N + 1, dtype=np.int32)
        nb = lib.mspa_format_int_keys_host(prefix, 0, N, text.ctypes.data, cap, offs.ctypes.data) if N else 0
        p2i_keys = string_array(N, text, offs, nb) if N else pa.array([], type=pa.string())
        return pa.table({"key": pa.concat_arrays([i2p_keys, p2i_keys]), "values": pa.concat_arrays([i2p_vals, p2i_vals])})

    def quoted_image_ids(self) -> List[bytes]:
        return [json.dumps(i).encode() for i in self.image_ids]

    def _format_on_host(self, lib, string_array, ptr, pa):
        """Both value columns by libmspa's host formatters (sequential loops: ~40 ms of one core per 320-frame scene)."""
        F, N = len(self.image_ids), self.n_points
        i2p_off = np.ascontiguousarray(self.i2p_offsets, dtype=np.int64)
        i2p_idx = np.ascontiguousarray(self.i2p_indices, dtype=np.int32)
        p2i_off = np.ascontiguousarray(self.p2i_offsets, dtype=np.int64)
        p2i_idx = np.ascontiguousarray(self._p2i_sorted_ids(), dtype=np.int32)
        # values of image_to_points: integer lists
        cap = 2 * F + 13 * len(i2p_idx) + 16
        text, offs = np.empty(cap, dtype=np.uint8), np.empty(F + 1, dtype=np.int32)
        nb = lib.mspa_format_int_lists_host(ptr(i2p_off), ptr(i2p_idx), F, text.ctypes.data, cap, offs.ctypes.data) if F else 0
        i2p_vals = string_array(F, text, offs, nb) if F else pa.array([], type=pa.string())
        # values of point_to_images: lists of quoted image ids
        quoted = self.quoted_image_ids()
        tok_off = np.concatenate([[0], np.cumsum([len(q) for q in quoted])]).astype(np.int32)
        tokens = np.frombuffer(b"".join(quoted) or b"\0", dtype=np.uint8)
        longest = max([len(q) for q in quoted], default=0)
        cap = 2 * N + (longest + 2) * len(p2i_idx) + 16
        text, offs = np.empty(cap, dtype=np.uint8), np.empty(N + 1, dtype=np.int32)
        nb = lib.mspa_format_token_lists_host(ptr(p2i_off), ptr(p2i_idx), N, tokens.ctypes.data, tok_off.ctypes.data, F,
                                              text.ctypes.data, cap, offs.ctypes.data) if N else 0
        p2i_vals = string_array(N, text, offs, nb) if N else pa.array([], type=pa.string())
        return i2p_vals, p2i_vals


def from_bits(bits, image_ids: List[str], n_points: int, text: bool = False, indices: bool = True) -> VisibilityCSR:
    """K1's bitsets [F, ceil(N/64)] (device int64 tensor) -> both CSR tables, compacted on the device.
    ``text``: also write both lists' JSON text on the device (K10) and bring it along as arrow string buffers -- what
    ``to_arrow`` then uses (only when the image ids are in sorted() order, MVI:117: always, for ScanNet's zero-padded frame
    numbers).  ``indices=False``: the index arrays themselves stay on the device (a parquet sweep that keeps nothing needs only
    the text: 83 MB instead of 83 + 39)."""
    from . import engine
    F = len(image_ids)
    if F == 0 or n_points == 0:
        return VisibilityCSR(list(image_ids), n_points, np.zeros(F + 1, np.int64), np.zeros(0, np.int32),
                             np.zeros(n_points + 1, np.int64), np.zeros(0, np.int32))
    import torch
    o1, i1 = engine.bitset_csr(bits)
    t = engine.bits_transpose(bits)                      # [n_words * 64, ceil(F / 64)]; rows >= N are padding (all zero)
    o2, i2 = engine.bitset_csr(t[:n_points].contiguous() if t.shape[0] != n_points else t)
    ids = list(image_ids)
    text = text and ids == sorted(ids)
    want = [o1, i1 if indices or not text else None, o2, i2 if indices or not text else None]
    if text:
        csr0 = VisibilityCSR(ids, n_points, None, None, None, None)
        want += list(engine.format_lists_device(o1, i1)[::-1]) + list(engine.format_lists_device(o2, i2, csr0.quoted_image_ids())[::-1])
    # ~100 MB per 320-frame scene: into pinned blocks (torch's caching host allocator), all copies behind ONE wait on the
    # CURRENT stream -- the sweeps call this on an encoder thread with a stream of its own, next to the sweep thread's kernels
    host = []
    for a in want:
        if a is None:
            host.append(None)
            continue
        h = torch.empty(a.shape, dtype=a.dtype, pin_memory=True)
        h.copy_(a, non_blocking=True)
        host.append(h)
    torch.cuda.current_stream(bits.device).synchronize()
    host = [None if h is None else h.numpy() for h in host]
    csr = VisibilityCSR(ids, n_points, *host[:4])
    if text:
        csr.i2p_text, csr.p2i_text = (host[4], host[5]), (host[6], host[7])
    return csr


class SceneRowGroups:
    """Scene-addressable reader of the visibility-index parquet (columns ``key`` = "scene:kind:item", ``values`` = JSON text;
    MVI:38-73): which row groups hold which scene is read off the ``key`` column's min / max statistics in the footer, so a rank
    of a sharded job loads the rows of ITS scenes only -- the reference loads the whole table (13 GB for the train split) into a
    dict in every process (COVIS:60-70, IH:486-500).  A row group whose keys span several scenes (files written with pandas'
    default 1 M-row groups) is read once and kept.  ``scene_dict(scene_id)`` -> {key: JSON text} of that scene."""

    def __init__(self, parquet_file: str):
        import pyarrow.parquet as pq
        self.file = pq.ParquetFile(parquet_file)
        md = self.file.metadata
        key_col = self.file.schema_arrow.get_field_index("key")
        self._by_scene: Dict[str, List[int]] = {}
        self._mixed: List[tuple] = []                       # (row group, first scene, last scene)
        self._cache: Dict[int, Dict[str, Dict[str, str]]] = {}
        for g in range(md.num_row_groups):
            st = md.row_group(g).column(key_col).statistics
            if st is None or not st.has_min_max:
                self._mixed.append((g, "", "\U0010ffff"))
                continue
            lo, hi = str(st.min).split(":", 1)[0], str(st.max).split(":", 1)[0]
            if lo == hi:
                self._by_scene.setdefault(lo, []).append(g)
            else:
                self._mixed.append((g, lo, hi))

    def _split_group(self, g: int) -> Dict[str, Dict[str, str]]:
        if g not in self._cache:
            t = self.file.read_row_group(g, columns=["key", "values"])
            per: Dict[str, Dict[str, str]] = {}
            for k, v in zip(t.column("key").to_pylist(), t.column("values").to_pylist()):
                per.setdefault(k.split(":", 1)[0], {})[k] = v
            self._cache[g] = per
        return self._cache[g]

    def scene_dict(self, scene_id: str) -> Dict[str, str]:
        out: Dict[str, str] = {}
        for g in self._by_scene.get(scene_id, []):
            t = self.file.read_row_group(g, columns=["key", "values"])
            out.update(zip(t.column("key").to_pylist(), t.column("values").to_pylist()))
        for g, lo, hi in self._mixed:
            if lo <= scene_id <= hi:
                out.update(self._split_group(g).get(scene_id, {}))
        return out
