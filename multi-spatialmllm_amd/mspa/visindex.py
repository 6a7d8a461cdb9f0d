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
from typing import Dict, List

import numpy as np


@dataclasses.dataclass
class VisibilityCSR:
    image_ids: List[str]
    n_points: int
    i2p_offsets: np.ndarray      # [F + 1] int64
    i2p_indices: np.ndarray      # [nnz] int32 vertex indices, ascending within an image
    p2i_offsets: np.ndarray      # [N + 1] int64
    p2i_indices: np.ndarray      # [nnz] int32 image indices (positions in image_ids), ascending within a vertex

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
        ``visibility_dict_to_frame``."""
        import pyarrow as pa
        import pyarrow.compute as pc

        def json_lists(offsets, elements):
            lists = pa.ListArray.from_arrays(pa.array(offsets.astype(np.int32)), elements)
            body = pc.binary_join(lists, ", ")
            return pc.binary_join_element_wise(pa.scalar("["), body, pa.scalar("]"), pa.scalar(""))

        if self.i2p_offsets[-1] >= 2 ** 31 or self.p2i_offsets[-1] >= 2 ** 31:
            raise ValueError("more than 2^31 entries in one scene's index")
        i2p_vals = json_lists(self.i2p_offsets, pc.cast(pa.array(self.i2p_indices), pa.string()))
        quoted = pa.array([json.dumps(i) for i in self.image_ids], type=pa.string())
        p2i_vals = json_lists(self.p2i_offsets, quoted.take(pa.array(self._p2i_sorted_ids())) if len(self.p2i_indices)
                              else pa.array([], type=pa.string()))
        i2p_keys = pa.array([f"{scene_id}:image_to_points:{i}" for i in self.image_ids], type=pa.string())
        p2i_keys = pc.binary_join_element_wise(pa.scalar(f"{scene_id}:point_to_images:"),
                                               pc.cast(pa.array(np.arange(self.n_points, dtype=np.int64)), pa.string()),
                                               pa.scalar(""))
        return pa.table({"key": pa.concat_arrays([i2p_keys, p2i_keys]), "values": pa.concat_arrays([i2p_vals, p2i_vals])})


def from_bits(bits, image_ids: List[str], n_points: int) -> VisibilityCSR:
    """K1's bitsets [F, ceil(N/64)] (device int64 tensor) -> both CSR tables, compacted on the device."""
    from . import engine
    F = len(image_ids)
    if F == 0 or n_points == 0:
        return VisibilityCSR(list(image_ids), n_points, np.zeros(F + 1, np.int64), np.zeros(0, np.int32),
                             np.zeros(n_points + 1, np.int64), np.zeros(0, np.int32))
    o1, i1 = engine.bitset_csr(bits)
    t = engine.bits_transpose(bits)                      # [n_words * 64, ceil(F / 64)]; rows >= N are padding (all zero)
    o2, i2 = engine.bitset_csr(t[:n_points].contiguous() if t.shape[0] != n_points else t)
    return VisibilityCSR(list(image_ids), n_points, o1.cpu().numpy(), i1.cpu().numpy(), o2.cpu().numpy(), i2.cpu().numpy())
