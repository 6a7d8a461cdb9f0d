"""mspa/visindex.py: the visibility index as CSR columns -> arrow table / nested dict, against the reference's formatting
(make_visibility_info.py:38-73: keys "scene:image_to_points:img" / "scene:point_to_images:idx", values json.dumps(list))."""
import json

import numpy as np
import pytest

from mspa import visindex
from spatial_engine.utils.scannet_utils.make_visibility_info import visibility_dict_to_frame


def csr_from_mask(mask, ids):
    F, N = mask.shape
    i2p = [np.nonzero(mask[k])[0].astype(np.int32) for k in range(F)]
    p2i = [np.nonzero(mask[:, v])[0].astype(np.int32) for v in range(N)]
    off = lambda lists: np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    cat = lambda lists: np.concatenate(lists).astype(np.int32) if lists and sum(len(x) for x in lists) else np.zeros(0, np.int32)
    return visindex.VisibilityCSR(list(ids), N, off(i2p), cat(i2p), off(p2i), cat(p2i))


@pytest.mark.parametrize("ids", [["00000", "00005", "00010", "00015", "00020"], ["7", "10", "9", "100", "8"]],
                         ids=["sorted", "unsorted"])
def test_csr_to_dict_and_arrow_match_reference_formatting(ids):
    rng = np.random.default_rng(3)
    mask = rng.random((len(ids), 333)) < 0.2
    mask[2] = False                                   # an image that sees nothing
    mask[:, 17] = False                               # a vertex nobody sees
    csr = csr_from_mask(mask, ids)
    d = csr.to_dict()
    for k, img in enumerate(ids):
        assert d["image_to_points"][img] == np.nonzero(mask[k])[0].tolist()
        assert all(type(v) is int for v in d["image_to_points"][img])
    for v in range(mask.shape[1]):
        assert d["point_to_images"][v] == sorted(ids[k] for k in np.nonzero(mask[:, v])[0])      # MVI:117
    assert list(d["point_to_images"].keys()) == list(range(mask.shape[1])) and d["point_to_images"][17] == []
    assert csr.empty_images() == [ids[2]]
    ref = visibility_dict_to_frame({"scene0001_00": d})
    got = csr.to_arrow("scene0001_00").to_pandas()
    assert list(got.columns) == ["key", "values"] and len(got) == len(ref)
    assert got["key"].tolist() == ref["key"].tolist()
    assert got["values"].tolist() == ref["values"].tolist()
    assert json.loads(got["values"][0]) == d["image_to_points"][ids[0]]


def test_empty_scene_index():
    csr = visindex.from_bits(None, [], 5)
    assert csr.to_dict() == {"image_to_points": {}, "point_to_images": {v: [] for v in range(5)}}
    t = csr.to_arrow("s").to_pandas()
    assert t["values"].tolist() == ["[]"] * 5 and t["key"][0] == "s:point_to_images:0"


def test_scene_row_groups_reads_only_what_a_scene_needs(tmp_path):
    """visindex.SceneRowGroups on the two layouts of the index file: one row group per scene (what make_visibility_info.run_split
    streams) and pandas' single big row group spanning all scenes; an unknown scene is an empty dict."""
    import pandas as pd
    import pyarrow as pa
    import pyarrow.parquet as pq
    from mspa import visindex
    rows = {}
    for s in range(5):
        sid = f"scene{s:04d}_00"
        rows[sid] = {f"{sid}:image_to_points:{5 * k:05d}": json.dumps(list(range(s, s + k))) for k in range(4)}
        rows[sid].update({f"{sid}:point_to_images:{v}": json.dumps([f"{5 * k:05d}" for k in range(v % 3)]) for v in range(6)})
    per_scene = str(tmp_path / "per_scene.parquet")
    with pq.ParquetWriter(per_scene, pa.schema([("key", pa.string()), ("values", pa.string())])) as w:
        for sid in rows:
            w.write_table(pa.table({"key": list(rows[sid]), "values": list(rows[sid].values())}))
    one_group = str(tmp_path / "one_group.parquet")
    pd.DataFrame({"key": [k for d in rows.values() for k in d], "values": [v for d in rows.values() for v in d.values()]}).to_parquet(one_group)
    for path, groups_of_scene in ((per_scene, 1), (one_group, 0)):
        idx = visindex.SceneRowGroups(path)
        assert all(len(idx._by_scene.get(sid, [])) == groups_of_scene for sid in rows)
        for sid in rows:
            assert idx.scene_dict(sid) == rows[sid]
        assert idx.scene_dict("scene9999_00") == {}
