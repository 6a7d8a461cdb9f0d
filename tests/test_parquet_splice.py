"""mspa/parquet_splice.py: a parquet file assembled from row groups encoded elsewhere reads back as the concatenation of the
tables, with pyarrow and with pandas (the readers the reference uses: pd.read_parquet, IH:452-475, CME:29-151)."""
import io

import numpy as np
import pandas as pd
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from mspa import parquet_splice as PS


def _pair_table(scene, n, seed):
    rng = np.random.default_rng(seed)
    ids = pa.array([f"{5 * k:05d}" for k in range(40)], pa.string())
    ov = rng.random(n) * 100
    ov[rng.random(n) < 0.3] = 0.0
    if n:
        ov[0] = np.nan
    return pa.table({"scene_id": pa.repeat(pa.scalar(scene, pa.string()), n),
                     "image_id1": ids.take(pa.array(rng.integers(0, 40, n), pa.int32())),
                     "image_id2": ids.take(pa.array(rng.integers(0, 40, n), pa.int32())),
                     "overlap": pa.array(ov), "distance": pa.array(rng.random(n)), "yaw": pa.array(rng.random(n) * 360 - 180),
                     "pitch": pa.array(rng.random(n) - 0.5)})


@pytest.mark.parametrize("options", [{}, {"use_dictionary": ["scene_id", "image_id1", "image_id2"]}, {"compression": "none"},
                                     {"use_dictionary": False, "compression": "zstd", "write_statistics": False},
                                     {"data_page_size": 4096, "write_page_index": True}])
def test_spliced_file_reads_back_as_the_concatenation(options, tmp_path):
    tables = [_pair_table(f"scene{k:04d}_00", n, k) for k, n in enumerate([780, 1, 0, 5000, 17, 120000])]
    minis = [PS.encode_row_group(t, **options) for t in tables]
    path = str(tmp_path / "spliced.parquet")
    with PS.SplicedParquetWriter(path) as w:
        for m in minis:
            w.append(m)
        assert w.num_rows == sum(t.num_rows for t in tables) and w.num_row_groups == 5
    want = pa.concat_tables(tables)
    got = pq.read_table(path)
    assert got.schema == want.schema and got.num_rows == want.num_rows
    for name in want.column_names:
        a, b = got.column(name).to_numpy(zero_copy_only=False), want.column(name).to_numpy(zero_copy_only=False)
        if a.dtype == np.float64:
            assert np.array_equal(a.view(np.int64), b.view(np.int64)), name          # NaN payloads included
        else:
            assert list(a) == list(b), name
    md = pq.read_metadata(path)
    assert md.num_row_groups == 5 and md.num_rows == want.num_rows
    assert [md.row_group(k).num_rows for k in range(5)] == [780, 1, 5000, 17, 120000]
    # every row group can be read on its own (offsets are right one by one, not just in sequence), statistics survive
    f = pq.ParquetFile(path)
    assert f.read_row_group(3).num_rows == 17 and f.read_row_group(4).column("image_id1")[119999] == tables[5].column("image_id1")[119999]
    if options.get("write_statistics", True):
        assert md.row_group(2).column(4).statistics.max == pq.read_metadata(io.BytesIO(minis[3])).row_group(0).column(4).statistics.max
    # pandas, and a filtered read
    df = pd.read_parquet(path)
    assert len(df) == want.num_rows and df.scene_id.iloc[-1] == "scene0005_00"
    assert len(pd.read_parquet(path, filters=[("scene_id", "==", "scene0003_00")])) == 5000
    # the same bytes whatever sink
    assert PS.splice(minis) == open(path, "rb").read()


def test_json_text_columns_and_errors(tmp_path):
    keys = [f"scene0000_00:point_to_images:{k}" for k in range(3000)]
    vals = ["[" + ", ".join(f'"{5 * j:05d}"' for j in range(k % 37)) + "]" for k in range(3000)]
    t = pa.table({"key": pa.array(keys, pa.string()), "values": pa.array(vals, pa.string())})
    minis = [PS.encode_row_group(t.slice(lo, 1000), use_dictionary=False) for lo in (0, 1000, 2000)]
    got = pq.read_table(io.BytesIO(PS.splice(minis)))
    assert got.equals(t)
    with pytest.raises(ValueError, match="schema"):
        PS.splice([minis[0], PS.encode_row_group(_pair_table("s", 3, 0))])
    with pytest.raises(ValueError, match="not a parquet file"):
        PS.splice([b"PAR1 nonsense"])
    with pytest.raises(ValueError, match="nothing was appended"):
        PS.SplicedParquetWriter(io.BytesIO()).close()
    # a file of only empty tables is a valid, empty file with the schema
    empty = pq.read_table(io.BytesIO(PS.splice([PS.encode_row_group(t.slice(0, 0))])))
    assert empty.num_rows == 0 and empty.schema == t.schema


def test_thrift_roundtrip_is_lossless_on_real_footers():
    """Decoding and re-encoding a footer pyarrow wrote gives pyarrow's bytes back: nothing in it is misread or dropped."""
    for opts in ({}, {"use_dictionary": False, "compression": "none"}, {"write_page_index": True, "data_page_size": 2048}):
        buf = PS.encode_row_group(_pair_table("scene0001_00", 4000, 7), **opts)
        mv = memoryview(buf)
        (flen,) = PS.struct.unpack("<I", mv[-8:-4])
        raw = bytes(mv[len(mv) - 8 - flen:len(mv) - 8])
        _pages, meta = PS.split_file(buf)
        w = PS._Writer()
        w.struct(meta)
        assert bytes(w.o) == raw
