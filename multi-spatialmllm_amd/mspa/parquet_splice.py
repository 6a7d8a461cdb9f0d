"""One parquet file out of row groups that were ENCODED ELSEWHERE -- rank 0 of a sharded sweep only copies bytes.

The reference's pair table and visibility index are single parquet files (CFR:28-57, MVI:38-73: ``df.to_parquet``).  In a
sharded sweep the rows of every scene used to travel to rank 0, which alone dictionary-encoded, compressed and wrote them:
7 ms per 320-frame scene for the pair table, 54 ms for the visibility index's JSON text -- a serial stage that capped the
index sweep at 1.6 x on 4 ranks (profiles/r06_dropin_ranks.md).  Encoding is the expensive part and it is embarrassingly
parallel, so the OWNER rank now encodes each of its scenes into a self-contained one-row-group parquet file in memory
(``encode_row_group``: pyarrow's own writer, nothing re-implemented there) and rank 0 splices these (``SplicedParquetWriter``):

  * a parquet file is ``PAR1`` + the row groups' page bytes + a thrift-compact ``FileMetaData`` footer + its length + ``PAR1``;
    the page bytes of a row group are position independent, only the footer holds absolute file offsets;
  * ``append`` copies a mini-file's page bytes [4, footer) to the output and keeps its ``RowGroup`` metadata with every offset
    (``RowGroup.file_offset``, ``ColumnChunk.file_offset``, the data / index / dictionary page offsets, page-index and bloom
    filter offsets if a writer ever emits them) moved by the distance the bytes moved, and the ordinal renumbered;
  * ``close`` writes ONE footer: the first mini-file's schema, key-value metadata, creator and column orders, the row counts
    summed, all row groups.

The thrift codec below is generic (a struct is a list of ``(field id, wire type, value)``), so fields this module does not know
about are carried through unchanged.  One code path for every world size: a single process encodes and splices too, hence the
files of 1 and N ranks are identical byte for byte.
"""
from __future__ import annotations

import io
import struct
from typing import BinaryIO, List, Optional, Tuple

MAGIC = b"PAR1"

# thrift compact wire types
T_STOP, T_TRUE, T_FALSE, T_BYTE, T_I16, T_I32, T_I64, T_DOUBLE, T_BINARY, T_LIST, T_SET, T_MAP, T_STRUCT = range(13)


class _Reader:
    def __init__(self, buf: bytes, pos: int = 0):
        self.b, self.p = memoryview(buf), pos

    def byte(self) -> int:
        v = self.b[self.p]
        self.p += 1
        return v

    def varint(self) -> int:
        out = shift = 0
        while True:
            c = self.byte()
            out |= (c & 0x7F) << shift
            if not c & 0x80:
                return out
            shift += 7

    def zigzag(self) -> int:
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def value(self, t: int):
        if t in (T_TRUE, T_FALSE):
            return t == T_TRUE
        if t == T_BYTE:
            v = self.byte()
            return v - 256 if v > 127 else v
        if t in (T_I16, T_I32, T_I64):
            return self.zigzag()
        if t == T_DOUBLE:
            v = bytes(self.b[self.p:self.p + 8])
            self.p += 8
            return v
        if t == T_BINARY:
            n = self.varint()
            v = bytes(self.b[self.p:self.p + n])
            self.p += n
            return v
        if t in (T_LIST, T_SET):
            h = self.byte()
            n, et = h >> 4, h & 0x0F
            if n == 15:
                n = self.varint()
            if et in (T_TRUE, T_FALSE):                    # list<bool>: one byte per element
                return et, [self.byte() for _ in range(n)]
            return et, [self.value(et) for _ in range(n)]
        if t == T_MAP:
            n = self.varint()
            if n == 0:
                return 0, 0, []
            h = self.byte()
            kt, vt = h >> 4, h & 0x0F
            return kt, vt, [(self.value(kt), self.value(vt)) for _ in range(n)]
        if t == T_STRUCT:
            return self.struct()
        raise ValueError(f"parquet footer: unknown thrift type {t}")

    def struct(self) -> List[Tuple[int, int, object]]:
        fields, last = [], 0
        while True:
            h = self.byte()
            if h == T_STOP:
                return fields
            t, delta = h & 0x0F, h >> 4
            fid = last + delta if delta else self.zigzag()
            last = fid
            fields.append((fid, t, self.value(t)))


class _Writer:
    def __init__(self):
        self.o = bytearray()

    def varint(self, v: int):
        while True:
            c = v & 0x7F
            v >>= 7
            if v:
                self.o.append(c | 0x80)
            else:
                self.o.append(c)
                return

    def zigzag(self, v: int):
        self.varint((v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1)

    def value(self, t: int, v):
        if t in (T_TRUE, T_FALSE):
            return                                          # carried by the field header
        if t == T_BYTE:
            self.o.append(v & 0xFF)
        elif t in (T_I16, T_I32, T_I64):
            self.zigzag(v)
        elif t == T_DOUBLE:
            self.o += v
        elif t == T_BINARY:
            self.varint(len(v))
            self.o += v
        elif t in (T_LIST, T_SET):
            et, items = v
            if len(items) < 15:
                self.o.append((len(items) << 4) | et)
            else:
                self.o.append(0xF0 | et)
                self.varint(len(items))
            for x in items:
                if et in (T_TRUE, T_FALSE):
                    self.o.append(x)
                else:
                    self.value(et, x)
        elif t == T_MAP:
            kt, vt, items = v
            self.varint(len(items))
            if items:
                self.o.append((kt << 4) | vt)
                for k, x in items:
                    self.value(kt, k)
                    self.value(vt, x)
        elif t == T_STRUCT:
            self.struct(v)
        else:
            raise ValueError(f"parquet footer: unknown thrift type {t}")

    def struct(self, fields):
        last = 0
        for fid, t, v in fields:
            if t in (T_TRUE, T_FALSE):
                t = T_TRUE if v else T_FALSE
            delta = fid - last
            if 0 < delta <= 15:
                self.o.append((delta << 4) | t)
            else:
                self.o.append(t)
                self.zigzag(fid)
            self.value(t, v)
            last = fid
        self.o.append(T_STOP)


def _get(fields, fid, default=None):
    for f, _t, v in fields:
        if f == fid:
            return v
    return default


def _shift(fields, fid, delta):
    """Moves an absolute file offset; an absent field, or one left at 0 (= unset: nothing lives at offset 0 but the magic) stays."""
    for k, (f, t, v) in enumerate(fields):
        if f == fid and v:
            fields[k] = (f, t, v + delta)


def _set(fields, fid, t, value):
    for k, (f, _t, _v) in enumerate(fields):
        if f == fid:
            fields[k] = (fid, t, value)
            return
    fields.append((fid, t, value))
    fields.sort(key=lambda x: x[0])


def split_file(buf) -> Tuple[memoryview, List]:
    """(page bytes of all row groups, decoded FileMetaData) of an in-memory parquet file."""
    mv = memoryview(buf).cast("B")
    if len(mv) < 12 or bytes(mv[:4]) != MAGIC or bytes(mv[-4:]) != MAGIC:
        raise ValueError("parquet_splice: not a parquet file")
    (flen,) = struct.unpack("<I", mv[-8:-4])
    fstart = len(mv) - 8 - flen
    if fstart < 4:
        raise ValueError("parquet_splice: corrupt footer length")
    meta = _Reader(mv, fstart).struct()
    return mv[4:fstart], meta


def encode_row_group(table, **writer_options) -> bytes:
    """``table`` as a self-contained parquet file with ONE row group (pyarrow's own writer: dictionary pages, compression,
    statistics are whatever ``writer_options`` say); a table without rows gives a file without row groups."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    sink = pa.BufferOutputStream()
    with pq.ParquetWriter(sink, table.schema, **writer_options) as w:
        if table.num_rows:
            w.write_table(table, row_group_size=table.num_rows)
    return sink.getvalue()


class SplicedParquetWriter:
    """``append(mini_file_bytes)`` in the order the row groups are to appear; ``close()`` writes the footer.  ``num_rows`` and
    ``num_row_groups`` are kept up to date.  Every mini-file must carry the same schema (checked)."""

    def __init__(self, path_or_file):
        self._own = isinstance(path_or_file, (str, bytes))
        self.f: BinaryIO = open(path_or_file, "wb") if self._own else path_or_file
        self.f.write(MAGIC)
        self.pos = 4
        self.head: Optional[List] = None            # FileMetaData of the first mini-file (schema, kv metadata, creator, orders)
        self.row_groups: List = []
        self.num_rows = 0
        self.closed = False

    @property
    def num_row_groups(self) -> int:
        return len(self.row_groups)

    def append(self, buf) -> int:
        """Returns the number of rows added."""
        pages, meta = split_file(buf)
        if self.head is None:
            self.head = meta
        elif _get(meta, 2) != _get(self.head, 2):
            raise ValueError("parquet_splice: the row groups do not share one schema")
        _et, groups = _get(meta, 4, (T_STRUCT, []))
        delta = self.pos - 4
        rows = 0
        for rg in groups:
            _shift(rg, 5, delta)                                        # RowGroup.file_offset
            _ct, cols = _get(rg, 1)
            for cc in cols:
                for fid in (2, 4, 6):                                   # ColumnChunk.file_offset, offset / column index offsets
                    _shift(cc, fid, delta)
                md = _get(cc, 3)
                if md is not None:
                    for fid in (9, 10, 11, 14):                         # data / index / dictionary page, bloom filter offsets
                        _shift(md, fid, delta)
            if _get(rg, 7) is not None:
                _set(rg, 7, T_I16, len(self.row_groups))                # RowGroup.ordinal
            rows += _get(rg, 3, 0)
            self.row_groups.append(rg)
        if len(pages):
            self.f.write(pages)
            self.pos += len(pages)
        self.num_rows += rows
        return rows

    def close(self):
        if self.closed:
            return
        self.closed = True
        try:
            if self.head is None:
                raise ValueError("parquet_splice: nothing was appended, there is no schema to write")
            meta = list(self.head)
            _set(meta, 3, T_I64, self.num_rows)
            _set(meta, 4, T_LIST, (T_STRUCT, self.row_groups))
            w = _Writer()
            w.struct(meta)
            self.f.write(bytes(w.o))
            self.f.write(struct.pack("<I", len(w.o)))
            self.f.write(MAGIC)
        finally:
            if self._own:
                self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
        elif self._own and not self.closed:
            self.closed = True
            self.f.close()


def splice(bufs, path_or_file=None) -> Optional[bytes]:
    """Convenience: the spliced file of ``bufs`` written to ``path_or_file``, or returned as bytes."""
    sink = io.BytesIO() if path_or_file is None else path_or_file
    w = SplicedParquetWriter(sink)
    for b in bufs:
        w.append(b)
    w.close()
    return sink.getvalue() if path_or_file is None else None
