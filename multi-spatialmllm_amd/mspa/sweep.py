"""Sharded, prefetched sweeps over on-disk scenes: what runs behind the drop-in entry points.

The reference fans scenes out over a ``multiprocessing.Pool`` (CFR:222-229 ``num_workers=25``, MVI:151-156,
OM_C:584-585) because every scene is seconds of NumPy on one core.  Here a scene is ~0.2 ms of kernels, so the fan-out is
reshaped around what is left:

  * **one process per GPU** (``shard.context_from_env``): the scenes of a split are cut into consecutive *windows* of
    ``world x per_rank`` scenes, each window dealt to the ranks longest-processing-time-first (``shard.lpt_assign``, cost
    ~ F^2 N / 64 + F N).  No collective on the data path; after each window ONE exchange towards rank 0 -- the numeric rows
    through ``shard.collate_records`` (RCCL gather of float64 records), finished text / arrow buffers through
    ``shard.gather_bytes`` -- and rank 0 consumes the window's scenes in the split's own order, so what it writes is byte for
    byte what a single process writes (windows bound what rank 0 has to hold: a few scenes per rank, not the split);
  * **``num_workers`` = host decode threads** feeding the GPU: scene n+1's depth PNGs are read and inflated by native
    threads (``mspa.ingest``) and its vertices loaded while scene n's H2D copy runs on the copy stream and scene n-1's
    kernels on the compute stream (``SceneLoader`` -> ``upload.ScenePrefetcher``).

torch is plumbing here (streams, collectives); nothing is computed in this module.
"""
from __future__ import annotations

import collections
import contextlib
import dataclasses
import struct
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import shard


@dataclasses.dataclass
class HostScene:
    """One scene in host memory, in the form ``upload.UploadSlot.stage_and_upload`` consumes."""
    scene_id: str
    K: np.ndarray
    A: np.ndarray
    E: Dict[str, np.ndarray]
    depth: Dict[str, np.ndarray]           # valid frames only; views of one [F, h, w] block
    color_hw: Tuple[int, int]
    points: Optional[np.ndarray] = None    # [N, >=3] float64
    depth_scale: float = 0.001
    load_s: float = 0.0                    # wall time the loader spent on this scene (decode + np.load)
    packed: object = None                  # ingest.PackedDepth: the frames still compressed (decoded on the device); then
    #                                        ``depth`` is empty and ``depth_ids`` names the frames ``packed`` holds, in order
    depth_ids: Optional[List[str]] = None
    prepared: Optional[dict] = None        # upload.prepare_tables(K, A, E, points): computed on the loader's thread, so that the one
    #                                        staging thread only copies it (absent: the staging thread computes it itself)


class Timings:
    """Wall-clock seconds per stage of a sweep, summed over scenes.  The stages overlap (that is the point), so the sum
    exceeds the sweep's own wall time; ``as_dict`` reports both."""

    def __init__(self):
        self._lock = threading.Lock()
        self.s: Dict[str, float] = collections.defaultdict(float)
        self.n: Dict[str, int] = collections.defaultdict(int)

    def add(self, stage: str, seconds: float, count: int = 1):
        with self._lock:
            self.s[stage] += seconds
            self.n[stage] += count

    class _Span:
        def __init__(self, owner, stage):
            self.owner, self.stage = owner, stage

        def __enter__(self):
            self.t = time.perf_counter()

        def __exit__(self, *exc):
            self.owner.add(self.stage, time.perf_counter() - self.t)

    def span(self, stage: str):
        return Timings._Span(self, stage)

    def as_dict(self) -> Dict[str, float]:
        with self._lock:
            return {k: round(v, 6) for k, v in sorted(self.s.items())}


class SceneLoader:
    """``for host_scene in SceneLoader(load, keys)``: ``load(key)`` runs on worker threads, up to ``lookahead`` scenes ahead of
    the consumer, results in key order.  ``load`` spends its time in native code that holds no interpreter lock (PNG
    inflate, np.load, LAPACK), so two scenes in flight keep ``2 x num_workers`` host cores busy."""

    def __init__(self, load: Callable, keys: Iterable, lookahead: int = 2, timings: Optional[Timings] = None):
        self.load, self.keys, self.lookahead, self.timings = load, keys, max(1, int(lookahead)), timings

    def _timed(self, key):
        t = time.perf_counter()
        out = self.load(key)
        dt = time.perf_counter() - t
        if self.timings is not None:
            self.timings.add("decode", dt)
        if hasattr(out, "load_s"):
            out.load_s = dt
        return out

    def __iter__(self):
        it = iter(self.keys)
        pending: collections.deque = collections.deque()
        ex = ThreadPoolExecutor(max_workers=self.lookahead, thread_name_prefix="mspa-load")
        try:
            for key in it:
                pending.append(ex.submit(self._timed, key))
                if len(pending) >= self.lookahead:
                    break
            while pending:
                fut = pending.popleft()
                nxt = next(it, _END)
                if nxt is not _END:
                    pending.append(ex.submit(self._timed, nxt))
                yield fut.result()
        finally:
            for fut in pending:
                fut.cancel()
            ex.shutdown(wait=True)


_END = object()


def windows(costs: Sequence[float], world: int, per_rank: int = 8, rank0_share: float = 1.0) -> List[List[List[int]]]:
    """Consecutive windows of ``world * per_rank`` items, each dealt longest-first: ``result[w][rank]`` = ascending indices.
    ``rank0_share`` < 1 gives rank 0 (the writer) that fraction of an even share of each window's cost."""
    n, size = len(costs), max(1, world * max(1, int(per_rank)))
    out = []
    for lo in range(0, n, size):
        hi = min(n, lo + size)
        bins = shard.lpt_assign(list(costs[lo:hi]), world, rank0_share=rank0_share)
        out.append([[lo + i for i in b] for b in bins])
    return out


# ---- framing of a window's blobs: [n_items] then per item (index, n_blobs, len_0 .. len_k) + the bytes --------------------
def _pack_blobs(items: List[Tuple[int, Sequence]]) -> bytes:
    head = [struct.pack("<q", len(items))]
    body = []
    for index, blobs in items:
        blobs = [memoryview(b).cast("B") if not isinstance(b, (bytes, bytearray)) else b for b in blobs]
        head.append(struct.pack(f"<qq{len(blobs)}q", index, len(blobs), *[len(b) for b in blobs]))
        body.extend(blobs)
    return b"".join(head + body)


def _unpack_blobs(buf: np.ndarray) -> Dict[int, List[np.ndarray]]:
    if buf.size == 0:
        return {}
    mv = memoryview(buf)                                      # headers through struct; the payloads stay views of `buf`
    (n,) = struct.unpack_from("<q", mv, 0)
    pos, table = 8, []
    for _ in range(n):
        index, k = struct.unpack_from("<qq", mv, pos)
        lens = struct.unpack_from(f"<{k}q", mv, pos + 16)
        table.append((index, lens))
        pos += 16 + 8 * k
    out: Dict[int, List[np.ndarray]] = {}
    for index, lens in table:
        parts = []
        for ln in lens:
            parts.append(buf[pos:pos + ln])
            pos += ln
        out[index] = parts
    return out


class _WindowWriter:
    """Rank 0's ordered writer: ``consume`` of window w runs on this thread while window w + 1 is produced and exchanged.

    The reference hands worker results back asynchronously (``apply_async`` + ``r.get()``, CFR:222-229, MVI:151-156); here
    the other ranks would otherwise stand in the next window's first collective until rank 0 had converted and written
    every rank's scenes.  At most ``depth`` exchanged windows wait in the queue (rank 0 holds windows, never the split); a
    failure in ``consume`` is kept and raised on the sweep's thread -- at the next window's failure vote, so that every
    rank leaves together -- and the thread drains what is queued without touching it."""

    def __init__(self, consume: Callable, record_width: Optional[int], timings: "Timings", depth: int = 2):
        import queue
        self.consume, self.record_width, self.timings = consume, record_width, timings
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, int(depth)))
        self.error: Optional[BaseException] = None
        self.thread = threading.Thread(target=self._run, name="mspa-writer", daemon=True)
        self.thread.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            if self.error is not None:
                continue
            indices, rows_by_index, parts = job
            t = time.perf_counter()
            try:
                blobs_by_index: Dict[int, List[np.ndarray]] = {}
                for p in parts:
                    blobs_by_index.update(_unpack_blobs(p))
                for index in indices:
                    rows = rows_by_index.get(index)
                    if self.record_width is not None and rows is None:
                        rows = np.zeros((0, self.record_width))
                    self.consume(index, rows, blobs_by_index[index])
            except BaseException as e:                     # noqa: BLE001 -- re-raised on the sweep's thread
                self.error = e
            finally:
                self.timings.add("consume", time.perf_counter() - t)

    def put(self, indices, rows_by_index, parts):
        t = time.perf_counter()
        self.q.put((indices, rows_by_index, parts))
        self.timings.add("writer_backpressure", time.perf_counter() - t)

    def close(self) -> Optional[BaseException]:
        """Waits until everything queued has been written (or skipped after a failure); returns the failure, if any."""
        t = time.perf_counter()
        self.q.put(None)
        self.thread.join()
        self.timings.add("writer_drain", time.perf_counter() - t)
        return self.error


from .hostinfo import quiet_collector, quietly  # noqa: E402,F401  (kept importable from here: the sweeps are where it was found)


def sharded_sweep(costs: Sequence[float], ctx: Optional[shard.DistContext], work_items: Callable[[List[int]], Iterator],
                  produce: Callable, consume: Callable, record_width: Optional[int] = None, per_rank: Optional[int] = None,
                  timings: Optional[Timings] = None, writer_depth: int = 2, rank0_share: Optional[float] = None) -> None:
    """``_sharded_sweep`` (below: the arguments are its) inside ``quiet_collector``."""
    with quiet_collector():
        _sharded_sweep(costs, ctx, work_items, produce, consume, record_width, per_rank, timings, writer_depth, rank0_share)


def _sharded_sweep(costs: Sequence[float], ctx: Optional[shard.DistContext], work_items: Callable[[List[int]], Iterator],
                   produce: Callable, consume: Callable, record_width: Optional[int] = None, per_rank: Optional[int] = None,
                   timings: Optional[Timings] = None, writer_depth: int = 2, rank0_share: Optional[float] = None) -> None:
    """Run ``produce(index, item) -> (records | None, [blob, ...])`` for this rank's items and ``consume(index, records,
    blobs)`` on rank 0 for EVERY item, in index order.

    ``work_items(indices)`` yields this rank's items (prefetched however it likes) in the order of ``indices``.
    ``records`` is a [n, record_width] float64 tensor (any device) or None; blobs are bytes-like.  With a communicator the
    records of a window go to rank 0 through ``shard.collate_records(dst=0)`` and the blobs through ``shard.gather_bytes``;
    without one (``ctx`` None) nothing is exchanged and the same ``consume`` calls happen in the same order.

    A window is ``world x per_rank`` items (default: ``MSPA_WINDOW_PER_RANK`` from the environment, else 8).
    A blob -- or the whole list of an item's blobs -- may be a CALLABLE returning the bytes (the list): it runs on a small encoder pool (``MSPA_ENCODE_THREADS``, default: half of the CPUs this rank may use, 2 .. 8) while the
    sweep thread goes on to the next item, and is waited for at the window's exchange -- formatting and compressing a scene's text
    (63 ms for a visibility index) then overlaps the next scenes' kernels instead of standing between them.
    ``consume`` runs on a writer thread of rank 0 (``_WindowWriter``: window w is written while window w + 1 is produced),
    still strictly in index order, so the files stay those of a one-process run.  ``rank0_share`` (default: the environment's
    ``MSPA_RANK0_SHARE``, else 1.0) scales the share of each window's work dealt to rank 0, the rank that also writes.
    ``timings`` gains ``wait_at_exchange`` (seconds this rank stood in the window's first collective: on ranks > 0 that is
    the wait for the slowest producer, and for rank 0 if its writer ever fell behind), ``consume`` (the writer's busy time),
    ``writer_backpressure`` (the sweep thread blocked on a full writer queue) and ``writer_drain`` (the tail after the last
    window)."""
    import os
    import torch
    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    if rank0_share is None:
        rank0_share = float(os.environ.get("MSPA_RANK0_SHARE", "1.0"))
    if per_rank is None:
        per_rank = int(os.environ.get("MSPA_WINDOW_PER_RANK", "8"))
    wins = windows(costs, world, per_rank, rank0_share=rank0_share)
    order = [i for w in wins for i in w[rank]]
    items = iter(work_items(order))
    timings = timings or Timings()
    writer = _WindowWriter(consume, record_width, timings, writer_depth) if rank == 0 else None
    from . import hostinfo
    n_enc = int(os.environ.get("MSPA_ENCODE_THREADS", "0")) or max(2, min(8, hostinfo.effective_cpus() // 2))
    encoders = ThreadPoolExecutor(max_workers=n_enc, thread_name_prefix="mspa-encode")

    def timed_encode(fn):
        with timings.span("encode_deferred"):
            return fn()

    def vote(failure: Optional[BaseException]) -> None:
        """One int per window: did every rank get through its scenes (and is rank 0's writer alive)?  Raises on EVERY rank."""
        if writer is not None and failure is None:
            failure = writer.error
        if ctx is not None:
            import torch.distributed as dist
            flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=ctx.collective_device)
            with timings.span("wait_at_exchange"):
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctx.group)
                failed = int(flag.item())
            if failed:
                if failure is not None:
                    raise failure
                raise RuntimeError("sharded_sweep: another rank failed in this window (its own traceback says why)")
        elif failure is not None:
            raise failure

    def finish(w, local_rows, local_blobs) -> None:
        """Window ``w``'s deferred blobs, its vote and its one exchange; rank 0 hands the result to its writer."""
        failure: Optional[BaseException] = None
        with timings.span("encode_wait"):
            for n, (index, blobs) in enumerate(local_blobs):
                try:
                    if hasattr(blobs, "result"):
                        blobs = blobs.result()
                    local_blobs[n] = (index, [b.result() if hasattr(b, "result") else b for b in blobs])
                except Exception as e:                     # a failure on an encoder thread counts like one in produce
                    failure = e
                    break
        vote(failure)
        with timings.span("exchange"):
            rows_by_index: Dict[int, np.ndarray] = {}
            if record_width is not None:
                dev = ctx.collective_device if ctx is not None else "cpu"
                local = torch.cat([r.to(dev) for r in local_rows], 0) if local_rows else \
                    torch.zeros((0, record_width + 1), dtype=torch.float64, device=dev)
                table = shard.collate_records(local, ctx, dst=0) if ctx is not None else local
                if rank == 0:
                    table = table.cpu().numpy()
                    tags = table[:, 0].astype(np.int64)
                    # each item's rows are contiguous (cat per item, ranks concatenated): cut at the tag changes
                    cuts = np.flatnonzero(np.diff(tags)) + 1 if len(tags) else np.zeros(0, np.int64)
                    for lo, hi in zip(np.concatenate([[0], cuts]).astype(np.int64), np.concatenate([cuts, [len(tags)]]).astype(np.int64)):
                        if hi > lo:
                            rows_by_index[int(tags[lo])] = table[lo:hi, 1:]
            packed = _pack_blobs(local_blobs)
            if ctx is not None:
                parts = shard.gather_bytes(packed, ctx, dst=0)
            else:
                parts = [np.frombuffer(packed, dtype=np.uint8)]
        if writer is not None:
            writer.put(sorted(i for b in w for i in b), rows_by_index, parts)

    # Every collective of the sweep -- a window's vote and its exchange -- runs on ONE thread of its own, in submission order, and
    # up to two windows behind the sweep thread: the encoder threads work on window w's deferred blobs (85 ms per visibility
    # index) while windows w + 1 and w + 2 are being produced, and the sweep thread never stands in `encode_wait` or in the
    # exchange (it used to, 0.34 s of a 0.80 s pass of the index sweep).  Every rank runs the same sequence of collectives --
    # finish(0), finish(1), ... -- so a rank whose produce fails in window w still finishes w - 1 and then brings the failure to
    # w's vote: everyone leaves together, the windows before the failing one are complete.
    exchanger = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mspa-exchange")
    behind: collections.deque = collections.deque()

    left = {"early": False}

    def on_exchange_thread(fn, *a):
        # Once a vote has failed every rank has left the sequence of collectives AT THAT VOTE: what was queued behind it must
        # not start another one (the other ranks would never join it).
        if left["early"]:
            return None
        if ctx is not None and getattr(ctx.device, "type", "cpu") == "cuda":
            torch.cuda.set_device(ctx.device)
        try:
            return fn(*a)
        except BaseException:
            left["early"] = True
            raise

    def settle(limit: int) -> None:
        while len(behind) > limit:
            behind.popleft().result()                      # raises what finish raised: its own failure, or the vote's

    try:
        for w in wins:
            local_rows, local_blobs = [], []
            for index in w[rank]:
                try:
                    item = next(items)
                    with timings.span("produce"):
                        records, blobs = produce(index, item)
                except Exception as e:                     # a missing file, a bad pose: the other ranks must not be left
                    settle(0)                              # waiting in a window's exchange (they would, until the timeout).
                    exchanger.submit(on_exchange_thread, vote, e).result()     # The windows before this one are complete; THIS
                    raise                                  # window's vote carries the failure (a world of one: vote raises it)
                if record_width is not None:
                    if records is None:
                        records = torch.zeros((0, record_width), dtype=torch.float64)
                    tagged = torch.empty((records.shape[0], record_width + 1), dtype=torch.float64, device=records.device)
                    tagged[:, 0] = index
                    tagged[:, 1:] = records
                    local_rows.append(tagged)
                if callable(blobs):                        # the whole list deferred (its members may share expensive work)
                    local_blobs.append((index, encoders.submit(timed_encode, blobs)))
                else:
                    local_blobs.append((index, [encoders.submit(timed_encode, b) if callable(b) else b for b in blobs]))
            settle(1)
            behind.append(exchanger.submit(on_exchange_thread, finish, w, local_rows, local_blobs))
        settle(0)
        for _ in items:                                        # drain: lets the prefetcher's generator finish cleanly
            pass
    except BaseException:
        exchanger.shutdown(wait=True)                      # (windows already submitted are exchanged or fail at their vote)
        encoders.shutdown(wait=True)
        if writer is not None:
            writer.close()          # windows exchanged before the failure are complete: they are written, then the error leaves
        raise
    exchanger.shutdown(wait=True)
    encoders.shutdown(wait=True)
    # the tail: rank 0 waits for its writer, and the ranks agree once more that nothing failed after the last window's vote
    vote(writer.close() if writer is not None else None)


SIDE_STREAMS = 4


def side_stream(device):
    """The calling THREAD's side stream on ``device``.  A sweep's ``produce`` only launches a scene's kernels; what has to wait for
    them -- K9 compaction, downloads, encoding -- runs as a deferred blob on an encoder thread, behind an event, on a side stream:
    beside the sweep thread's kernels instead of in front of the next scene's.  Keyed by the thread's NAME, folded onto
    ``SIDE_STREAMS`` streams: every sweep starts a new encoder pool whose threads carry the same names ("mspa-encode_0" ...), so a
    process that sweeps many times keeps using the same few streams instead of walking through torch's pool of 32 and landing on a
    decode slot's stream (``_lib.own_stream``); and eight decode slots + the copy stream + four side streams + the default stream
    stay inside the 16 hardware queues the runtime is given (mspa/__init__.py) -- what an encoder thread puts on its stream is an
    event wait and a few small downloads it then blocks on, two threads taking turns on one stream lose nothing."""
    from . import _lib
    name = threading.current_thread().name
    tail = name.rsplit("_", 1)[-1]
    k = int(tail) if tail.isdigit() else sum(name.encode())
    return _lib.own_stream(f"side-{k % SIDE_STREAMS}", device)


def prefetched_scenes(host_scenes: Iterable[HostScene], device="cuda", timings: Optional[Timings] = None,
                      decode_on_device: bool = False):
    """HostScenes -> resident ``SceneOnDevice`` objects through pinned staging on a copy stream (upload.ScenePrefetcher);
    a yielded scene stays valid until the next one is asked for.  ``decode_on_device``: the scenes carry compressed frames
    (``HostScene.packed``) -- more upload slots, so that several scenes' decode kernels run side by side."""
    from .upload import ScenePrefetcher
    yield from ScenePrefetcher(host_scenes, device=device, timings=timings, decode_on_device=decode_on_device)
