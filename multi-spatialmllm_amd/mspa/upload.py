"""Host -> device staging of scenes: pinned buffers, a copy stream, and a prefetching iterator.

A ScanNet sweep is bounded by getting the depth frames onto the device (196 MB for a 320-frame scene against 0.2 ms of
kernels), so the upload is what has to overlap: ``ScenePrefetcher`` stages scene n+1 into pinned memory on a worker
thread (NumPy copies and the LAPACK inverses release the GIL) and enqueues its H2D copy on a side stream while the caller
runs K1 / K2 / K4 on scene n; three slots, recycled through events, no allocation in steady state.

torch is plumbing here (pinned allocations, streams, events); nothing is computed.
"""
from __future__ import annotations

import os
import queue
import threading
import time
from typing import Iterable, Iterator, Optional

import numpy as np
import torch

from . import engine
from .scene import SceneOnDevice, valid_image_ids


# Three slots, not two: with two, the producer cannot start staging scene n+2 before the consumer has released scene n, and
# staging (3 ms), H2D (4 ms) and the consumer's kernels + download (1.2 ms) run back to back instead of side by side.
UPLOAD_SLOTS = int(os.environ.get("MSPA_UPLOAD_SLOTS", "3"))
# Scenes whose depth frames are decoded ON the device need many of them in flight: one wave inflates one frame and a lone wave
# takes ~70 ms for a 640 x 480 frame, so a 320-frame scene alone keeps 320 of the chip's ~3 600 such waves busy; eight scenes side
# by side (each on its own stream) run at 27 k frames/s (profiles/r06_device_ingest.md).  Slots cost HBM, not host time: depth +
# scanline scratch + compressed bytes = 0.5 GB per 320-frame scene.  The FRAMES in flight are capped as well: a decode wave owns
# 9 KB (11 KB until the ring shrank to 2 KB) of its compute unit's LDS for its whole life, 14 of them left nothing for the geometry kernels of the scene being
# consumed (K1: 13.4 KB per workgroup), which then wait milliseconds for a wave to retire (measured: 16 ms per scene with 3 200
# frames in flight); at 2 560 (10 per unit) a third of every unit's LDS stays free.
DECODE_SLOTS = int(os.environ.get("MSPA_DECODE_SLOTS", "8"))
DECODE_MAX_FRAMES = int(os.environ.get("MSPA_DECODE_MAX_FRAMES", "2560"))
# What takes a scene's frames off the in-flight count: "taken" -- the consumer has taken the scene (with 8 slots the slots bind
# first: a new scene is admitted when the consumer RELEASES one); "complete" -- its decode kernels have retired on the device
# (an event query), so that with more slots than DECODE_MAX_FRAMES / frames-per-scene the cap's worth of frames is decoding at all times.
DECODE_GATE = os.environ.get("MSPA_DECODE_GATE", "taken")
# Tried and dropped as the default: decode streams that leave 32 compute units alone (mspa_stream_create_reserving, a CU mask).
# Every masked stream is a hardware queue of its own; ten of them next to the process's other queues oversubscribe the queue
# slots and the firmware time-slices them: 47 -> 12.5 scenes/s (profiles/r06_dropin_decode.md).  0 = plain streams.
DECODE_RESERVED_CUS = int(os.environ.get("MSPA_DECODE_RESERVED_CUS", "0"))


_DECODE_STREAMS_MADE = [0]


def _decode_stream(device) -> "torch.cuda.Stream":
    """A stream of its own for one slot's on-device decode (never one of torch's 32 pooled streams: ``_lib.own_stream``); with
    DECODE_RESERVED_CUS, CU-masked so that that many compute units stay free for other kernels."""
    import ctypes
    from . import _lib
    if DECODE_RESERVED_CUS <= 0:
        _DECODE_STREAMS_MADE[0] += 1
        return _lib.own_stream(f"decode-slot-{_DECODE_STREAMS_MADE[0]}", device)
    ptr = ctypes.c_void_p(0)
    with torch.cuda.device(device):
        info = _lib.device_info(torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
        reserve = DECODE_RESERVED_CUS if info["n_cu"] >= 4 * max(1, DECODE_RESERVED_CUS) else 0
        _lib.check(_lib.load().mspa_stream_create_reserving(int(reserve), ctypes.byref(ptr)))
    return torch.cuda.ExternalStream(ptr.value, device=device)
STAGE_THREADS = int(os.environ.get("MSPA_STAGE_THREADS", "4"))           # native copy threads per staged chunk
STAGE_CHUNK_FRAMES = int(os.environ.get("MSPA_STAGE_CHUNK_FRAMES", "160"))  # frames per chunk (98 MB at 640 x 480)


def prepare_tables(K, A, E, points=None, ids=None) -> dict:
    """Everything ``UploadSlot.stage_and_upload`` derives from a scene's poses and vertices, as plain arrays: the frames with a
    finite pose, ``A @ E`` per frame, the frame / camera records of the kernels, K4's pose tables, the vertices' xyz columns as
    one contiguous block.  A loader thread MAY compute this ahead (``sweep.HostScene.prepared``, ``MSPA_PREPARE_ON_LOADER=1``) so
    that the one staging thread only copies the arrays into its page-locked buffers; without it the staging thread calls this
    function itself, so the bytes are the same either way.  Measured (tools/ab_prepare.sh, profiles/r06_sweep_timeline.md): the
    staging thread's 6-7 ms per scene drop to 3 ms -- and the from-disk sweep gets SLOWER (88-117 against 122-135 scenes/s, five
    A/B rounds in two boxes): with scenes arriving faster the device-side H2D and inflate of each scene take 1.3-1.5 x as long
    and the slots are held longer than the staging saved.  Off by default."""
    if ids is None:
        ids = valid_image_ids(E)
    F = len(ids)
    K, A = np.asarray(K, np.float64), np.asarray(A, np.float64)
    # one batched matmul; bit-identical to A @ E per frame (tests/test_host_cpu.py)
    E_al = list(np.matmul(A, np.stack([np.asarray(E[i], np.float64) for i in ids]))) if F else []
    out = {"ids": ids, "K": K, "A": A, "E_al": E_al, "fmats": None, "cmats": None, "pose": None, "xyz": None}
    if F:
        out["fmats"] = engine.frame_matrices(K, A, [E[i] for i in ids])
        out["cmats"] = engine.camera_matrices(K, E_al)
        # K4's per-frame tables travel with the scene: uploaded by the consumer they would be pageable copies queued
        # behind the next scene's 197 MB on the same copy engine
        pose = np.empty((18 * F,), dtype=np.float64)
        pose[:16 * F].reshape(F, 16)[:] = np.stack(E_al).reshape(F, 16)
        pose[16 * F:17 * F], pose[17 * F:18 * F] = engine.extract_yaw_pitch_host(E_al)
        out["pose"] = pose
    if points is not None and int(points.shape[0]):
        out["xyz"] = np.ascontiguousarray(np.asarray(points, np.float64)[:, :3])
    return out


class UploadSlot:
    """Pinned host buffers + device buffers of one scene; grown to the largest scene seen, then reused."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cap_frames = self.cap_points = 0
        self.depth_hw = None
        self.h_depth = self.d_depth = self.h_xyz = self.d_xyz = None
        self.h_fmats = self.d_fmats = self.h_cmats = self.d_cmats = self.h_pose = self.d_pose = None
        self.ready = torch.cuda.Event()          # recorded on the copy stream when the scene's tensors are resident
        self.free = None                         # recorded on the consumer's stream when it is done with them
        # on-device decode of compressed frames (ingest.PackedDepth): the slot's own stream, scanline scratch, per-frame status
        self.stream = None
        self.d_comp = self.d_raw = self.d_status = self.d_off = self.d_nb = None
        self.h_status = self.h_off = self.h_nb = None
        self.pending_decode = None               # (PackedDepth, F): status still to be looked at (ScenePrefetcher._consume)
        self.decoded = None                      # event behind the slot's decode kernels

    def _ensure(self, n_frames, depth_hw, n_points, copy_stream, host_depth=True):
        grown = False
        if n_frames > self.cap_frames or depth_hw != self.depth_hw:
            grown = True
            self.cap_frames, self.depth_hw = max(n_frames, self.cap_frames), depth_hw
            shape = (self.cap_frames,) + tuple(depth_hw)
            self.h_depth = None                  # pinned staging of DECODED frames: only for scenes that arrive decoded
            self.d_depth = torch.empty(shape, dtype=torch.int16, device=self.device)
            self.h_fmats = torch.empty((self.cap_frames, engine._lib.FRAME_MATS, 16), dtype=torch.float64).pin_memory()
            self.d_fmats = torch.empty_like(self.h_fmats, device=self.device)
            self.h_cmats = torch.empty((self.cap_frames, engine._lib.CAM_MATS, 16), dtype=torch.float64).pin_memory()
            self.d_cmats = torch.empty_like(self.h_cmats, device=self.device)
            self.h_pose = torch.empty((self.cap_frames * 18,), dtype=torch.float64).pin_memory()   # A @ E, yaw, pitch (K4)
            self.d_pose = torch.empty_like(self.h_pose, device=self.device)
        if host_depth and (self.h_depth is None or tuple(self.h_depth.shape) != tuple(self.d_depth.shape)):
            self.h_depth = torch.empty(tuple(self.d_depth.shape), dtype=torch.int16).pin_memory()
        if n_points > self.cap_points:
            grown = True
            self.cap_points = n_points
            self.h_xyz = torch.empty((n_points, 3), dtype=torch.float64).pin_memory()
            self.d_xyz = torch.empty((n_points, 3), dtype=torch.float64, device=self.device)
        if grown:
            # The caching allocator hands out blocks on the ALLOCATING thread's current stream and may return one the
            # consumer has just freed there while its last kernel (a K2 workspace, pair outputs) is still queued.  The
            # copies below run on `copy_stream`: order them behind everything already enqueued on the allocation stream.
            copy_stream.wait_stream(torch.cuda.current_stream(self.device))

    def _ensure_decode(self, n_frames, depth_hw, comp_bytes):
        h, w = depth_hw
        pitch = (h * (2 * w + 1) + 255) // 256 * 256
        if self.d_raw is None or self.d_raw.shape[0] < n_frames or self.d_raw.shape[1] != pitch:
            self.d_raw = torch.empty((max(n_frames, self.cap_frames), pitch), dtype=torch.uint8, device=self.device)
            self.d_status = torch.zeros((self.d_raw.shape[0],), dtype=torch.int32, device=self.device)
            self.d_off = torch.zeros((self.d_raw.shape[0],), dtype=torch.int64, device=self.device)
            self.d_nb = torch.zeros_like(self.d_off)
            self.h_status = torch.zeros((self.d_raw.shape[0],), dtype=torch.int32).pin_memory()
            self.h_off = torch.zeros((self.d_raw.shape[0],), dtype=torch.int64).pin_memory()
            self.h_nb = torch.zeros_like(self.h_off).pin_memory()
        if self.d_comp is None or self.d_comp.numel() < comp_bytes:
            self.d_comp = torch.empty(((int(comp_bytes) + (16 << 20) - 1) // (16 << 20) * (16 << 20),), dtype=torch.uint8,
                                      device=self.device)

    def _upload_and_decode(self, packed, F, stream):
        """The compressed frames -> ``d_depth[:F]`` on ``stream``: one H2D copy of the packed bytes, one wave per frame inflates
        (mspa_inflate_blocks_device), one wave per frame undoes the row filters (mspa_png_unfilter_device).  The per-frame
        status comes back asynchronously; ``finish_decode`` looks at it once ``ready`` has passed."""
        h, w = packed.hw
        grown = self.d_raw is None or self.d_raw.shape[0] < F or self.d_comp is None or self.d_comp.numel() < packed.capacity
        self._ensure_decode(F, (h, w), packed.capacity)
        if grown:                                    # fresh blocks of the caching allocator: behind whatever the allocating
            stream.wait_stream(torch.cuda.current_stream(self.device))      # thread's stream still has queued on them
        self.h_off[:F] = torch.from_numpy(packed.offsets)
        self.h_nb[:F] = torch.from_numpy(np.where(packed.status == 0, packed.nbytes, 0))    # a declined file: an empty stream
        with torch.cuda.stream(stream):
            self.d_comp[:packed.capacity].copy_(packed.buf[:packed.capacity], non_blocking=True)
            self.d_off[:F].copy_(self.h_off[:F], non_blocking=True)
            self.d_nb[:F].copy_(self.h_nb[:F], non_blocking=True)
            engine.inflate_blocks_device(self.d_comp, self.d_off[:F], self.d_nb[:F], h * (2 * w + 1), self.d_raw, self.d_status)
            engine.png_unfilter_device(self.d_raw[:F], h, w, self.d_status, self.d_depth[:F])
            self.h_status[:F].copy_(self.d_status[:F], non_blocking=True)
            if self.decoded is None:
                self.decoded = torch.cuda.Event()
            self.decoded.record(stream)          # the scene's decode waves have retired (DECODE_GATE = "complete" polls it)
        self.pending_decode = (packed, F)

    def finish_decode(self):
        """After ``ready``: frames the device declined (another pixel format, a damaged stream, a failed checksum) are decoded
        by the host reader and copied in on the current stream -- the result is the host path's for every input.  Hands the
        packed buffer back to its pool."""
        if self.pending_decode is None:
            return
        packed, F = self.pending_decode
        self.pending_decode = None
        self.ready.synchronize()
        bad = np.nonzero(self.h_status[:F].numpy())[0]
        if len(bad):
            from . import ingest
            from spatial_engine.utils.scannet_utils.handler import _images
            host = ingest.read_depth_frames([packed.paths[int(k)] for k in bad], 4, general_reader=_images.read_depth)
            self.d_depth[torch.from_numpy(bad).to(self.device)] = torch.from_numpy(host.view(np.int16)).to(self.device)
        packed.release()

    def stage_and_upload(self, sc, copy_stream) -> SceneOnDevice:
        """Fill the pinned buffers from ``sc`` (K, A, E, depth, color_hw, points) and enqueue the copies on ``copy_stream``.
        A scene that carries ``packed`` compressed frames (``sweep.HostScene.packed``) is decoded on the device instead, on
        the slot's own stream, so that several scenes' decodes run side by side."""
        prep = getattr(sc, "prepared", None)
        ids = prep["ids"] if prep is not None else valid_image_ids(sc.E)
        F = len(ids)
        packed = getattr(sc, "packed", None)
        if packed is not None:
            if self.stream is None:
                self.stream = _decode_stream(self.device)       # lives as long as the slot (slots are pooled per process)
            copy_stream = self.stream
        # a scene without depth frames is an empty scene (as SceneOnDevice treats it), not a StopIteration
        first_shape = tuple(packed.hw) if packed is not None else \
            tuple(next(iter(sc.depth.values())).shape) if len(sc.depth) else (self.depth_hw or tuple(sc.color_hw))
        points = getattr(sc, "points", None)
        N = 0 if points is None else int(points.shape[0])
        if self.free is not None:
            self.free.synchronize()              # the previous user of this slot has finished (host-side wait: we overwrite
        self._ensure(max(F, 1), first_shape, max(N, 1), copy_stream, host_depth=packed is None)  # pinned memory the earlier copy may still be reading)
        # Depth frames: straight into pinned memory (no np.stack temporary), a chunk of frames at a time by native copy
        # threads, each chunk's H2D enqueued as soon as it is staged -- the link is busy while the next chunk is gathered
        # and while the matrices below are prepared.
        hd = self.h_depth.numpy().view(np.uint16) if packed is None else None
        frames = []
        for i in (ids if packed is None else ()):
            f = sc.depth[i]
            if f.dtype != np.uint16 or not f.flags.c_contiguous:
                f = np.ascontiguousarray(f, dtype=np.uint16)
            frames.append(f)
        def depth_job():
            if packed is not None:                      # compressed frames: H2D + decode kernels on the slot's stream
                torch.cuda.set_device(copy_stream.device)
                if list(sc.depth_ids) != list(ids):
                    raise ValueError("stage_and_upload: the packed frames are not the scene's valid frames")
                self._upload_and_decode(packed, F, copy_stream)
                return                                # on a helper thread: the gather holds no interpreter lock, so the matrix
            torch.cuda.set_device(copy_stream.device)   # preparation below runs beside it
            with torch.cuda.stream(copy_stream):
                for lo in range(0, F, STAGE_CHUNK_FRAMES):
                    hi = min(F, lo + STAGE_CHUNK_FRAMES)
                    engine.gather_blocks_host(frames[lo:hi], hd[lo:hi], STAGE_THREADS)
                    self.d_depth[lo:hi].copy_(self.h_depth[lo:hi], non_blocking=True)

        depth_done = _stage_pool().submit(depth_job)
        try:
            if prep is None:                      # (a scene that arrives with its tables -- the sweeps' loader threads -- skips this)
                prep = prepare_tables(sc.K, sc.A, sc.E, points, ids)
            K, A, E_al = prep["K"], prep["A"], prep["E_al"]
            if F:
                self.h_fmats.numpy()[:F] = prep["fmats"]
                self.h_cmats.numpy()[:F] = prep["cmats"]
                self.h_pose.numpy()[:18 * F] = prep["pose"]
            if N:
                np.copyto(self.h_xyz.numpy()[:N], prep["xyz"])
        except BaseException:
            depth_done.exception()               # a bad pose must not leave the helper writing into a slot that is handed back
            raise
        depth_done.result()                      # re-raises what the helper raised; the depth copies are enqueued
        with torch.cuda.stream(copy_stream):
            self.d_fmats[:F].copy_(self.h_fmats[:F], non_blocking=True)
            self.d_cmats[:F].copy_(self.h_cmats[:F], non_blocking=True)
            self.d_pose[:18 * F].copy_(self.h_pose[:18 * F], non_blocking=True)
            if N:
                self.d_xyz[:N].copy_(self.h_xyz[:N], non_blocking=True)
            self.ready.record(copy_stream)
        scene = SceneOnDevice.from_resident(K, A, ids, E_al, self.d_depth[:F], self.d_fmats[:F], self.d_cmats[:F],
                                            self.d_xyz[:N] if N else None, tuple(sc.color_hw), self.device,
                                            pose_tables=(self.d_pose[:16 * F].view(F, 16), self.d_pose[16 * F:17 * F],
                                                         self.d_pose[17 * F:18 * F]))
        scene.depth_scale = float(getattr(sc, "depth_scale", 0.001))      # the handler's depth_value_scale (IH:76)
        return scene


_STAGE_POOL = None


def _stage_pool():
    """One helper thread for the depth gather of the scene being staged (created on first use)."""
    global _STAGE_POOL
    if _STAGE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _STAGE_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mspa-stage")
    return _STAGE_POOL


_SLOT_POOL: dict = {}     # device -> idle UploadSlots: pinning 40-200 MB costs tens of milliseconds, so slots outlive a prefetcher
_SLOT_POOL_LOCK = threading.Lock()


def _take_slot(device) -> UploadSlot:
    with _SLOT_POOL_LOCK:
        pool = _SLOT_POOL.setdefault(str(device), [])
        if pool:
            return pool.pop()
    return UploadSlot(device)


def _give_slots(device, slots):
    with _SLOT_POOL_LOCK:
        _SLOT_POOL.setdefault(str(device), []).extend(slots)


class ScenePrefetcher:
    """``for scene in ScenePrefetcher(host_scenes): ...`` -- yields ``SceneOnDevice`` objects whose tensors are already
    resident (the consumer's stream waits on the upload event, the host does not), while the next scene is being staged and
    copied.  A yielded scene is valid until the next iteration step (its slot is recycled two scenes later)."""

    def __init__(self, scenes: Iterable, device="cuda", slots: Optional[int] = None, threaded: bool = True, timings=None,
                 decode_on_device: bool = False):
        self.scenes = scenes
        self.device = torch.device(device)
        if slots is None:
            slots = DECODE_SLOTS if decode_on_device else UPLOAD_SLOTS
        self.n_slots = max(2, int(slots))
        self.threaded = threaded
        self.timings = timings                        # mspa.sweep.Timings: "stage" = pinned staging + H2D enqueue, per scene

    def _stage(self, slot, sc, copy_stream):
        if self.timings is None:
            return slot.stage_and_upload(sc, copy_stream)
        with self.timings.span("stage"):
            return slot.stage_and_upload(sc, copy_stream)

    def __iter__(self) -> Iterator[SceneOnDevice]:
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        from . import _lib
        copy_stream = _lib.own_stream("sweep-copy", dev_index)       # (not torch.cuda.Stream(): see _lib.own_stream)
        free_slots: "queue.Queue[UploadSlot]" = queue.Queue()
        my_slots = [_take_slot(self.device) for _ in range(self.n_slots)]
        for sl in my_slots:
            free_slots.put(sl)
        ready: "queue.Queue" = queue.Queue(maxsize=self.n_slots)
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        in_flight = {"frames": 0}
        in_flight_lock = threading.Lock()

        def producer():
            try:
                torch.cuda.set_device(dev_index)
                for sc in self.scenes:
                    n_packed = len(sc.packed) if getattr(sc, "packed", None) is not None else 0
                    while n_packed and not stop.is_set():       # frames being decoded on the device: capped (see DECODE_MAX_FRAMES)
                        with in_flight_lock:
                            if DECODE_GATE == "complete":
                                for sl in my_slots:
                                    if getattr(sl, "in_flight_frames", 0) and sl.decoded is not None and getattr(sl, "decode_launched", False) \
                                            and sl.decoded.query():
                                        in_flight["frames"] -= sl.in_flight_frames
                                        sl.in_flight_frames = 0
                            if in_flight["frames"] == 0 or in_flight["frames"] + n_packed <= DECODE_MAX_FRAMES:
                                in_flight["frames"] += n_packed
                                break
                        time.sleep(0.0005)
                    slot = None
                    while slot is None and not stop.is_set():
                        try:
                            slot = free_slots.get(timeout=0.05)
                        except queue.Empty:
                            pass
                    if slot is not None:
                        slot.decode_launched = False
                        slot.in_flight_frames = n_packed
                    if slot is None:
                        return
                    staged = self._stage(slot, sc, copy_stream)
                    slot.decode_launched = True              # (its `decoded` event is this scene's from here on)
                    if not put((staged, slot, None)):
                        return
            except BaseException as e:                 # surfaces in the consumer
                put((None, None, e))
                return
            put((None, None, None))

        if self.threaded:
            worker: Optional[threading.Thread] = threading.Thread(target=producer, daemon=True)
            worker.start()
        else:
            worker = None
        try:
            if worker is None:                          # same protocol without the thread (deterministic order for tests)
                it = iter(self.scenes)
                pending = None
                while True:
                    nxt = next(it, None)
                    staged = None
                    if nxt is not None:
                        slot = free_slots.get()
                        staged = (self._stage(slot, nxt, copy_stream), slot)
                    if pending is not None:
                        yield from self._consume(pending, free_slots)
                    if staged is None:
                        break
                    pending = staged
                return
            while True:
                scene, slot, err = ready.get()
                if err is not None:
                    raise err
                if scene is None:
                    break
                cur = torch.cuda.current_stream()
                cur.wait_event(slot.ready)
                slot.finish_decode()                    # (host wait for this scene's decode) -> its frames leave the in-flight count
                with in_flight_lock:
                    in_flight["frames"] -= getattr(slot, "in_flight_frames", 0)
                    slot.in_flight_frames = 0
                yield from self._consume((scene, slot), free_slots)
        finally:
            stop.set()                                  # an abandoned iteration: the producer leaves at its next queue poll
            if worker is not None:
                worker.join()
            torch.cuda.synchronize(dev_index)           # neither stream touches the slots any more
            _give_slots(self.device, my_slots)

    @staticmethod
    def _consume(item, free_slots):
        scene, slot = item
        cur = torch.cuda.current_stream()
        cur.wait_event(slot.ready)                      # device-side dependency; the host does not block
        slot.finish_decode()                            # frames decoded on the device: their status, the host reader for the rest
        yield scene
        slot.free = torch.cuda.Event()
        slot.free.record(torch.cuda.current_stream())
        free_slots.put(slot)
