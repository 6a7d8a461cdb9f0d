"""Host -> device staging of scenes: pinned buffers, a copy stream, and a prefetching iterator.

A ScanNet sweep is bounded by getting the depth frames onto the device (196 MB for a 320-frame scene against 0.2 ms of
kernels), so the upload is what has to overlap: ``ScenePrefetcher`` stages scene n+1 into pinned memory on a worker
thread (NumPy copies and the LAPACK inverses release the GIL) and enqueues its H2D copy on a side stream while the caller
runs K1 / K2 / K4 on scene n; two slots, recycled through events, no allocation in steady state.

torch is plumbing here (pinned allocations, streams, events); nothing is computed.
"""
from __future__ import annotations

import queue
import threading
from typing import Iterable, Iterator, Optional

import numpy as np
import torch

from . import engine
from .scene import SceneOnDevice, valid_image_ids


class UploadSlot:
    """Pinned host buffers + device buffers of one scene; grown to the largest scene seen, then reused."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cap_frames = self.cap_points = 0
        self.depth_hw = None
        self.h_depth = self.d_depth = self.h_xyz = self.d_xyz = None
        self.h_fmats = self.d_fmats = self.h_cmats = self.d_cmats = None
        self.ready = torch.cuda.Event()          # recorded on the copy stream when the scene's tensors are resident
        self.free = None                         # recorded on the consumer's stream when it is done with them

    def _ensure(self, n_frames, depth_hw, n_points):
        if n_frames > self.cap_frames or depth_hw != self.depth_hw:
            self.cap_frames, self.depth_hw = max(n_frames, self.cap_frames), depth_hw
            shape = (self.cap_frames,) + tuple(depth_hw)
            self.h_depth = torch.empty(shape, dtype=torch.int16).pin_memory()
            self.d_depth = torch.empty(shape, dtype=torch.int16, device=self.device)
            self.h_fmats = torch.empty((self.cap_frames, engine._lib.FRAME_MATS, 16), dtype=torch.float64).pin_memory()
            self.d_fmats = torch.empty_like(self.h_fmats, device=self.device)
            self.h_cmats = torch.empty((self.cap_frames, 2, 16), dtype=torch.float64).pin_memory()
            self.d_cmats = torch.empty_like(self.h_cmats, device=self.device)
        if n_points > self.cap_points:
            self.cap_points = n_points
            self.h_xyz = torch.empty((n_points, 3), dtype=torch.float64).pin_memory()
            self.d_xyz = torch.empty((n_points, 3), dtype=torch.float64, device=self.device)

    def stage_and_upload(self, sc, copy_stream) -> SceneOnDevice:
        """Fill the pinned buffers from ``sc`` (K, A, E, depth, color_hw, points) and enqueue the copies on ``copy_stream``."""
        ids = valid_image_ids(sc.E)
        F = len(ids)
        first = next(iter(sc.depth.values()))
        points = getattr(sc, "points", None)
        N = 0 if points is None else int(points.shape[0])
        if self.free is not None:
            self.free.synchronize()              # the previous user of this slot has finished (host-side wait: we overwrite
        self._ensure(max(F, 1), tuple(first.shape), max(N, 1))       # pinned memory the earlier copy may still be reading)
        hd = self.h_depth.numpy()
        for k, i in enumerate(ids):              # straight into pinned memory: no np.stack temporary
            np.copyto(hd[k].view(np.uint16), sc.depth[i], casting="same_kind")
        K, A = np.asarray(sc.K, np.float64), np.asarray(sc.A, np.float64)
        E_al = [A @ np.asarray(sc.E[i], np.float64) for i in ids]
        if F:
            self.h_fmats.numpy()[:F] = engine.frame_matrices(K, A, [sc.E[i] for i in ids])
            self.h_cmats.numpy()[:F] = engine.camera_matrices(K, E_al)
        if N:
            np.copyto(self.h_xyz.numpy()[:N], np.asarray(points, np.float64)[:, :3])
        with torch.cuda.stream(copy_stream):
            self.d_depth[:F].copy_(self.h_depth[:F], non_blocking=True)
            self.d_fmats[:F].copy_(self.h_fmats[:F], non_blocking=True)
            self.d_cmats[:F].copy_(self.h_cmats[:F], non_blocking=True)
            if N:
                self.d_xyz[:N].copy_(self.h_xyz[:N], non_blocking=True)
            self.ready.record(copy_stream)
        return SceneOnDevice.from_resident(K, A, ids, E_al, self.d_depth[:F], self.d_fmats[:F], self.d_cmats[:F],
                                           self.d_xyz[:N] if N else None, tuple(sc.color_hw), self.device)


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

    def __init__(self, scenes: Iterable, device="cuda", slots: int = 2, threaded: bool = True):
        self.scenes = scenes
        self.device = torch.device(device)
        self.n_slots = max(2, int(slots))
        self.threaded = threaded

    def __iter__(self) -> Iterator[SceneOnDevice]:
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        copy_stream = torch.cuda.Stream(device=dev_index)
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

        def producer():
            try:
                torch.cuda.set_device(dev_index)
                for sc in self.scenes:
                    slot = None
                    while slot is None and not stop.is_set():
                        try:
                            slot = free_slots.get(timeout=0.05)
                        except queue.Empty:
                            pass
                    if slot is None or not put((slot.stage_and_upload(sc, copy_stream), slot, None)):
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
                        staged = (slot.stage_and_upload(nxt, copy_stream), slot)
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
        yield scene
        slot.free = torch.cuda.Event()
        slot.free.record(torch.cuda.current_stream())
        free_slots.put(slot)
