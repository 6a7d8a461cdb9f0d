"""Host-side ingest of on-disk scenes: a scene's depth frames read and decoded by native threads.

The reference reads one frame per call (``cv2.imread(path, -1)``, info_handler.py:149-155) inside the per-image loops of
CFR.process_scene / MVI.process_scene and hides that behind a process pool over scenes (CFR:222-229, MVI:151-156).  Here one
process feeds one GPU: ``read_depth_frames`` hands the file names of a scene's frames to ``mspa_read_depth_png_host``
(csrc/host_ingest.hip), which fills one contiguous [F, h, w] uint16 block without touching the interpreter.  Frames in a
format the native reader does not take (status 2: not 16-bit greyscale non-interlaced PNG) go through the caller's general
reader one by one -- that is image decoding, not geometry: there is no fallback for the kernels anywhere.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, Optional, Sequence

import numpy as np

from . import _lib


class BlockPool:
    """Reusable destinations for ``read_depth_frames``.  A fresh 39-200 MB NumPy array per scene has every page faulted in by
    the decode threads as they first touch it -- with 25-64 threads on one address space that serialises on the kernel's memory
    map: 7.4 ms per 64-frame scene into a fresh array against 5.2 ms into a reused one on 25 threads, 6.1 against 2.7 ms on 64
    (tools/ingest_bench.py on the GPU box, round 5).  Buffers are flat, sized in 32 MB steps, handed out as [F, h, w] views and
    come back when the scene that holds them is garbage-collected (``weakref.finalize`` in the handler)."""
    STEP = 32 << 20

    def __init__(self, max_free: int = 6):
        import threading
        self._free, self._lock, self.max_free = [], threading.Lock(), max_free

    def take(self, shape) -> np.ndarray:
        need = int(np.prod(shape)) * 2
        with self._lock:
            fit = [b for b in self._free if need <= b.nbytes <= max(2 * need, self.STEP)]
            if fit:
                buf = min(fit, key=lambda b: b.nbytes)
                self._free = [b for b in self._free if b is not buf]
            else:
                buf = None
        if buf is None:
            buf = np.empty(max(self.STEP, -(-need // self.STEP) * self.STEP), dtype=np.uint8)
        return buf[:need].view(np.uint16).reshape(shape)

    def give(self, block: np.ndarray):
        base = block
        while isinstance(base.base, np.ndarray):
            base = base.base
        with self._lock:
            if len(self._free) < self.max_free and all(b is not base for b in self._free):
                self._free.append(base)


DEFAULT_POOL = BlockPool(max_free=8)      # one per process: destinations outlive a handler (a split's scenes, then the next split's)


def png_header(path: str):
    """(h, w, bit_depth, colour_type, interlace) of a PNG file's IHDR."""
    v = [ctypes.c_int32(0) for _ in range(5)]
    _lib.check(_lib.load().mspa_png_header_host(path.encode(), *[ctypes.byref(x) for x in v]))
    return tuple(x.value for x in v)


def read_depth_frames(paths: Sequence[str], n_threads: int = 8, general_reader: Optional[Callable[[str], np.ndarray]] = None,
                      memory: Optional[Dict[str, np.ndarray]] = None, out: Optional[np.ndarray] = None,
                      pool: Optional[BlockPool] = None) -> np.ndarray:
    """[F, h, w] uint16: the depth frames at ``paths``, in order.  ``memory`` (path -> array) serves frames that are not on
    disk (synthetic scenes, tests); ``general_reader(path)`` decodes what the native reader declines.  ``out`` may be a
    preallocated (e.g. pinned) destination of the right shape; ``pool`` a ``BlockPool`` to take it from (the caller gives it back)."""
    paths = list(paths)
    F = len(paths)
    memory = memory or {}
    if F == 0:
        return np.zeros((0, 0, 0), dtype=np.uint16) if out is None else out[:0]
    on_disk = [k for k, p in enumerate(paths) if p not in memory]
    if on_disk:
        try:
            h, w, bits, ctype, lace = png_header(paths[on_disk[0]])
            native = bits == 16 and ctype == 0 and lace == 0
        except _lib.MspaError:
            if general_reader is None:
                raise
            first = np.asarray(general_reader(paths[on_disk[0]]))
            (h, w), native = first.shape[:2], False
    else:
        h, w = memory[paths[0]].shape[:2]
        native = False
    if out is None:
        out = pool.take((F, h, w)) if pool is not None else np.empty((F, h, w), dtype=np.uint16)
    elif out.shape != (F, h, w) or out.dtype != np.uint16 or not out.flags.c_contiguous:
        raise ValueError("read_depth_frames: `out` must be a C-contiguous uint16 array of shape (F, h, w)")

    def general(k):
        if general_reader is None:
            raise ValueError(f"{paths[k]}: not a 16-bit greyscale PNG and no general reader was given")
        a = np.asarray(general_reader(paths[k]))
        if a.shape[:2] != (h, w):
            raise ValueError(f"{paths[k]}: frame of {a.shape[:2]}, the scene's first frame is {(h, w)}")
        out[k] = a if a.ndim == 2 else a[..., 0]

    for k, p in enumerate(paths):
        if p in memory:
            a = np.asarray(memory[p])
            if a.shape[:2] != (h, w):
                raise ValueError(f"{p}: frame of {a.shape[:2]}, the scene's first frame is {(h, w)}")
            out[k] = a
    if on_disk and not native:
        for k in on_disk:
            general(k)
    elif on_disk:
        if len(on_disk) == F:
            dst, idx = out, None
        else:                                              # a mix of registered and on-disk frames: decode, then scatter
            dst, idx = np.empty((len(on_disk), h, w), dtype=np.uint16), on_disk
        enc = [paths[k].encode() for k in on_disk]
        arr = (ctypes.c_char_p * len(enc))(*enc)
        status = np.zeros(len(enc), dtype=np.int32)
        _lib.check(_lib.load().mspa_read_depth_png_host(arr, len(enc), h, w, dst.ctypes.data, int(max(1, n_threads)),
                                                        status.ctypes.data))
        if idx is not None:
            out[idx] = dst
        for j in np.nonzero(status)[0]:
            k, st = on_disk[int(j)], int(status[j])
            if st == 1:
                raise FileNotFoundError(paths[k])
            if st == 2:
                general(k)
            else:
                raise ValueError(f"{paths[k]}: corrupt PNG stream")
    return out


class PinnedBytesPool:
    """Reusable page-locked byte buffers for the packed compressed frames of a scene (what the H2D copy reads).  Pinning 100 MB
    costs tens of milliseconds, so buffers are kept; sized in 16 MB steps; ``take`` returns a flat uint8 torch tensor."""
    STEP = 16 << 20

    def __init__(self, max_free: int = 12):
        import threading
        self._free, self._lock, self.max_free = [], threading.Lock(), max_free

    def take(self, nbytes: int):
        import torch
        with self._lock:
            fit = [b for b in self._free if b.numel() >= nbytes]
            if fit:
                buf = min(fit, key=lambda b: b.numel())
                self._free = [b for b in self._free if b is not buf]
                return buf
        n = max(self.STEP, -(-int(nbytes) // self.STEP) * self.STEP)
        t = torch.empty(n, dtype=torch.uint8)
        try:
            # (not torch.empty(..., pin_memory=True): page-locked memory nobody has touched yet makes the pack threads' reads fault it in
            # and the buffer's first H2D crawl -- a cold 96-scene pass took 2.3-2.6 s that way against 1.3 s; pin_memory() copies
            # from `t` and thereby touches every page.  tools/sweep_timeline.py --passes 1)
            return t.pin_memory()
        except RuntimeError:                                # no GPU in this process (CPU tests): pageable memory works too
            return t

    def give(self, buf):
        with self._lock:
            if len(self._free) < self.max_free and all(b is not buf for b in self._free):
                self._free.append(buf)


PINNED_POOL = PinnedBytesPool()


class PackedDepth:
    """A scene's depth frames as they sit on disk, packed for ONE H2D copy: ``buf`` (flat uint8, page-locked) holds frame k's
    scanline zlib stream at ``offsets[k]``, ``nbytes[k]`` long; ``status[k]`` != 0 marks a file the packer declined (the
    device decode then reports it and the host reader takes that frame).  ``release()`` hands the buffer back to its pool."""

    def __init__(self, paths, hw, buf, offsets, nbytes, status, capacity, pool=None):
        self.paths, self.hw, self.buf, self.offsets, self.nbytes, self.status, self.capacity = \
            list(paths), tuple(hw), buf, offsets, nbytes, status, int(capacity)
        self._pool = pool

    def __len__(self):
        return len(self.paths)

    def release(self):
        if self._pool is not None and self.buf is not None:
            self._pool.give(self.buf)
        self.buf = None


def pack_scene_depth(paths: Sequence[str], n_threads: int = 8, pool: Optional[PinnedBytesPool] = None) -> Optional[PackedDepth]:
    """``PackedDepth`` of the 16-bit greyscale PNG files at ``paths`` (sized by the first one), or None when the first file is
    not such a PNG -- the caller then reads the scene with ``read_depth_frames``."""
    paths = list(paths)
    if not paths:
        return None
    try:
        h, w, bits, ctype, lace = png_header(paths[0])
    except _lib.MspaError:
        return None
    if bits != 16 or ctype != 0 or lace != 0:
        return None
    pool = pool or PINNED_POOL
    F = len(paths)
    enc = [q.encode() for q in paths]
    arr = (ctypes.c_char_p * F)(*enc)
    offsets, nbytes = np.zeros(F, dtype=np.int64), np.zeros(F, dtype=np.int64)
    status = np.zeros(F, dtype=np.int32)
    need = ctypes.c_int64(0)
    lib = _lib.load()
    _lib.check(lib.mspa_png_pack_idat_host(arr, F, int(h), int(w), None, 0, offsets.ctypes.data, nbytes.ctypes.data,
                                           status.ctypes.data, ctypes.byref(need), 1))
    cap = max(int(need.value), 16)
    buf = pool.take(cap)
    _lib.check(lib.mspa_png_pack_idat_host(arr, F, int(h), int(w), buf.data_ptr(), int(buf.numel()), offsets.ctypes.data,
                                           nbytes.ctypes.data, status.ctypes.data, ctypes.byref(need), int(max(1, n_threads))))
    return PackedDepth(paths, (h, w), buf, offsets, nbytes, status, cap, pool)


def pack_depth_pngs(paths: Sequence[str], h: int, w: int, n_threads: int = 8, out=None):
    """The scanline zlib streams of the 16-bit greyscale ``h x w`` PNG files at ``paths``, packed into ONE host buffer for one
    H2D copy (mspa_png_pack_idat_host: files read into their slots by native threads, IDAT payloads moved to the slot's
    front).  ``out``: a uint8 NumPy array to pack into (e.g. a view of pinned memory), grown by the caller when the returned
    capacity exceeds it.  Returns (buffer, offsets int64 [F], nbytes int64 [F], status int32 [F], capacity)."""
    paths = list(paths)
    F = len(paths)
    enc = [p.encode() for p in paths]
    arr = (ctypes.c_char_p * max(F, 1))(*enc)
    offsets, nbytes = np.zeros(F, dtype=np.int64), np.zeros(F, dtype=np.int64)
    status = np.zeros(F, dtype=np.int32)
    need = ctypes.c_int64(0)
    lib = _lib.load()
    _lib.check(lib.mspa_png_pack_idat_host(arr, F, int(h), int(w), None, 0, offsets.ctypes.data, nbytes.ctypes.data,
                                           status.ctypes.data, ctypes.byref(need), int(n_threads)))
    cap = int(need.value)
    if out is None or out.nbytes < cap:
        out = np.empty(max(cap, 16), dtype=np.uint8)
    _lib.check(lib.mspa_png_pack_idat_host(arr, F, int(h), int(w), out.ctypes.data, int(out.nbytes), offsets.ctypes.data,
                                           nbytes.ctypes.data, status.ctypes.data, ctypes.byref(need), int(max(1, n_threads))))
    return out, offsets, nbytes, status, cap


def read_depth_frames_device(paths: Sequence[str], device="cuda", n_threads: int = 8, out=None,
                             general_reader: Optional[Callable[[str], np.ndarray]] = None, hw=None):
    """[F, h, w] int16 device tensor (the uint16 depth values): the depth PNGs at ``paths`` DECODED ON THE DEVICE -- the host only
    reads the files and packs their compressed scanline streams (``pack_depth_pngs``), one H2D copy carries them, one wave per
    frame inflates (mspa_inflate_blocks_device), one wave per frame undoes the row filters (mspa_png_unfilter_device).  A
    frame the device declines (another pixel format, a damaged stream, a failed checksum) is decoded by the host path
    (``read_depth_frames``) and uploaded on its own, so the result is the host path's result for every input.  Synchronises
    once (the per-frame status has to be read)."""
    import torch
    from . import engine
    paths = list(paths)
    F = len(paths)
    if F == 0:
        return torch.zeros((0, 0, 0), dtype=torch.int16, device=device)
    if hw is None:
        h, w, bits, ctype, lace = png_header(paths[0])
    else:
        h, w = hw
    buf, offsets, nbytes, st_host, cap = pack_depth_pngs(paths, h, w, n_threads)
    src = torch.from_numpy(buf[:max(cap, 16)]).to(device, non_blocking=False)
    off_d = torch.from_numpy(offsets).to(device)
    nb_d = torch.from_numpy(np.where(st_host == 0, nbytes, 0)).to(device)          # a frame the packer declined: an empty stream
    raw, status = engine.inflate_blocks_device(src, off_d, nb_d, h * (2 * w + 1))
    frames = engine.png_unfilter_device(raw, h, w, status, out)
    bad = np.nonzero(status.cpu().numpy())[0]
    if len(bad):
        host = read_depth_frames([paths[int(k)] for k in bad], n_threads, general_reader=general_reader)
        frames[torch.from_numpy(bad).to(device)] = torch.from_numpy(host.view(np.int16)).to(device)
    return frames[:F]
