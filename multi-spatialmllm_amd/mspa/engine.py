"""PyTorch-ROCm front end of the geometry kernels.

torch is plumbing here: it owns device memory and the stream; every computation is a call into
libmspa.so through ``data_ptr()``.  Host-side matrix preparation (``np.linalg.inv``, ``A @ E``)
uses the very NumPy routines the reference uses (IH:57, IH:113-124, OPS:313) so that what reaches
the kernels is bit-identical to what the reference would multiply with.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

_AFFINE_ROW = np.array([0.0, 0.0, 0.0, 1.0])


def _require(condition: bool, what: str):
    """Argument checks in front of raw device pointers: a real exception, not an ``assert`` that ``python -O`` drops."""
    if not condition:
        raise ValueError(f"mspa.engine: requirement not met: {what}")


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("mspa.engine needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback -- the CPU restatement under oracle/ is test-only")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    _require(t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only")
    return t.data_ptr()


def check_affine(name: str, m: np.ndarray):
    """The kernels take affine matrices with finite entries (include/mspa.h); the reference drops
    frames with non-finite poses before any projection (IH:409-418)."""
    m = np.asarray(m, dtype=np.float64)
    if m.shape != (4, 4) or not np.all(np.isfinite(m)):
        raise ValueError(f"{name}: expected a finite 4x4 matrix")
    if not np.array_equal(m[3], _AFFINE_ROW):
        raise ValueError(f"{name}: last row must be exactly [0, 0, 0, 1], got {m[3]}")
    return m


def _check_affine_stack(name: str, m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    if m.ndim != 3 or m.shape[1:] != (4, 4) or not np.all(np.isfinite(m)):
        raise ValueError(f"{name}: expected finite 4x4 matrices")
    if not np.array_equal(m[:, 3, :], np.broadcast_to(_AFFINE_ROW, (m.shape[0], 4))):
        bad = int(np.nonzero(~(m[:, 3, :] == _AFFINE_ROW).all(axis=1))[0][0])
        raise ValueError(f"{name}[{bad}]: last row must be exactly [0, 0, 0, 1], got {m[bad, 3]}")
    return m


def frame_matrices(K: np.ndarray, A: Optional[np.ndarray], E_list: Sequence[np.ndarray]) -> np.ndarray:
    """[F, 8, 16] float64 frame records for mspa_pair_reproject (slot order of include/mspa.h).  All frames in one batched
    ``A @ E`` / ``np.linalg.inv`` call: NumPy runs the same per-matrix dgemm / dgesv as for a single 4x4, so every entry is
    bit-identical to the reference's one-frame-at-a-time expressions (tests/test_host_cpu.py pins that)."""
    K = check_affine("K", K)
    A = np.eye(4) if A is None else check_affine("A", A)
    Kinv = check_affine("inv(K)", np.linalg.inv(K))                       # OPS:313
    F = len(E_list)
    out = np.empty((F, _lib.FRAME_MATS, 16), dtype=np.float64)
    if F == 0:
        return out
    E = _check_affine_stack("E", np.stack([np.asarray(e, dtype=np.float64) for e in E_list]))
    AE = A @ E                                                             # IH:113-124
    Einv_al = _check_affine_stack("inv(A@E)", np.linalg.inv(AE))          # IH:57
    out[:, _lib.MAT_KINV] = Kinv.reshape(16)
    out[:, _lib.MAT_E] = E.reshape(F, 16)
    out[:, _lib.MAT_A] = A.reshape(16)
    out[:, _lib.MAT_EINV_ALIGNED] = Einv_al.reshape(F, 16)
    out[:, _lib.MAT_K] = K.reshape(16)
    # composed products for MSPA_PAIR_FAST (any float64 evaluation order will do: lanes near a
    # decision boundary are re-evaluated with the exact chain inside the kernel)
    out[:, _lib.MAT_UNPROJ] = (AE @ Kinv).reshape(F, 16)
    out[:, _lib.MAT_REPROJ] = (K @ Einv_al).reshape(F, 16)
    out[:, _lib.MAT_BOUNDS] = frame_bounds(Kinv, E, A, Einv_al, K)
    return out


def frame_bounds(Kinv: np.ndarray, E: np.ndarray, A: np.ndarray, Einv_al: np.ndarray, K: np.ndarray) -> np.ndarray:
    """[F, 16] float64: slot MSPA_MAT_BOUNDS of the frame records -- the coefficients from which the fast kernels derive
    their guard band as a BOUND (include/mspa.h; DESIGN.md section 4, "guard").  With |.| entrywise:
      Uabs = |A| |E| |inv(K)|        (frame as frame 1: pixel * depth -> aligned world)
      Nabs = |K| |inv(A E)|          (frame as frame 2: aligned world -> homogeneous image coordinates)
    Every float64 evaluation order of K2 inv(A E2) A E1 inv(K1) p -- the reference's five products, the composed 3x4 of the
    fast kernels -- stays within GUARD_C * 2^-53 * (Nabs Uabs |p|) of the exact value; the kernels evaluate an upper
    estimate of that product of magnitudes per tile from these ten numbers.  Same values as mspa_frame_bounds_host up to
    the summation order of the magnitudes (tests/test_host_cpu.py)."""
    F = E.shape[0]
    Ua = (np.abs(A) @ np.abs(E)) @ np.abs(Kinv)                           # [F, 4, 4]
    Na = np.abs(K) @ np.abs(Einv_al)
    c = _lib.GUARD_C * 2.0 ** -53
    b = np.zeros((F, 16), dtype=np.float64)
    b[:, 0:4] = Ua[:, :3, :].max(axis=1)
    b[:, 3] *= 1000.0
    b[:, 4:7] = c * ((Na[:, :3, 0] + Na[:, :3, 1]) + Na[:, :3, 2])
    b[:, 8:11] = c * 1000.0 * Na[:, :3, 3]
    return b


def fast_path_ok(K: np.ndarray) -> bool:
    """MSPA_PAIR_FAST reads the camera-2 depth off the third image row: K[2] must be 0 0 1 0."""
    return bool(np.array_equal(np.asarray(K, dtype=np.float64)[2], np.array([0.0, 0.0, 1.0, 0.0])))


def camera_matrices(K: np.ndarray, E_aligned_list: Sequence[np.ndarray]) -> np.ndarray:
    """[n, 3, 16] float64 records for mspa_vertex_visibility: inv(E_aligned), K (one batched inverse, see frame_matrices) and
    the guard-bound coefficients of the composed kernels (slot MSPA_CAM_BOUNDS of include/mspa.h: magnitudes of
    |K| |inv(E_aligned)|; the same numbers mspa_camera_bounds_host computes)."""
    K = check_affine("K", K)
    n = len(E_aligned_list)
    out = np.zeros((n, _lib.CAM_MATS, 16), dtype=np.float64)
    if n == 0:
        return out
    E = _check_affine_stack("E_aligned", np.stack([np.asarray(e, dtype=np.float64) for e in E_aligned_list]))
    Einv = _check_affine_stack("inv(E_aligned)", np.linalg.inv(E))                         # IH:57
    out[:, _lib.CAM_EINV] = Einv.reshape(n, 16)
    out[:, _lib.CAM_K] = K.reshape(16)
    Na = np.abs(K) @ np.abs(Einv)
    c = _lib.GUARD_C * 2.0 ** -53 * 1000.0
    nr = (Na[:, :3, 0] + Na[:, :3, 1]) + Na[:, :3, 2]
    out[:, _lib.CAM_BOUNDS, 0] = c * (nr[:, 0] + nr[:, 1])
    out[:, _lib.CAM_BOUNDS, 1] = c * nr[:, 2]
    out[:, _lib.CAM_BOUNDS, 2] = c * (Na[:, 0, 3] + Na[:, 1, 3])
    out[:, _lib.CAM_BOUNDS, 3] = c * Na[:, 2, 3]
    return out


def depth_to_device(depth: np.ndarray, device="cuda") -> torch.Tensor:
    """uint16 depth frames -> device int16 tensor holding the same bits (torch has few uint16 ops)."""
    d = np.ascontiguousarray(depth, dtype=np.uint16)
    return torch.from_numpy(d.view(np.int16)).to(device)


def gather_blocks_host(blocks, dst: np.ndarray, n_threads: int = 4) -> None:
    """``dst[k] = blocks[k]`` for equally shaped, C-contiguous host arrays of ``dst``'s dtype, copied by ``n_threads``
    native threads (mspa_gather_blocks_host): the staging of a scene's depth frames into pinned memory."""
    n = len(blocks)
    if n == 0:
        return
    if dst.shape[0] < n or not dst.flags.c_contiguous:
        raise ValueError("gather_blocks_host: destination too small or not contiguous")
    block_bytes = int(dst[0].nbytes)
    ptrs = (ctypes.c_void_p * n)()
    for k, b in enumerate(blocks):
        if b.nbytes != block_bytes or b.dtype.itemsize != dst.dtype.itemsize or not b.flags.c_contiguous:
            raise ValueError("gather_blocks_host: block %d is not a contiguous array of the destination's frame size" % k)
        ptrs[k] = b.ctypes.data
    _lib.check(_lib.load().mspa_gather_blocks_host(ptrs, n, block_bytes, dst.ctypes.data, int(n_threads)))


def inflate_blocks_device(src: torch.Tensor, offsets: torch.Tensor, nbytes: torch.Tensor, block_bytes: int,
                          out: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None):
    """``len(offsets)`` zlib streams resident on the device -- stream k = ``src[offsets[k] : offsets[k] + nbytes[k]]`` (uint8;
    offsets multiples of 8) -- inflated by one wave each into ``out[k, :block_bytes]`` (mspa_inflate_blocks_device; replaces
    the per-frame zlib.decompress / cv2.imread of the reference, extract_posed_images.py:49-57, info_handler.py:149-155).
    Returns (out [n, pitch] uint8, status [n] int32): status 0 = inflated to exactly ``block_bytes`` with a matching
    Adler-32; anything else = decode that block on the host.  Only enqueues; read ``status`` after a synchronisation."""
    _require_gpu()
    n = int(offsets.shape[0])
    _require(src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous() and src.dim() == 1, "src: a flat uint8 device tensor")
    _require(offsets.is_cuda and offsets.dtype == torch.int64 and nbytes.is_cuda and nbytes.dtype == torch.int64
             and nbytes.shape[0] == n, "offsets / nbytes: int64 device tensors of one length")
    pitch = (int(block_bytes) + 255) // 256 * 256
    if out is None:
        out = torch.empty((n, pitch), dtype=torch.uint8, device=src.device)
    _require(out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.dim() == 2 and out.shape[0] >= n
             and out.shape[1] >= block_bytes and out.shape[1] % 256 == 0, "out: [n, pitch] uint8 with a pitch that is a multiple of 256")
    if status is None:
        status = torch.empty((n,), dtype=torch.int32, device=src.device)
    work = torch.empty((max(n, 1),), dtype=torch.int32, device=src.device)
    _lib.check(_lib.load().mspa_inflate_blocks_device(_ptr(src), _ptr(offsets), _ptr(nbytes), int(src.numel()), n, int(block_bytes),
                                                      _ptr(out), int(out.shape[1]), _ptr(status), _ptr(work), _stream_ptr()))
    return out, status


def png_unfilter_device(raw: torch.Tensor, h: int, w: int, status: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``raw[k, : h * (2 w + 1)]`` -- the inflated scanlines of n 16-bit greyscale PNG images -- to ``out[k, h, w]`` (int16
    storage of the uint16 depth values, like every depth tensor here): the five PNG row filters undone on the device
    (mspa_png_unfilter_device).  Images with ``status[k] != 0`` are skipped; a filter byte > 4 sets status 3."""
    _require_gpu()
    n = int(raw.shape[0])
    _require(raw.is_cuda and raw.dtype == torch.uint8 and raw.dim() == 2 and raw.is_contiguous()
             and raw.shape[1] >= h * (2 * w + 1), "raw: [n, pitch] uint8 with pitch >= h * (2 w + 1)")
    _require(status.is_cuda and status.dtype == torch.int32 and status.shape[0] >= n, "status: int32 [n] on the device")
    if out is None:
        out = torch.empty((n, h, w), dtype=torch.int16, device=raw.device)
    _require(out.is_cuda and out.dtype == torch.int16 and out.is_contiguous() and tuple(out.shape[-2:]) == (h, w)
             and out.shape[0] >= n, "out: [n, h, w] int16 on the device")
    _lib.check(_lib.load().mspa_png_unfilter_device(_ptr(raw), int(raw.shape[1]), n, int(h), int(w), _ptr(out), _ptr(status),
                                                    _stream_ptr()))
    return out


def _require_pinhole(mats: torch.Tensor):
    """MSPA_PAIR_FAST reads the camera-2 depth off the third image row: K's third row must be 0 0 1 0 in EVERY frame record
    (include/mspa.h).  The records live on the device, so the check is one read-back of all frames' rows the first time a
    table OBJECT is seen; the verdict rides on that tensor object (with its in-place version counter), so later launches on
    the same table enqueue without touching the host -- a per-call read-back would synchronise the stream in the hot enqueue
    path -- and a new table can never inherit another one's verdict (round 3 keyed a global cache by storage address, which
    the caching allocator hands to the next scene's table: ADVICE round 3).  ``SceneOnDevice`` and the bench keep one table
    object per scene / run; a caller that re-wraps the storage in a fresh tensor per call pays the read-back each time."""
    cached = getattr(mats, "_mspa_pinhole", None)
    if cached is None or cached[0] != mats._version:
        rows = mats[:, _lib.MAT_K, 8:12]
        want = torch.tensor([0.0, 0.0, 1.0, 0.0], dtype=torch.float64, device=mats.device)
        ok = bool((rows == want).all().item()) if mats.shape[0] else True
        cached = (mats._version, ok)
        try:
            mats._mspa_pinhole = cached
        except AttributeError:                            # a tensor subclass without a __dict__: check every time
            pass
    _require(cached[1], "MSPA_PAIR_FAST needs a pinhole K (third row 0 0 1 0) in every frame record; use flags=0 for this camera")


PAIR_OUTPUTS = ("vis_bits", "vis_u8", "valid_u8", "pix_i16", "xyz_f32", "rgba", "xyz_f64", "uv_f64",
                "depth_f64", "counts")


def alloc_pair_outputs(n_pairs: int, image_hw: Tuple[int, int], outputs: Iterable[str], device="cuda"):
    H, W = image_hw
    P = H * W
    shapes = {
        "vis_bits": ((n_pairs, (P + 63) // 64), torch.int64),
        "vis_u8": ((n_pairs, P), torch.uint8),
        "valid_u8": ((n_pairs, P), torch.uint8),
        "pix_i16": ((n_pairs, P, 2), torch.int16),
        "xyz_f32": ((n_pairs, P, 3), torch.float32),
        "rgba": ((n_pairs, P), torch.int32),
        "xyz_f64": ((n_pairs, P, 3), torch.float64),
        "uv_f64": ((n_pairs, P, 2), torch.float64),
        "depth_f64": ((n_pairs, P), torch.float64),
        "counts": ((n_pairs, 2), torch.int32),
    }
    out = {}
    for name in outputs:
        if name not in shapes:
            raise ValueError(f"unknown pair output {name!r}; choose from {PAIR_OUTPUTS}")
        shape, dtype = shapes[name]
        out[name] = torch.empty(shape, dtype=dtype, device=device)
    return out


def pair_reproject(depth: torch.Tensor, mats: torch.Tensor, pairs: torch.Tensor, image_hw: Tuple[int, int],
                   out: Dict[str, torch.Tensor], rgb: Optional[torch.Tensor] = None, flags: int = 0):
    """Enqueue K3 on the current stream.  depth [F,DH,DW] int16(bits of uint16), mats [F,7,16] f64,
    pairs [B,2] int32, rgb [F,H,W,3] uint8 (only for out['rgba']).  ``out`` comes from
    alloc_pair_outputs and is filled in place."""
    _require_gpu()
    lib = _lib.load()
    _require(depth.dtype in (torch.int16, torch.uint16) and depth.dim() == 3, "depth.dtype in (torch.int16, torch.uint16) and depth.dim() == 3")
    _require(mats.dtype == torch.float64 and mats.shape[1:] == (_lib.FRAME_MATS, 16), "mats.dtype == torch.float64 and mats.shape[1:] == (_lib.FRAME_MATS, 16)")
    _require(pairs.dtype == torch.int32 and pairs.dim() == 2 and pairs.shape[1] == 2, "pairs.dtype == torch.int32 and pairs.dim() == 2 and pairs.shape[1] == 2")
    F, DH, DW = depth.shape
    _require(mats.shape[0] == F, "mats.shape[0] == F")
    if flags & _lib.PAIR_FAST:
        _require_pinhole(mats)
    H, W = image_hw
    if rgb is not None:
        _require(rgb.dtype == torch.uint8 and tuple(rgb.shape) == (F, H, W, 3), "rgb.dtype == torch.uint8 and tuple(rgb.shape) == (F, H, W, 3)")
    g = lambda k: _ptr(out.get(k))
    _lib.check(lib.mspa_pair_reproject(
        _ptr(depth), _ptr(rgb), _ptr(mats), F, _ptr(pairs), pairs.shape[0], DH, DW, H, W,
        g("vis_bits"), g("vis_u8"), g("valid_u8"), g("pix_i16"), g("xyz_f32"), g("rgba"),
        g("xyz_f64"), g("uv_f64"), g("depth_f64"), g("counts"), flags, _stream_ptr()))
    return out


def corr_tiles(image_hw: Tuple[int, int]) -> Tuple[int, int]:
    """(stripes across, bands down) of the 64 x 48-pixel tiles the compacted correspondence output is segmented by."""
    H, W = image_hw
    return (W + _lib.CORR_TILE_W - 1) // _lib.CORR_TILE_W, (H + _lib.CORR_TILE_H - 1) // _lib.CORR_TILE_H


def alloc_pair_correspondences(n_pairs: int, image_hw: Tuple[int, int], device="cuda", counts: bool = True):
    """Caller-owned outputs of ``pair_correspondences`` (include/mspa.h, mspa_pair_correspondences)."""
    H, W = image_hw
    ns, nb = corr_tiles(image_hw)
    out = {"vis_bits": torch.empty((n_pairs, (H * W + 63) // 64), dtype=torch.int64, device=device),
           "cpix": torch.empty((n_pairs, ns * nb, _lib.CORR_TILE_CAP, 2), dtype=torch.int16, device=device),
           "tile_counts": torch.empty((n_pairs, ns * nb), dtype=torch.int32, device=device)}
    if counts:
        out["counts"] = torch.empty((n_pairs, 2), dtype=torch.int32, device=device)
    return out


def pair_correspondences(depth: torch.Tensor, mats: torch.Tensor, pairs: torch.Tensor, image_hw: Tuple[int, int],
                         out: Optional[Dict[str, torch.Tensor]] = None, flags: int = _lib.PAIR_FAST,
                         workspace: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Enqueue K3 with the compacted correspondence output: the visibility bitset plus, per 64 x 48 tile of frame 1, the
    frame-2 pixel (xi, yi) of its VISIBLE pixels in (row, column) order.  One fused kernel on whole-tile shapes with
    MSPA_PAIR_FAST (640x480); other shapes / the reference-order mode go through a dense table in ``workspace`` (allocated
    here when not given).  ``out`` from ``alloc_pair_correspondences``."""
    _require_gpu()
    lib = _lib.load()
    _require(depth.dtype in (torch.int16, torch.uint16) and depth.dim() == 3, "depth: [F, DH, DW] int16 / uint16")
    _require(mats.dtype == torch.float64 and mats.shape[1:] == (_lib.FRAME_MATS, 16), "mats: float64 [F, 8, 16]")
    _require(pairs.dtype == torch.int32 and pairs.dim() == 2 and pairs.shape[1] == 2, "pairs: int32 [B, 2]")
    F, DH, DW = depth.shape
    _require(mats.shape[0] == F, "mats.shape[0] == F")
    if flags & _lib.PAIR_FAST:
        _require_pinhole(mats)
    H, W = image_hw
    n = pairs.shape[0]
    if out is None:
        out = alloc_pair_correspondences(n, image_hw, depth.device)
    ns, nb = corr_tiles(image_hw)
    _require(tuple(out["cpix"].shape) == (n, ns * nb, _lib.CORR_TILE_CAP, 2) and out["cpix"].dtype == torch.int16,
             "out['cpix']: int16 [n_pairs, n_tiles, 3072, 2]")
    _require(tuple(out["tile_counts"].shape) == (n, ns * nb) and out["tile_counts"].dtype == torch.int32,
             "out['tile_counts']: int32 [n_pairs, n_tiles]")
    _require(tuple(out["vis_bits"].shape) == (n, (H * W + 63) // 64) and out["vis_bits"].dtype == torch.int64,
             "out['vis_bits']: int64 [n_pairs, ceil(P / 64)]")
    need = int(lib.mspa_pair_correspondences_workspace_bytes(n, DH, DW, H, W, flags))
    _require(need >= 0, "image size within [2, 32767]")
    if need == 0 and depth.data_ptr() & 3:
        # a depth VIEW at an odd 2-byte offset: the fused kernel's 4-byte LDS-DMA cannot take it and the C entry point goes
        # through the dense table, which needs a workspace although the size query (it cannot see the pointer) said 0
        need = n * H * W * 4
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty((need + 3) // 4, dtype=torch.int32, device=depth.device)
    _lib.check(lib.mspa_pair_correspondences(
        _ptr(depth), _ptr(mats), F, _ptr(pairs), n, DH, DW, H, W, _ptr(out["vis_bits"]), _ptr(out["cpix"]),
        _ptr(out["tile_counts"]), _ptr(out.get("counts")), _ptr(workspace) if need else None,
        workspace.numel() * workspace.element_size() if need else 0, flags, _stream_ptr()))
    return out


def correspondences_rowmajor(out: Dict[str, torch.Tensor], image_hw: Tuple[int, int], pair: int):
    """One pair's compacted correspondences re-ordered to ``np.nonzero(visible)`` order (torch indexing as plumbing, for
    consumers and tests that want the flat view): returns (pixel index i = y * W + x in frame 1 [n_vis] int64, xi [n_vis],
    yi [n_vis] int16 in frame 2), all on the device."""
    H, W = image_hw
    ns, nb = corr_tiles(image_hw)
    bits = out["vis_bits"][pair]
    P = H * W
    shifts = torch.arange(64, device=bits.device, dtype=torch.int64)
    vis = ((bits.unsqueeze(1) >> shifts) & 1).reshape(-1)[:P].reshape(H, W).bool()
    # rank of every visible pixel inside its tile, in (row, column) order of the tile
    pad = torch.zeros((nb * _lib.CORR_TILE_H, ns * _lib.CORR_TILE_W), dtype=torch.bool, device=bits.device)
    pad[:H, :W] = vis
    tiles = pad.reshape(nb, _lib.CORR_TILE_H, ns, _lib.CORR_TILE_W).permute(0, 2, 1, 3).reshape(nb * ns, -1)
    rank = torch.cumsum(tiles.to(torch.int32), dim=1) - 1
    rank_img = rank.reshape(nb, ns, _lib.CORR_TILE_H, _lib.CORR_TILE_W).permute(0, 2, 1, 3).reshape(nb * _lib.CORR_TILE_H, -1)[:H, :W]
    yy, xx = torch.nonzero(vis, as_tuple=True)
    t = (yy // _lib.CORR_TILE_H) * ns + (xx // _lib.CORR_TILE_W)
    e = out["cpix"][pair][t, rank_img[yy, xx].long()]
    return yy * W + xx, e[:, 0], e[:, 1]


def vertex_visibility(xyz: torch.Tensor, cam_mats: torch.Tensor, depth: torch.Tensor, image_hw: Tuple[int, int],
                      want: Iterable[str] = ("bits", "count"), homogeneous: bool = False,
                      depth_scale: float = 0.001) -> Dict[str, torch.Tensor]:
    """Enqueue K1.  xyz [N,3] or [N,C>=3] float64 rows (or a [3,N] SoA tensor with soa=True layout
    given as xyz.t()); cam_mats [I,3,16]; depth [I,DH,DW].  Returns the requested outputs.
    ``homogeneous``: the rows are general homogeneous points (x, y, z, w) (``project_points`` takes any [N, 4], IH:46-72);
    ``depth_scale``: the handler's ``depth_value_scale`` (IH:76, IH:368)."""
    _require_gpu()
    lib = _lib.load()
    _require(xyz.dtype == torch.float64 and xyz.dim() == 2 and xyz.is_cuda, "xyz.dtype == torch.float64 and xyz.dim() == 2 and xyz.is_cuda")
    n = xyz.shape[0]
    ps, cs = xyz.stride(0), xyz.stride(1)
    _require(xyz.shape[1] >= (4 if homogeneous else 3) and ps > 0 and cs > 0, "xyz: [N, >= 3] rows ([N, >= 4] when homogeneous)")
    I, DH, DW = depth.shape
    _require(cam_mats.dtype == torch.float64 and tuple(cam_mats.shape) == (I, _lib.CAM_MATS, 16), "cam_mats: float64 [n_images, 3, 16] (engine.camera_matrices)")
    H, W = image_hw
    dev = xyz.device
    out: Dict[str, torch.Tensor] = {}
    want = tuple(want)
    if "bits" in want:
        out["bits"] = torch.empty((I, (n + 63) // 64), dtype=torch.int64, device=dev)
    if "mask" in want:
        out["mask"] = torch.empty((I, n), dtype=torch.uint8, device=dev)
    if "uv" in want:
        out["uv"] = torch.empty((I, n, 2), dtype=torch.float64, device=dev)
    if "depth" in want:
        out["depth"] = torch.empty((I, n), dtype=torch.float64, device=dev)
    if "count" in want:
        out["count"] = torch.empty((I,), dtype=torch.int32, device=dev)
    _require(depth.dtype in (torch.int16, torch.uint16), "depth.dtype in (torch.int16, torch.uint16)")
    _lib.check(lib.mspa_vertex_visibility_ex(
        xyz.data_ptr(), n, ps, cs, 1 if homogeneous else 0, _ptr(cam_mats), I, _ptr(depth), DH, DW, H, W, float(depth_scale),
        _ptr(out.get("bits")), _ptr(out.get("mask")), _ptr(out.get("uv")), _ptr(out.get("depth")),
        _ptr(out.get("count")), _stream_ptr()))
    return out


def all_pairs(n: int, device="cuda") -> torch.Tensor:
    """(i, j), i < j, in the row-major order of the reference's nested loops (CFR:176-178)."""
    return torch.triu_indices(n, n, offset=1, device=device).t().contiguous().to(torch.int32)


def pair_overlap(bits: torch.Tensor, pairs: torch.Tensor, want_counts: bool = False):
    """Enqueue K2 on K1's bitsets.  Returns overlap [n_pairs] f64 (+ inter, union int32)."""
    _require_gpu()
    lib = _lib.load()
    _require(bits.dtype == torch.int64 and bits.dim() == 2 and bits.is_contiguous(), "bits.dtype == torch.int64 and bits.dim() == 2 and bits.is_contiguous()")
    _require(pairs.dtype == torch.int32 and pairs.dim() == 2 and pairs.shape[1] == 2 and pairs.is_contiguous(), "pairs.dtype == torch.int32 and pairs.dim() == 2 and pairs.shape[1] == 2 and pairs.is_conti")
    n_pairs = pairs.shape[0]
    overlap = torch.empty((n_pairs,), dtype=torch.float64, device=bits.device)
    inter = torch.empty((n_pairs,), dtype=torch.int32, device=bits.device) if want_counts else None
    uni = torch.empty((n_pairs,), dtype=torch.int32, device=bits.device) if want_counts else None
    _lib.check(lib.mspa_pair_overlap(bits.data_ptr(), bits.shape[0], bits.shape[1], pairs.data_ptr(), n_pairs,
                                     overlap.data_ptr(), _ptr(inter), _ptr(uni), _stream_ptr()))
    return (overlap, inter, uni) if want_counts else overlap


def _overlap_workspace(n_a: int, n_b: int, n_words: int, device) -> torch.Tensor:
    nbytes = int(_lib.load().mspa_overlap_workspace_bytes(n_a, n_b, n_words))
    return torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.int32, device=device)


def scene_overlap(bits: torch.Tensor, want_counts: bool = False):
    """Enqueue the tiled K2 over ALL pairs i < j of a scene's bitsets, in ``all_pairs`` order (CFR:176-178).
    Returns overlap [F(F-1)/2] f64 (+ inter, union int32) -- identical to ``pair_overlap(bits, all_pairs(F))``."""
    _require_gpu()
    lib = _lib.load()
    _require(bits.dtype == torch.int64 and bits.dim() == 2 and bits.is_contiguous() and bits.is_cuda,
             "bits: contiguous device int64 [F, n_words]")
    F, n_words = bits.shape
    n_pairs = F * (F - 1) // 2
    dev = bits.device
    overlap = torch.empty((n_pairs,), dtype=torch.float64, device=dev)
    inter = torch.empty((n_pairs,), dtype=torch.int32, device=dev) if want_counts else None
    uni = torch.empty((n_pairs,), dtype=torch.int32, device=dev) if want_counts else None
    if n_pairs and n_words:
        ws = _overlap_workspace(F, F, n_words, dev)
        _lib.check(lib.mspa_scene_overlap(bits.data_ptr(), F, n_words, ws.data_ptr(), ws.numel() * 4, overlap.data_ptr(),
                                          _ptr(inter), _ptr(uni), _stream_ptr()))
    return (overlap, inter, uni) if want_counts else overlap


def overlap_matrix(bits_a: torch.Tensor, bits_b: torch.Tensor) -> torch.Tensor:
    """Enqueue the tiled K2 on a rectangle: |a_i & b_j| for every row pair, [n_a, n_b] int32 (object visibility:
    objects x images, compute_object_visibility.py:72-152)."""
    _require_gpu()
    lib = _lib.load()
    for t in (bits_a, bits_b):
        _require(t.dtype == torch.int64 and t.dim() == 2 and t.is_contiguous() and t.is_cuda,
                 "bits: contiguous device int64 [rows, n_words]")
    _require(bits_a.shape[1] == bits_b.shape[1], "both bitset tables must have the same word count")
    n_a, n_words = bits_a.shape
    n_b = bits_b.shape[0]
    out = torch.empty((n_a, n_b), dtype=torch.int32, device=bits_a.device)
    if n_a and n_b and n_words:
        ws = _overlap_workspace(n_a, n_b, n_words, bits_a.device)
        _lib.check(lib.mspa_overlap_matrix(bits_a.data_ptr(), n_a, bits_b.data_ptr(), n_b, n_words, ws.data_ptr(),
                                           ws.numel() * 4, out.data_ptr(), _stream_ptr()))
    return out


def bitset_csr(bits: torch.Tensor):
    """K9: rows of a bit matrix -> CSR of their set bits.  bits [R, n_words] int64 (device) -> (offsets [R+1] int64,
    indices [nnz] int32), both on the device; row r's set-bit positions, ascending, are indices[offsets[r]:offsets[r+1]]."""
    _require_gpu()
    lib = _lib.load()
    _require(bits.dtype == torch.int64 and bits.dim() == 2 and bits.is_contiguous() and bits.is_cuda,
             "bits: contiguous device int64 [rows, n_words]")
    R, n_words = bits.shape
    dev = bits.device
    if R == 0 or n_words == 0:
        return torch.zeros((R + 1,), dtype=torch.int64, device=dev), torch.zeros((0,), dtype=torch.int32, device=dev)
    counts = torch.empty((R * n_words,), dtype=torch.int32, device=dev)
    _lib.check(lib.mspa_bits_popcount(bits.data_ptr(), R * n_words, counts.data_ptr(), _stream_ptr()))
    incl = torch.cumsum(counts, dim=0, dtype=torch.int64)              # torch as plumbing: one prefix sum
    word_offsets = (incl - counts).contiguous()
    total = int(incl[-1].item())
    indices = torch.empty((total,), dtype=torch.int32, device=dev)
    if total:
        _lib.check(lib.mspa_bits_expand(bits.data_ptr(), R, n_words, word_offsets.data_ptr(), indices.data_ptr(), _stream_ptr()))
    offsets = torch.cat([word_offsets[::n_words], incl[-1:]])
    return offsets, indices


def format_lists_device(offsets: torch.Tensor, values: torch.Tensor, tokens: Optional[Sequence[bytes]] = None):
    """K10: the JSON text of every list of a CSR table, written on the device (``json.dumps`` form: "[1, 2, 3]", "[]";
    make_visibility_info.py:38-73).  ``offsets`` [n + 1] int64, ``values`` [nnz] int32 (device).  ``tokens`` None: the items are
    integers; else item e is the text ``tokens[values[e]]`` (bytes, e.g. an already quoted image id).  Returns (text [bytes] uint8,
    text_offsets [n + 1] int32) on the device: arrow's string layout."""
    _require_gpu()
    lib = _lib.load()
    _require(offsets.dtype == torch.int64 and offsets.dim() == 1 and offsets.is_cuda and offsets.is_contiguous() and offsets.numel() >= 1,
             "offsets: contiguous device int64 [n_lists + 1]")
    _require(values.dtype == torch.int32 and values.dim() == 1 and values.is_cuda and values.is_contiguous(), "values: contiguous device int32 [nnz]")
    dev = offsets.device
    n, nnz = offsets.numel() - 1, values.numel()
    tok = tok_off = None
    n_tokens = 0
    if tokens is not None:
        n_tokens = len(tokens)
        lens = np.array([len(t) for t in tokens], dtype=np.int64)
        tok_off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
        tok = torch.from_numpy(np.frombuffer(b"".join(tokens) or b"\0", dtype=np.uint8).copy()).to(dev)
    cost = torch.empty((max(nnz, 1),), dtype=torch.int32, device=dev)
    bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.check(lib.mspa_format_list_costs_device(_ptr(values) if nnz else None, nnz, _ptr(tok_off), n_tokens, _ptr(cost), _ptr(bad), _stream_ptr()))
    T = torch.zeros((nnz + 1,), dtype=torch.int64, device=dev)                   # torch as plumbing: two prefix sums
    if nnz:
        torch.cumsum(cost[:nnz], dim=0, dtype=torch.int64, out=T[1:])
    NE = torch.zeros((n + 1,), dtype=torch.int64, device=dev)
    if n:
        torch.cumsum((offsets[1:] > offsets[:-1]).to(torch.int64), dim=0, out=NE[1:])
    head = torch.stack([2 * n + T[-1] - 2 * NE[-1], bad[0].to(torch.int64)]).cpu()     # one small read-back: the size to allocate
    total = int(head[0])
    _require(int(head[1]) == 0, "format_lists_device: token id out of range")
    _require(total <= 0x7fffffff, "format_lists_device: more than 2 GiB of text")
    text = torch.empty((max(total, 1),), dtype=torch.uint8, device=dev)
    text_offsets = torch.empty((n + 1,), dtype=torch.int32, device=dev)
    _lib.check(lib.mspa_format_lists_device(_ptr(offsets), _ptr(values) if nnz else None, n, nnz, _ptr(T), _ptr(NE), _ptr(tok), _ptr(tok_off),
                                            _ptr(text), total, _ptr(text_offsets), _stream_ptr()))
    return text[:total], text_offsets


def bits_transpose(bits: torch.Tensor) -> torch.Tensor:
    """K9: [R, n_words] int64 bit matrix -> its transpose [n_words * 64, ceil(R / 64)] (padding bits zero)."""
    _require_gpu()
    lib = _lib.load()
    _require(bits.dtype == torch.int64 and bits.dim() == 2 and bits.is_contiguous() and bits.is_cuda,
             "bits: contiguous device int64 [rows, n_words]")
    R, n_words = bits.shape
    out = torch.empty((n_words * 64, (R + 63) // 64), dtype=torch.int64, device=bits.device)
    if R and n_words:
        _lib.check(lib.mspa_bits_transpose(bits.data_ptr(), R, n_words, out.data_ptr(), _stream_ptr()))
    else:
        out.zero_()
    return out


def extract_yaw_pitch_host(E_aligned_list: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """Per-frame angles as CFR:86-100 computes them, with NumPy on the host (the reference's own libm / BLAS calls are the
    only way to be bit-identical with it).  The arctan2 / arcsin / degrees ufuncs run once over all frames (the same
    element loops a scalar call goes through); the norm stays one ``dot`` per frame, because that is what
    ``np.linalg.norm`` of a 3-vector is and a vectorised sum of squares rounds differently.  tests/test_host_cpu.py holds
    this against the oracle's literal per-frame form bit for bit."""
    n = len(E_aligned_list)
    if n == 0:
        return np.zeros(0, dtype=np.float64), np.zeros(0, dtype=np.float64)
    Z = np.ascontiguousarray(np.stack([np.asarray(E) for E in E_aligned_list])[:, :3, 2], dtype=np.float64)
    yaw = np.degrees(np.arctan2(Z[:, 1], Z[:, 0]))
    norm = np.sqrt(np.array([z.dot(z) for z in Z], dtype=np.float64))       # np.linalg.norm(z) == sqrt(z.dot(z)) for real 1-D z
    pitch = np.degrees(np.arcsin(Z[:, 2] / norm))
    return yaw, pitch


def extract_yaw_pitch(E_aligned: torch.Tensor):
    """Device form of CFR:86-100 for frames that are already resident: E_aligned [F,16] f64 -> (yaw [F], pitch [F]) in
    degrees.  Within a few ulp of ``extract_yaw_pitch_host`` (device libm is not glibc); the pair-table writers keep the
    host path so that the parquet columns are the reference's bits."""
    _require_gpu()
    lib = _lib.load()
    _require(E_aligned.dtype == torch.float64 and E_aligned.dim() == 2 and E_aligned.shape[1] == 16, "E_aligned: float64 [F, 16]")
    F = E_aligned.shape[0]
    yaw = torch.empty((F,), dtype=torch.float64, device=E_aligned.device)
    pitch = torch.empty((F,), dtype=torch.float64, device=E_aligned.device)
    _lib.check(lib.mspa_extract_yaw_pitch(_ptr(E_aligned), F, _ptr(yaw), _ptr(pitch), _stream_ptr()))
    return yaw, pitch


def pair_pose(E_aligned: torch.Tensor, Einv_aligned: torch.Tensor, yaw: torch.Tensor, pitch: torch.Tensor,
              pairs: torch.Tensor) -> torch.Tensor:
    """Enqueue K4.  E_aligned / Einv_aligned [F,16] f64, yaw / pitch [F] f64, pairs [n,2] i32 ->
    [n,6] f64: distance, dyaw, dpitch, translation of inv(E_i) @ E_j."""
    _require_gpu()
    lib = _lib.load()
    F = E_aligned.shape[0]
    for t in (E_aligned, Einv_aligned):
        _require(t.dtype == torch.float64 and tuple(t.shape) == (F, 16), "t.dtype == torch.float64 and tuple(t.shape) == (F, 16)")
    _require(yaw.dtype == torch.float64 and pitch.dtype == torch.float64 and pairs.dtype == torch.int32, "yaw.dtype == torch.float64 and pitch.dtype == torch.float64 and pairs.dtype == torch.int32")
    out = torch.empty((pairs.shape[0], 6), dtype=torch.float64, device=E_aligned.device)
    _lib.check(lib.mspa_pair_pose(_ptr(E_aligned), _ptr(Einv_aligned), _ptr(yaw), _ptr(pitch), F, _ptr(pairs),
                                  pairs.shape[0], _ptr(out), _stream_ptr()))
    return out


def track_to_world(tracks_xyz: torch.Tensor, c2w: Optional[torch.Tensor], fx_fy_cx_cy: Sequence[float],
                   image_hw: Tuple[int, int], want=("world", "uvn", "ok")) -> Dict[str, torch.Tensor]:
    """Enqueue K5a.  tracks_xyz [T,P,3] f64, c2w [T,16] f64 (np.linalg.inv(extrinsics_w2c) on the host)."""
    _require_gpu()
    lib = _lib.load()
    import ctypes
    T, P, _ = tracks_xyz.shape
    _require(tracks_xyz.dtype == torch.float64 and tracks_xyz.is_contiguous(), "tracks_xyz.dtype == torch.float64 and tracks_xyz.is_contiguous()")
    dev = tracks_xyz.device
    out = {}
    if "world" in want:
        out["world"] = torch.empty((T, P, 3), dtype=torch.float64, device=dev)
    if "uvn" in want:
        out["uvn"] = torch.empty((T, P, 2), dtype=torch.float64, device=dev)
    if "ok" in want:
        out["ok"] = torch.empty((T, P), dtype=torch.uint8, device=dev)
    intr = (ctypes.c_double * 4)(*[float(v) for v in fx_fy_cx_cy])
    H, W = image_hw
    _lib.check(lib.mspa_track_to_world(_ptr(tracks_xyz), _ptr(c2w), T, P, intr, H, W, _ptr(out.get("world")),
                                       _ptr(out.get("uvn")), _ptr(out.get("ok")), _stream_ptr()))
    return out


def track_displacement(world: torch.Tensor, w2c: torch.Tensor, c2w: torch.Tensor, triples: torch.Tensor,
                       obj_threshold: float = 0.01, cam_threshold: float = 0.01):
    """Enqueue K5b.  Returns ([n,5] f64: distance, dx, dy, dz (camera-1 axes), binning distance;
    [n,2] u8: point_moving, cam_moving)."""
    _require_gpu()
    lib = _lib.load()
    T, P, _ = world.shape
    _require(triples.dtype == torch.int32 and triples.dim() == 2 and triples.shape[1] == 3, "triples.dtype == torch.int32 and triples.dim() == 2 and triples.shape[1] == 3")
    n = triples.shape[0]
    out = torch.empty((n, 5), dtype=torch.float64, device=world.device)
    flags = torch.empty((n, 2), dtype=torch.uint8, device=world.device)
    _lib.check(lib.mspa_track_displacement(_ptr(world), _ptr(w2c), _ptr(c2w), T, P, _ptr(triples), n,
                                           obj_threshold, cam_threshold, _ptr(out), _ptr(flags), _stream_ptr()))
    return out, flags


def check_visibility(uv: torch.Tensor, point_depth: Optional[torch.Tensor], depth_image: Optional[torch.Tensor],
                     image_hw: Tuple[int, int], want=("visible",), depth_scale: float = 0.001) -> Dict[str, torch.Tensor]:
    """The reference's three predicates on already-projected points (IH:337-386)."""
    _require_gpu()
    lib = _lib.load()
    n = uv.shape[0]
    _require(uv.dtype == torch.float64 and uv.dim() == 2 and uv.shape[1] == 2 and uv.is_contiguous(), "uv.dtype == torch.float64 and uv.dim() == 2 and uv.shape[1] == 2 and uv.is_contiguous()")
    out = {k: torch.empty((n,), dtype=torch.uint8, device=uv.device) for k in want}
    dh, dw = (depth_image.shape[-2], depth_image.shape[-1]) if depth_image is not None else (0, 0)
    H, W = image_hw
    _lib.check(lib.mspa_check_visibility_ex(_ptr(uv), _ptr(point_depth), n, _ptr(depth_image), dh, dw, H, W, float(depth_scale),
                                            _ptr(out.get("in_bounds")), _ptr(out.get("by_depth")),
                                            _ptr(out.get("visible")), _stream_ptr()))
    return out


def select_common_point(bits: torch.Tensor, selections: torch.Tensor) -> torch.Tensor:
    """Enqueue K6a.  bits [F, n_words] int64 (K1), selections [n, 3] int32 (image1, image2, j) ->
    [n] int32 vertex index: element j of np.intersect1d of the two visible lists (-1 if out of range)."""
    _require_gpu()
    lib = _lib.load()
    _require(bits.dtype == torch.int64 and bits.dim() == 2 and bits.is_contiguous(), "bits.dtype == torch.int64 and bits.dim() == 2 and bits.is_contiguous()")
    _require(selections.dtype == torch.int32 and selections.dim() == 2 and selections.shape[1] == 3, "selections.dtype == torch.int32 and selections.dim() == 2 and selections.shape[1] == 3")
    out = torch.empty((selections.shape[0],), dtype=torch.int32, device=bits.device)
    _lib.check(lib.mspa_select_common_point(_ptr(bits), bits.shape[0], bits.shape[1], _ptr(selections.contiguous()),
                                            selections.shape[0], _ptr(out), _stream_ptr()))
    return out


def project_samples(xyz: torch.Tensor, cam_mats: torch.Tensor, depth: torch.Tensor, image_hw: Tuple[int, int],
                    samples: torch.Tensor, depth_scale: float = 0.001):
    """Enqueue K6b.  samples [n, 2] int32 (vertex, image) -> (uv [n,2] f64, depth [n] f64, visible [n] u8)."""
    _require_gpu()
    lib = _lib.load()
    _require(xyz.dtype == torch.float64 and xyz.dim() == 2 and xyz.shape[1] >= 3, "xyz.dtype == torch.float64 and xyz.dim() == 2 and xyz.shape[1] >= 3")
    _require(samples.dtype == torch.int32 and samples.dim() == 2 and samples.shape[1] == 2, "samples.dtype == torch.int32 and samples.dim() == 2 and samples.shape[1] == 2")
    I, DH, DW = depth.shape
    _require(xyz.is_cuda and xyz.stride(0) > 0 and xyz.stride(1) > 0, "xyz: device tensor with positive strides")
    _require(depth.dtype in (torch.int16, torch.uint16), "depth.dtype in (torch.int16, torch.uint16)")
    _require(cam_mats.dtype == torch.float64 and tuple(cam_mats.shape) == (I, _lib.CAM_MATS, 16), "cam_mats: float64 [n_images, 3, 16] (engine.camera_matrices)")
    n = samples.shape[0]
    dev = xyz.device
    uv = torch.empty((n, 2), dtype=torch.float64, device=dev)
    d = torch.empty((n,), dtype=torch.float64, device=dev)
    vis = torch.empty((n,), dtype=torch.uint8, device=dev)
    H, W = image_hw
    _lib.check(lib.mspa_project_samples_ex(xyz.data_ptr(), xyz.shape[0], xyz.stride(0), xyz.stride(1), _ptr(cam_mats), I,
                                           _ptr(depth), DH, DW, H, W, float(depth_scale), _ptr(samples.contiguous()), n,
                                           _ptr(uv), _ptr(d), _ptr(vis), _stream_ptr()))
    return uv, d, vis


def track_rigidity_loss(tracks_xyz: torch.Tensor, smoothing_factor: float = 0.01) -> torch.Tensor:
    """Enqueue K7: [T,P,3] f64 tracks -> [P,P] f64 accumulated thresholded distance change (OM_C:66-78)."""
    _require_gpu()
    lib = _lib.load()
    _require(tracks_xyz.dtype == torch.float64 and tracks_xyz.dim() == 3 and tracks_xyz.is_contiguous(), "tracks_xyz.dtype == torch.float64 and tracks_xyz.dim() == 3 and tracks_xyz.is_contiguous()")
    T, P, _ = tracks_xyz.shape
    out = torch.empty((P, P), dtype=torch.float64, device=tracks_xyz.device)
    _lib.check(lib.mspa_track_rigidity_loss(_ptr(tracks_xyz), T, P, float(smoothing_factor), _ptr(out), _stream_ptr()))
    return out


def object_extents(vis_bits: torch.Tensor, xyz: torch.Tensor, obj_offsets: torch.Tensor, obj_vertices: torch.Tensor):
    """Enqueue K8: per (object, image) min / max xyz of the object's visible vertices and their count.
    vis_bits [F, n_words] int64, xyz [V,3] f64, objects as CSR (int32).  Returns (lo [O,F,3], hi [O,F,3], count [O,F])."""
    _require_gpu()
    lib = _lib.load()
    _require(vis_bits.dtype == torch.int64 and vis_bits.dim() == 2 and vis_bits.is_contiguous(), "vis_bits.dtype == torch.int64 and vis_bits.dim() == 2 and vis_bits.is_contiguous()")
    _require(xyz.dtype == torch.float64 and xyz.dim() == 2 and xyz.shape[1] == 3 and xyz.is_contiguous(), "xyz.dtype == torch.float64 and xyz.dim() == 2 and xyz.shape[1] == 3 and xyz.is_contiguous(")
    _require(obj_offsets.dtype == torch.int32 and obj_vertices.dtype == torch.int32, "obj_offsets.dtype == torch.int32 and obj_vertices.dtype == torch.int32")
    F, n_words = vis_bits.shape
    O = obj_offsets.numel() - 1
    dev = vis_bits.device
    lo = torch.empty((O, F, 3), dtype=torch.float64, device=dev)
    hi = torch.empty((O, F, 3), dtype=torch.float64, device=dev)
    count = torch.empty((O, F), dtype=torch.int32, device=dev)
    _lib.check(lib.mspa_object_extents(_ptr(vis_bits), F, n_words, _ptr(xyz), xyz.shape[0], _ptr(obj_offsets),
                                       _ptr(obj_vertices), obj_vertices.numel(), O, _ptr(lo), _ptr(hi), _ptr(count),
                                       _stream_ptr()))
    return lo, hi, count


def track_pair_distances(world: torch.Tensor, points: Sequence[int], visible_frames: Sequence[np.ndarray]):
    """Enqueue K5c: for each selected point, the distances between its world positions in every two of its visible
    frames (i < j, row-major).  Returns a list of float64 NumPy arrays, one per point (n(n-1)/2 entries)."""
    _require_gpu()
    lib = _lib.load()
    _require(world.dtype == torch.float64 and world.dim() == 3 and world.is_contiguous(), "world.dtype == torch.float64 and world.dim() == 3 and world.is_contiguous()")
    T, P, _ = world.shape
    S = len(points)
    if S == 0:
        return []
    lens = np.array([len(v) for v in visible_frames], dtype=np.int64)
    f_off = np.zeros(S + 1, dtype=np.int64)
    f_off[1:] = np.cumsum(lens)
    o_off = np.zeros(S + 1, dtype=np.int64)
    o_off[1:] = np.cumsum(lens * (lens - 1) // 2)
    dev = world.device
    frames = np.concatenate([np.asarray(v, dtype=np.int32) for v in visible_frames]) if f_off[-1] else np.zeros(0, np.int32)
    if frames.size and (frames.min() < 0 or frames.max() >= T):
        raise ValueError("frame index outside the track")
    pts = np.asarray(points, dtype=np.int32)
    if pts.min() < 0 or pts.max() >= P:
        raise ValueError("point index outside the track")
    out = torch.empty((int(o_off[-1]),), dtype=torch.float64, device=dev)
    if o_off[-1]:
        # named tensors: a temporary would be freed (and its block reused by the next upload) before the launch
        pts_t = torch.from_numpy(pts).to(dev)
        f_off_t = torch.from_numpy(f_off.astype(np.int32)).to(dev)
        frames_t = torch.from_numpy(frames).to(dev)
        o_off_t = torch.from_numpy(o_off).to(dev)
        _lib.check(lib.mspa_track_pair_distances(_ptr(world), T, P, _ptr(pts_t), _ptr(f_off_t), _ptr(frames_t), S,
                                                 int(lens.max()), _ptr(o_off_t), _ptr(out), _stream_ptr()))
    host = out.cpu().numpy()
    return [host[o_off[s]:o_off[s + 1]] for s in range(S)]
