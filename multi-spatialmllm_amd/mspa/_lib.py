"""ctypes binding of libmspa.so.  Fails loudly: no library, no engine."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int32, c_int64, c_uint32, c_void_p, POINTER

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MSPA_LIB", os.path.join(_PKG_ROOT, "libmspa.so"))

MSPA_OK, MSPA_EINVAL, MSPA_EHIP, MSPA_EUNSUPPORTED = 0, -1, -2, -3
MAT_KINV, MAT_E, MAT_A, MAT_EINV_ALIGNED, MAT_K, MAT_UNPROJ, MAT_REPROJ, MAT_BOUNDS, FRAME_MATS = 0, 1, 2, 3, 4, 5, 6, 7, 8
GUARD_C = 256.0
CAM_EINV, CAM_K, CAM_BOUNDS, CAM_MATS = 0, 1, 2, 3
PAIR_FAST = 1
PAIR_STREAM = 2
PAIR_WORD_STRIPES = 0x200
CORR_TILE_W, CORR_TILE_H, CORR_TILE_CAP = 64, 48, 64 * 48
KERNEL_NONE, KERNEL_PAIR_EXACT, KERNEL_PAIR_FAST, KERNEL_PAIR_FAST_LINEAR, KERNEL_PAIR_FAST_TIGHT, KERNEL_PAIR_FAST_SCALED, KERNEL_PAIR_FAST_RECT = range(7)


class MspaError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmspa error {code}: {message}")
        self.code = code


_lib = None

# name -> (restype, argtypes); mirrors include/mspa.h one to one
_SIGNATURES = {
    "mspa_version": (c_int, []),
    "mspa_last_error_string": (c_char_p, []),
    "mspa_frame_bounds_host": (c_int, [c_void_p, c_int32]),
    "mspa_camera_bounds_host": (c_int, [c_void_p, c_int32]),
    "mspa_device_info": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int64), POINTER(c_int),
                                 c_char_p, c_int]),
    "mspa_pair_reproject": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64,
                                    c_int32, c_int32, c_int32, c_int32,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "mspa_pair_reproject_last_kernel": (c_int, []),
    "mspa_corr_tiles": (c_int64, [c_int32, c_int32]),
    "mspa_pair_correspondences_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32, c_int32, c_int32, c_uint32]),
    "mspa_pair_correspondences": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_uint32, c_void_p]),
    "mspa_compact_correspondences": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "mspa_vertex_visibility": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p,
                                       c_int32, c_int32, c_int32, c_int32,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mspa_vertex_visibility_ex": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p, c_int32, c_void_p,
                                          c_int32, c_int32, c_int32, c_int32, c_double,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mspa_check_visibility_ex": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_double,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "mspa_project_samples_ex": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32,
                                        c_int32, c_int32, c_int32, c_double, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "mspa_pair_overlap": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "mspa_overlap_workspace_bytes": (c_int64, [c_int32, c_int32, c_int64]),
    "mspa_scene_overlap": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mspa_overlap_matrix": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "mspa_bits_popcount": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mspa_bits_expand": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "mspa_bits_transpose": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "mspa_format_int_lists_host": (c_int64, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "mspa_format_token_lists_host": (c_int64, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_int64,
                                               c_void_p]),
    "mspa_format_int_keys_host": (c_int64, [c_char_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "mspa_format_list_costs_device": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "mspa_format_lists_device": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_void_p, c_void_p]),
    "mspa_gather_blocks_host": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int32]),
    "mspa_inflate_blocks_host": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32]),
    "mspa_read_depth_png_host": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "mspa_inflate_zlib_fast_host": (c_int, [c_void_p, c_int64, c_void_p, c_int64]),
    "mspa_png_header_host": (c_int, [c_char_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                     POINTER(c_int32)]),
    "mspa_stream_create_reserving": (c_int, [c_int32, POINTER(c_void_p)]),
    "mspa_stream_destroy": (c_int, [c_void_p]),
    "mspa_inflate_blocks_device": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                           c_void_p, c_void_p]),
    "mspa_png_unfilter_device": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "mspa_png_pack_idat_host": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                        POINTER(c_int64), c_int32]),
    "mspa_check_visibility": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "mspa_pair_pose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p,
                               c_void_p]),
    "mspa_extract_yaw_pitch": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "mspa_track_to_world": (c_int, [c_void_p, c_void_p, c_int32, c_int32, POINTER(c_double), c_int32, c_int32,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "mspa_select_common_point": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "mspa_project_samples": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32,
                                     c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "mspa_object_extents": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "mspa_track_pair_distances": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                          c_void_p, c_void_p, c_void_p]),
    "mspa_track_rigidity_loss": (c_int, [c_void_p, c_int32, c_int32, c_double, c_void_p, c_void_p]),
    "mspa_track_displacement": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                        c_double, c_double, c_void_p, c_void_p, c_void_p]),
}


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"libmspa.so not found at {LIB_PATH}. Build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the geometry kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError here = header and library out of sync
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES.keys())


def check(rc: int):
    if rc != MSPA_OK:
        msg = load().mspa_last_error_string()
        raise MspaError(rc, msg.decode() if msg else "unknown")


def version() -> int:
    return load().mspa_version()


def device_info(device: int = 0) -> dict:
    n_cu, wave, clock = c_int(0), c_int(0), c_int(0)
    hbm = c_int64(0)
    name = ctypes.create_string_buffer(64)
    check(load().mspa_device_info(device, ctypes.byref(n_cu), ctypes.byref(wave), ctypes.byref(hbm),
                                  ctypes.byref(clock), name, 64))
    return {"name": name.value.decode(), "n_cu": n_cu.value, "wave_size": wave.value,
            "hbm_bytes": hbm.value, "clock_khz": clock.value}


_OWN_STREAMS: dict = {}
_OWN_STREAMS_LOCK = __import__("threading").Lock()


def own_stream(role: str, device):
    """A HIP stream of this process's OWN for a long-lived role (a decode slot, the sweeps' copy stream, an encoder thread's side
    stream), created once per (role, device) through mspa_stream_create_reserving(0) and kept: ``torch.cuda.ExternalStream``.

    ``torch.cuda.Stream()`` is not that: it hands out the next of a pool of 32 streams per device, round robin.  A process that
    sweeps more than once asks for 1 + 8 streams per pass (copy stream, encoder threads) on top of its 8 decode slots -- from the
    third pass on an encoder thread's "own" stream IS a decode slot's stream, its downloads queue behind a 40 ms inflate and the
    slot's next scene behind them (the fourth pass of every multi-pass measurement was the slow one: profiles/r06_sweep_timeline.md)."""
    import torch
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (role, index)
    with _OWN_STREAMS_LOCK:
        st = _OWN_STREAMS.get(key)
        if st is None:
            ptr = ctypes.c_void_p(0)
            with torch.cuda.device(index):
                check(load().mspa_stream_create_reserving(0, ctypes.byref(ptr)))
            st = _OWN_STREAMS[key] = torch.cuda.ExternalStream(ptr.value, device=torch.device("cuda", index))
        return st
