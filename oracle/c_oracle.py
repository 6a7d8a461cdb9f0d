"""TEST INFRASTRUCTURE -- ctypes face of oracle/c_oracle.c (see that file's header)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmspa_c_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.mspa_c_pair_overlap.restype = ctypes.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _m(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(16))


def project_points(points_xyz, K, E_c2w):
    """inv() on the host exactly like IH:57, the rest in c_oracle.c."""
    pts = np.ascontiguousarray(points_xyz, dtype=np.float64)
    n, stride = pts.shape
    uv = np.empty((n, 2))
    d = np.empty(n)
    einv, k = _m(np.linalg.inv(E_c2w)), _m(K)
    lib().mspa_c_project_points(_p(pts), ctypes.c_int64(n), ctypes.c_int64(stride), _p(einv), _p(k), _p(uv), _p(d))
    return uv, d


def vertex_visibility(points_xyz, K, E_aligned, depth_image, image_hw):
    pts = np.ascontiguousarray(points_xyz, dtype=np.float64)
    n, stride = pts.shape
    dimg = np.ascontiguousarray(depth_image, dtype=np.uint16)
    mask = np.empty(n, dtype=np.uint8)
    uv = np.empty((n, 2))
    d = np.empty(n)
    einv, k = _m(np.linalg.inv(E_aligned)), _m(K)
    lib().mspa_c_vertex_visibility(_p(pts), ctypes.c_int64(n), ctypes.c_int64(stride), _p(einv), _p(k), _p(dimg),
                                   dimg.shape[0], dimg.shape[1], int(image_hw[0]), int(image_hw[1]),
                                   _p(mask), _p(uv), _p(d))
    return mask.astype(bool), uv, d


def frame_pair(depth1, depth2, K, E1, E2, A, image_hw):
    d1 = np.ascontiguousarray(depth1, dtype=np.uint16)
    d2 = np.ascontiguousarray(depth2, dtype=np.uint16)
    H, W = int(image_hw[0]), int(image_hw[1])
    P = H * W
    out = {"valid": np.empty(P, np.uint8), "xyz": np.empty((P, 3)), "uv2": np.empty((P, 2)),
           "depth2": np.empty(P), "xiyi": np.empty((P, 2), np.int64), "vis": np.empty(P, np.uint8)}
    counts = np.zeros(2, np.int64)
    mats = [_m(np.linalg.inv(K)), _m(E1), _m(A), _m(np.linalg.inv(A @ E2)), _m(K)]
    lib().mspa_c_frame_pair(_p(d1), _p(d2), d1.shape[0], d1.shape[1], H, W, *[_p(m) for m in mats],
                            _p(out["valid"]), _p(out["xyz"]), _p(out["uv2"]), _p(out["depth2"]),
                            _p(out["xiyi"]), _p(out["vis"]), _p(counts))
    out["valid"] = out["valid"].astype(bool)
    out["vis"] = out["vis"].astype(bool)
    out["xi"], out["yi"] = out["xiyi"][:, 0].copy(), out["xiyi"][:, 1].copy()
    out["n_valid"], out["n_vis"] = int(counts[0]), int(counts[1])
    return out


def pair_overlap(m1, m2):
    a = np.ascontiguousarray(m1, dtype=np.uint8)
    b = np.ascontiguousarray(m2, dtype=np.uint8)
    inter, uni = ctypes.c_int64(0), ctypes.c_int64(0)
    v = lib().mspa_c_pair_overlap(_p(a), _p(b), ctypes.c_int64(a.size), ctypes.byref(inter), ctypes.byref(uni))
    return v, inter.value, uni.value


def matmul4(a, b):
    out = np.empty(16)
    lib().mspa_c_matmul4(_p(_m(a)), _p(_m(b)), _p(out))
    return out.reshape(4, 4)
