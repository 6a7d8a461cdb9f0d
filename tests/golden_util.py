"""Load tests/golden/*.npz (reference outputs frozen by oracle/gen_golden.py)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class GoldenScene:
    def __init__(self, name):
        self.g = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        g = self.g
        self.K, self.A, self.points = g["K"], g["A"], g["points"]
        self.image_ids = [str(s) for s in g["image_ids"]]
        self.E = {i: g["E"][n] for n, i in enumerate(self.image_ids)}
        self.depth = {i: g["depth"][n] for n, i in enumerate(self.image_ids)}
        self.color = {i: g["color"][n] for n, i in enumerate(self.image_ids)} if "color" in g.files else {}
        self.color_hw = tuple(int(v) for v in g["color_hw"])
        self.depth_hw = tuple(int(v) for v in g["depth_hw"])
        self.valid_image_ids = [str(s) for s in g["valid_image_ids"]] if "valid_image_ids" in g.files else \
            list(self.image_ids)

    def __getitem__(self, k):
        return self.g[k]

    def json(self, k):
        return json.loads(str(self.g[k]))


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.int64)


def same_f64(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def close_f64(a, b, rtol=1e-9, scale=None):
    """|a-b| <= rtol * max(|b|, scale): the 1e-5-relative bar of BASELINE.json with 4 digits to spare."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    m = ~np.isnan(b)
    ref = np.abs(b[m])
    if scale is not None:
        ref = np.maximum(ref, scale)
    with np.errstate(invalid="ignore"):
        ok = (np.abs(a[m] - b[m]) <= rtol * ref) | (a[m] == b[m])
    return bool(np.all(ok))
