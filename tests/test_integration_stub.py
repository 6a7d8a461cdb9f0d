"""The ctypes stub printed in INTEGRATION.md section 1 is executed as written (only the library path is substituted) and
must give the oracle's visibility masks -- the document stays honest."""
import os
import re

import numpy as np
import pytest

from mspa import synth
from oracle import np_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_integration_md_stub_runs_and_matches_oracle():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", doc, re.S).group(1)
    lib = os.path.join(ROOT, "multi-spatialmllm_amd", "libmspa.so")
    code = code.replace('ctypes.CDLL("libmspa.so")', f"ctypes.CDLL({lib!r})")
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    sc = synth.make_scene(77, n_points=5000, n_frames=5, color_hw=(96, 128), depth_hw=(96, 128), invalid_pose_frac=0,
                          with_color=False)
    ids = sc.valid_image_ids
    Ea = [sc.A @ sc.E[i] for i in ids]
    mask = ns["check_point_visibility_batch"](sc.points[:, :3], sc.K, Ea, [sc.depth[i] for i in ids], sc.color_hw)
    want = np.stack([O.vertex_visibility(sc.points[:, :3], sc.K, e, sc.depth[i], sc.color_hw)[0] for e, i in zip(Ea, ids)])
    assert mask.dtype == bool and np.array_equal(mask, want) and mask.sum() > 100
