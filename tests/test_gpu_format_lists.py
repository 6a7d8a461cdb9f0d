"""K10 (csrc/format_lists.hip): the JSON text of a CSR table's lists written on the device, against json.dumps and libmspa's host
formatters (make_visibility_info.py:38-73 stores every list of the visibility index as its JSON text) -- bit for bit, incl. empty
lists, one-item lists, negative integers, 10-digit values, and a whole scene's index through ``visindex.from_bits(text=True)``."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def _lists_to_csr(lists):
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    flat = np.array([v for x in lists for v in x], dtype=np.int32)
    return off, flat


def _texts(text, offsets):
    t, o = text.cpu().numpy().tobytes(), offsets.cpu().numpy()
    return [t[o[k]:o[k + 1]].decode() for k in range(len(o) - 1)]


def test_integer_lists_equal_json_dumps():
    import torch
    from mspa import engine
    rng = np.random.default_rng(5)
    lists = [[], [0], [7, 8], [], [], [2147483647, -2147483648, -1, 10, 99, 100, 999999999, 1000000000], []]
    lists += [rng.integers(0, 200000, rng.integers(0, 40)).tolist() for _ in range(300)]
    lists += [list(range(5000))]
    off, flat = _lists_to_csr(lists)
    text, toff = engine.format_lists_device(torch.from_numpy(off).cuda(), torch.from_numpy(flat).cuda())
    assert _texts(text, toff) == [json.dumps(x) for x in lists]
    # nothing but empty lists, and no list at all
    text, toff = engine.format_lists_device(torch.zeros(4, dtype=torch.int64).cuda(), torch.zeros(0, dtype=torch.int32).cuda())
    assert _texts(text, toff) == ["[]", "[]", "[]"]
    text, toff = engine.format_lists_device(torch.zeros(1, dtype=torch.int64).cuda(), torch.zeros(0, dtype=torch.int32).cuda())
    assert _texts(text, toff) == [] and toff.tolist() == [0]


def test_token_lists_equal_json_dumps_and_bad_ids_are_refused():
    import torch
    from mspa import engine
    rng = np.random.default_rng(6)
    ids = [f"{5 * k:05d}" for k in range(320)] + ["a", "frame \"x\""]
    tokens = [json.dumps(i).encode() for i in ids]
    lists = [[], [0], [321, 320, 0], []] + [sorted(rng.choice(len(ids), rng.integers(0, 50), replace=False).tolist()) for _ in range(500)]
    off, flat = _lists_to_csr(lists)
    text, toff = engine.format_lists_device(torch.from_numpy(off).cuda(), torch.from_numpy(flat).cuda(), tokens)
    assert _texts(text, toff) == [json.dumps([ids[v] for v in x]) for x in lists]
    with pytest.raises(ValueError):
        engine.format_lists_device(torch.from_numpy(off).cuda(), torch.from_numpy(np.where(flat == 5, len(ids), flat).astype(np.int32)).cuda(), tokens)


def test_a_scenes_index_with_device_text_equals_the_host_formatters():
    import torch
    from mspa import engine, synth, visindex
    from mspa.scene import SceneOnDevice
    sc = synth.make_scene(777, n_points=20000, n_frames=9, color_hw=(96, 128), depth_hw=(96, 128), invalid_pose_frac=0.0, with_color=False)
    first = sc.valid_image_ids[0]
    sc.depth[first] = np.zeros_like(sc.depth[first])                          # an image that sees nothing: an empty list
    scene = SceneOnDevice(sc.K, sc.A, sc.E, sc.depth, sc.color_hw, sc.points, torch.device("cuda", 0))
    bits = scene._visibility()["bits"]
    n = int(scene.xyz.shape[0])
    on_host = visindex.from_bits(bits, scene.ids, n)
    on_dev = visindex.from_bits(bits, scene.ids, n, text=True, indices=False)
    assert on_dev.i2p_text is not None and on_dev.i2p_indices is None and on_dev.empty_images() == on_host.empty_images() == [first]
    a, b = on_host.to_arrow("scene0777_00"), on_dev.to_arrow("scene0777_00")
    assert a.equals(b) and a.num_rows == len(scene.ids) + n
    want = on_host.to_dict()
    vals = dict(zip(b.column("key").to_pylist(), b.column("values").to_pylist()))
    assert vals[f"scene0777_00:image_to_points:{scene.ids[1]}"] == json.dumps(want["image_to_points"][scene.ids[1]])
    assert vals["scene0777_00:point_to_images:17"] == json.dumps(want["point_to_images"][17])
    # ids that are not in sorted() order: the text comes from the host formatters (MVI:117 sorts per vertex)
    shuffled = visindex.from_bits(bits, list(scene.ids)[::-1], n, text=True)
    assert shuffled.i2p_text is None and shuffled.i2p_indices is not None
