"""mspa/tape.py on the CPU: the numeric-tape codec, the recorder / player pair over a stand-in engine, and the routing of
``engine`` through a proxy.  (The heads on real tapes: tests/test_gpu_heads.py::test_pipeline_end_to_end_and_sharding_invariance.)"""
import types

import numpy as np
import pytest
import torch

from mspa import tape


def test_codec_round_trip_and_row_padding():
    val = (torch.arange(6, dtype=torch.int32).reshape(2, 3), None,
           {"world": torch.full((2, 2), 0.1, dtype=torch.float64), "ok": torch.tensor([1, 0], dtype=torch.uint8)},
           [np.array([1.5, 2.5]), np.zeros(0)], torch.zeros((0, 3), dtype=torch.float64))
    rec = tape.Recorder(types.SimpleNamespace())
    rec.note(val)
    rec.note(torch.tensor([7], dtype=torch.int64))
    rows = rec.rows()
    assert rows.shape[1] == tape.WIDTH and rows.dtype == np.float64
    player = tape.Player(types.SimpleNamespace(), rows, "cpu")
    got = player.next()
    assert torch.equal(got[0], val[0]) and got[0].dtype == torch.int32 and got[1] is None
    assert torch.equal(got[2]["world"], val[2]["world"]) and got[2]["ok"].dtype == torch.uint8
    assert isinstance(got[3], list) and np.array_equal(got[3][0], val[3][0]) and got[3][1].shape == (0,)
    assert tuple(got[4].shape) == (0, 3)
    assert int(player.next()[0]) == 7
    with pytest.raises(ValueError):
        rec.note(torch.tensor([2 ** 60], dtype=torch.int64))               # not exactly representable on a float64 tape


def test_recorder_and_player_stand_in_for_the_engine_module():
    calls = []

    def pair_pose(a, b):
        calls.append("pair_pose")
        return torch.tensor([[a + b, 1.0 / 3.0]], dtype=torch.float64)

    real = types.SimpleNamespace(pair_pose=pair_pose, all_pairs=lambda n: "not taped")
    rec = tape.Recorder(real)
    r1, r2 = rec.pair_pose(1.0, 2.0), rec.pair_pose(5.0, 0.5)
    assert rec.all_pairs(3) == "not taped" and calls == ["pair_pose", "pair_pose"]
    player = tape.Player(real, rec.rows(), "cpu")
    assert torch.equal(player.pair_pose(None, None), r1) and torch.equal(player.pair_pose("ignored", 0), r2)
    assert player.all_pairs(3) == "not taped" and calls == ["pair_pose", "pair_pose"]      # nothing was launched again
    with pytest.raises(RuntimeError, match="exhausted"):
        player.pair_pose(0, 0)


def test_engine_as_routes_and_restores():
    import mspa
    from mspa import engine, scene
    proxy = object()
    with tape.engine_as(proxy):
        from mspa import engine as inside
        assert inside is proxy and scene.engine is proxy
    from mspa import engine as after
    assert after is engine and scene.engine is engine and mspa.engine is engine
    with pytest.raises(ZeroDivisionError):
        with tape.engine_as(proxy):
            1 / 0
    assert scene.engine is engine
