"""The sharded form of the multiple-choice correspondence builder (heads.visual_correspondence_dot_dataset with a communicator)
against the one-process form, on two gloo ranks with a synthetic numeric backend: same records, same marks, same generator
state at the end -- including a row whose random distractor lands on the correct pixel (the ranks agree on the first such row
and rewind together; VC_D:366)."""
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multi-spatialmllm_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from mspa import heads, shard  # noqa: E402

H, W = 480, 640
N_ROWS = 60


class Backend:
    """Deterministic numbers in place of K2 / K6: sizes and projections are functions of (scene, images, position)."""

    def __init__(self, clash_row=None, clash_pixel=None, rows=None):
        self.calls = []
        self.clash_row, self.clash_pixel, self.rows = clash_row, clash_pixel, rows

    def image_hw(self, scene_id):
        return (H, W)

    def common_counts(self, scene_id, pairs):
        self.calls.append(("counts", scene_id))
        if scene_id == "scene0003_00":
            return None                                        # a scene the visibility file does not know
        return [(hash_int(scene_id, a, b) % 7) * 50 for a, b in pairs]      # some rows have no common point

    def project(self, scene_id, jobs):
        self.calls.append(("project", scene_id))
        out = []
        for a, b, pos in jobs:
            h = hash_int(scene_id, a, b, pos)
            uv2 = (float(h % (W - 20)), float((h // 1000) % (H - 20)))
            if self.clash_row is not None:
                r = self.rows[self.clash_row]
                if r["scene_id"] == scene_id and {a, b} == {r["image_id1"], r["image_id2"]}:
                    uv2 = (float(self.clash_pixel[0]), float(self.clash_pixel[1]))
            out.append((pos * 3 + 1, (float((h // 7) % W), float((h // 11) % H)), uv2, True, True))
        return out


def hash_int(*parts):
    import hashlib
    return int.from_bytes(hashlib.sha256("|".join(map(str, parts)).encode()).digest()[:6], "little")


def _rows():
    rng = random.Random(3)
    rows = []
    for k in range(N_ROWS):
        s = rng.randrange(5)
        a, b = rng.sample(range(0, 400, 5), 2)
        rows.append({"scene_id": f"scene{s:04d}_00", "image_id1": f"{a:05d}", "image_id2": f"{b:05d}", "overlap": 20.0})
    return rows


def _single(clash):
    rows = _rows()
    marks = []
    rng = random.Random(9)
    be = Backend(*(clash or (None, None)), rows=rows)
    out = heads.visual_correspondence_dot_dataset(rows, be, rng=rng, on_mark=lambda *a: marks.append(a[0]))
    return out, marks, rng.getstate()


def _find_clash():
    """A row with a record and its first distractor pixel, from an unrigged run: make that row's correct pixel equal it."""
    out, _, _ = _single(None)
    live = [k for k, r in enumerate(out) if r is not None]
    k = live[len(live) // 2]
    return k, tuple(out[k]["p2_list"][1])


def _rank_main(rank, world, port, q, clash):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MSPA_DIST_BACKEND="gloo")
    ctx = shard.context_from_env(torch.device("cpu"))
    rows = _rows()
    marks = []
    rng = random.Random(9)
    be = Backend(*(clash or (None, None)), rows=rows)
    out = heads.visual_correspondence_dot_dataset(rows, be, rng=rng, on_mark=lambda *a: marks.append(a[0]), ctx=ctx)
    q.put((rank, [None if r is None else json.loads(r) for r in out], marks, rng.getstate(), sorted({s for _, s in be.calls})))
    ctx.barrier()
    ctx.close()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_dot_dataset_two_ranks_equals_one_process_with_and_without_a_rewind():
    import torch.multiprocessing as mp
    clash_free = None
    clash = _find_clash()
    for case in (clash_free, clash):
        want, want_marks, want_state = _single(case)
        assert sum(r is not None for r in want) >= 20
        if case is not None:                                       # the rigged row really went through the rejection path
            base, _, _ = _single(None)
            assert want[case[0]] is not None and want != base
        mpc = mp.get_context("spawn")
        q = mpc.Queue()
        port = _free_port()
        procs = [mpc.Process(target=_rank_main, args=(r, 2, port, q, case)) for r in range(2)]
        for p in procs:
            p.start()
        res = {r: (out, marks, state, scenes) for r, out, marks, state, scenes in (q.get(timeout=120) for _ in procs)}
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res[0][0] == json.loads(json.dumps(want))           # rank 0: every record, in row order
        assert all(r is None for r in res[1][0])
        assert sorted(res[0][1] + res[1][1]) == sorted(want_marks) and res[0][1] and res[1][1]      # every image drawn once, by its owner
        assert res[0][2] == want_state == res[1][2]                # the generator ends where one process leaves it, on every rank
        assert not (set(res[0][3]) & set(res[1][3])) and len(set(res[0][3]) | set(res[1][3])) == 5   # scenes are dealt, not shared


# ---- the chained build of the depth-comparison engines (draws depend on the numerics) ------------------------------------
SCENES = [f"scene{k:04d}_00" for k in range(11)]
TIED = {"scene0001_00", "scene0002_00", "scene0006_00", "scene0010_00"}       # scenes where a pair is skipped: more draws than guessed


def _fake_engine():
    for name in [m for m in sys.modules if m == "spatial_engine" or m.startswith("spatial_engine.")]:
        if not (getattr(sys.modules[name], "__file__", None) or "").startswith(PKG):     # the reference's package, imported by
            del sys.modules[name]                                                        # an earlier oracle-vs-reference test
    if sys.path[0] != PKG:
        sys.path.insert(0, PKG)
    from spatial_engine.depth_perception._coor_base import DepthCoorEngineBase

    class Info:
        def prefetched_scenes(self, ids, num_workers=8, device="cpu"):
            return iter([("resident", s) for s in ids])

        def get_sorted_keys(self):
            return list(SCENES)

    class Engine(DepthCoorEngineBase):
        CHAINED = True
        task_name = "fake_comparison"

        def __init__(self, warning_file):
            self.scene_info, self.warning_file = Info(), warning_file
            self.all_max_samples, self.max_n_points_per_image = 25, 1

        def _scene_records_on(self, scene, scene_id, _draws, dry_run=False):
            n = 3 + int(scene_id[5:9]) % 4
            draws = [random.random() for _ in range(n)]
            if dry_run:
                return []
            assert scene is None or scene == ("resident", scene_id)
            if scene_id in TIED:                                  # the evaluated scene turns out to need more of the stream
                self._warn(f"Warning: a pair of {scene_id} was skipped.\n")
                draws += [random.random(), random.random()]
            return [{"id": f"{scene_id}_{k}", "v": v, "conversations": [{"from": "human", "value": f"q{k}"}]} for k, v in enumerate(draws)]

        def generate_qa_training_single_scene(self, scene_id):
            return self._scene_records_on(None, scene_id, None)
    return Engine


def _chained_rank(rank, world, port, q, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MSPA_DIST_BACKEND="gloo")
    ctx = shard.context_from_env(torch.device("cpu"))
    random.seed(21)
    eng = _fake_engine()(os.path.join(out_dir, f"warn_w{world}.txt"))
    eng.generate_qa_training_data(os.path.join(out_dir, f"w{world}"))
    eng.all_max_samples = 7
    eng.generate_qa_eval_data(os.path.join(out_dir, f"w{world}_val"))      # a second call in a row: the generator is in step
    q.put((rank, random.getstate()))
    ctx.barrier()
    ctx.close()


def test_chained_depth_comparison_build_equals_one_process(tmp_path):
    import torch.multiprocessing as mp
    out_dir = str(tmp_path)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    random.seed(21)
    eng = _fake_engine()(os.path.join(out_dir, "warn_w1.txt"))
    eng.generate_qa_training_data(os.path.join(out_dir, "w1"))
    eng.all_max_samples = 7
    eng.generate_qa_eval_data(os.path.join(out_dir, "w1_val"))
    want_state = random.getstate()
    for world in (2, 3):
        mpc = mp.get_context("spawn")
        q = mpc.Queue()
        port = _free_port()
        procs = [mpc.Process(target=_chained_rank, args=(r, world, port, q, out_dir)) for r in range(world)]
        for p in procs:
            p.start()
        states = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(s == want_state for _, s in states)
        for sub in ("", "_val"):
            a = open(os.path.join(out_dir, f"w1{sub}", "fake_comparison.jsonl"), "rb").read()
            b = open(os.path.join(out_dir, f"w{world}{sub}", "fake_comparison.jsonl"), "rb").read()
            assert a == b and len(a) > 100
        assert open(os.path.join(out_dir, "warn_w1.txt")).read() == open(os.path.join(out_dir, f"warn_w{world}.txt")).read() != ""
