"""Task heads: QA records in the reference's InternVL chat schema, numerics from the HIP kernels.

Each head is split in two:
  * a **numeric stage** that runs on the GPU for a whole batch (K4 pair pose, K6a/K6b correspondence
    selection + projection, K5 track geometry) and returns plain arrays;
  * a **record stage**, pure Python, that draws from ``random`` in exactly the order the reference's
    scripts do and fills the templates (schemas, ids, integer formatting).
The record stage never computes geometry; handing it the reference's template tables and seeds
reproduces the reference's JSONL byte for byte (tests/test_heads_vs_reference.py checks that with
numerics supplied by the oracle, tests/test_gpu_heads.py checks the numeric stages on the MI355X).

Reference scripts mirrored (paths under spatial_engine/): camera_movement/
camera_movement_engine_train_val.py (CME), visual_correspondence/
visual_correspondence_qa_engine_coor_2_coor.py (VC_C), depth_perception/
depth_estimation_coor_engine.py (DE_C), object_movement/single_object_movement_engine_coord.py (OM_C).
"""
from __future__ import annotations

import json
import os
import random as _random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import templates as T
from .hostinfo import quietly


# --------------------------------------------------------------------------------------------
# shared helpers
# --------------------------------------------------------------------------------------------
def to_eval_sample(train_sample: dict) -> dict:
    """Train record -> eval record: the conversation collapses into ``text`` (CME:247-269)."""
    conversation = train_sample.pop("conversations")
    train_sample["text"] = conversation[0]["value"]
    return train_sample


class JsonLine(bytes):
    """One record already serialised by the rank that built it (``json.dumps(record).encode()``): what a sharded dataset
    builder hands to ``write_jsonl`` on rank 0 -- the text crossed the fabric once and is not parsed again."""
    __slots__ = ()


def write_jsonl(path: str, records: Iterable):
    with open(path, "wb") as f:
        for r in records:
            f.write((r if isinstance(r, JsonLine) else json.dumps(r).encode()) + b"\n")


def sample_indices(n: int, k: int, rng=_random) -> List[int]:
    """Positions ``random.sample(seq_of_len_n, k)`` would pick: the draw depends only on n and k, so
    sampling ``range(n)`` consumes the generator identically and yields the positions themselves."""
    return rng.sample(range(n), k)


# --------------------------------------------------------------------------------------------
# camera movement (CME:153-245)
# --------------------------------------------------------------------------------------------
def camera_movement_answer_values(d, yaw_angle: float, pitch_angle: float) -> dict:
    """The 15 answer fields from the camera-1-frame displacement ``d`` and the (already swapped and
    wrapped) angles: sign words and truncated integers exactly as CME:209-225."""
    d = np.asarray(d, dtype=np.float64)
    return {
        "x_movement": "right" if d[0] > 0 else "left",
        "y_movement": "down" if d[1] > 0 else "up",
        "z_movement": "forward" if d[2] > 0 else "backward",
        "yaw_movement": "left" if yaw_angle > 0 else "right",
        "pitch_movement": "up" if pitch_angle > 0 else "down",
        "x_distance": int(abs(d[0]) * 1000),
        "y_distance": int(abs(d[1]) * 1000),
        "z_distance": int(abs(d[2]) * 1000),
        "yaw_angle": int(abs(yaw_angle)),
        "pitch_angle": int(abs(pitch_angle)),
        "x_value": int(d[0] * 1000),
        "y_value": int(d[1] * 1000),
        "z_value": int(d[2] * 1000),
        "total_distance": int(np.linalg.norm(d) * 1000),
        "displacement_vector": d.tolist(),
    }


def camera_movement_draw(row: dict, question_type: str, templates: T.TemplateSet = T.CAMERA_MOVEMENT, rng=_random):
    """The four draws of one camera-movement record, in the reference's order (CME:163, 196, 203-204): swap coin, task
    description, question, answer template -- as (swap, ti, qi, ai).  They depend on nothing the kernels compute, so a
    sharded builder lets every rank run this (microseconds per row) for ALL rows and format only its own."""
    swap = rng.random() < 0.5                                                                  # CME:163-166
    ti = rng.choice(range(len(templates.task_description)))
    if float(row["overlap"]) < 0.1:
        raise NotImplementedError("overlap < 0.1 is not supported yet.")                       # CME:199-201
    qi = rng.choice(range(len(templates.questions[question_type])))
    ai = rng.choice(range(len(templates.answers[question_type])))
    return swap, ti, qi, ai


def camera_movement_record(row: dict, idx: int, question_type: str, rel_t_12, rel_t_21, image_hw: Tuple[int, int],
                           templates: T.TemplateSet = T.CAMERA_MOVEMENT, rng=_random, draw=None) -> dict:
    """One record of the camera-movement head.  ``row`` has scene_id, image_id1, image_id2, overlap, yaw,
    pitch, distance (a row of the pair table); ``rel_t_12`` / ``rel_t_21`` are the translation columns of
    inv(E1)@E2 and inv(E2)@E1 (K4 computes both directions).  ``draw``: the record's draws when they were made ahead
    (``camera_movement_draw``); otherwise they are made here, interleaved with the checks exactly as upstream."""
    scene_id, image1, image2 = row["scene_id"], row["image_id1"], row["image_id2"]
    overlap, yaw_angle, pitch_angle = float(row["overlap"]), float(row["yaw"]), float(row["pitch"])
    d = np.asarray(rel_t_12, dtype=np.float64)
    if (rng.random() < 0.5) if draw is None else draw[0]:    # CME:163-166
        yaw_angle, pitch_angle = -yaw_angle, -pitch_angle
        image1, image2 = image2, image1
        d = np.asarray(rel_t_21, dtype=np.float64)
    if abs(yaw_angle) > 180:                                 # CME:168-172
        yaw_angle = yaw_angle - 360 if yaw_angle > 0 else yaw_angle + 360
    distance = np.linalg.norm(d)
    assert abs(distance - row["distance"]) < 0.1, \
        f"distance is not close to the distance from df for {scene_id} {image1} {image2}."     # CME:193
    task_description = rng.choice(templates.task_description) if draw is None else templates.task_description[draw[1]]
    if overlap < 0.1:
        raise NotImplementedError("overlap < 0.1 is not supported yet.")                       # CME:199-201
    question = rng.choice(templates.questions[question_type]) if draw is None else templates.questions[question_type][draw[2]]
    answer_template = rng.choice(templates.answers[question_type]) if draw is None else templates.answers[question_type][draw[3]]
    answer_values = camera_movement_answer_values(d, yaw_angle, pitch_angle)
    H, W = image_hw
    return {
        "id": idx,
        "image": [f"{scene_id}/{image1}.jpg", f"{scene_id}/{image2}.jpg"],
        "conversations": [{"from": "human", "value": f"{task_description}\n{question}"},
                          {"from": "gpt", "value": answer_template.format(**answer_values)}],
        "height_list": [H] * 2,
        "width_list": [W] * 2,
        "answer_values": answer_values,
        "question_type": question_type,
        "gt_value": answer_values[question_type],
    }


def camera_movement_numeric(scene, rows: Sequence[dict]):
    """GPU stage: K4 for every row in both directions -> ([n,3] rel_t_12, [n,3] rel_t_21)."""
    import torch
    from . import engine
    F = len(scene.ids)
    idx = np.array([[scene.index[r["image_id1"]], scene.index[r["image_id2"]]] for r in rows], dtype=np.int32)
    both = torch.from_numpy(np.concatenate([idx, idx[:, ::-1]], axis=0).copy()).to(scene.device)
    out = engine.pair_pose(*scene.pose_tables(), both).cpu().numpy()
    n = len(rows)
    return out[:n, 3:6], out[n:, 3:6]


def camera_movement_records(scene, rows: Sequence[dict], question_type: str, image_hw, start_idx: int = 0,
                            templates: T.TemplateSet = T.CAMERA_MOVEMENT, rng=_random) -> List[dict]:
    t12, t21 = camera_movement_numeric(scene, rows)
    return [camera_movement_record(r, start_idx + k, question_type, t12[k], t21[k], image_hw, templates, rng)
            for k, r in enumerate(rows)]


@quietly
def camera_movement_dataset(rows: Sequence, frame_pose, image_hw_of, question_type: str,
                            templates: T.TemplateSet = T.CAMERA_MOVEMENT, rng=_random, device="cuda", ctx=None,
                            transform=None) -> List:
    """The record loop of CME.build_train_dataset (CME:295-299) for rows that may span many scenes: the relative
    poses of the rows come from one K4 launch over a table of the distinct frames, then the records are filled in
    row order -- the numerics never influence a draw, so the ``random`` stream is the per-row loop's.

    ``frame_pose(scene_id, image_id)`` -> axis-aligned camera-to-world 4x4; ``image_hw_of(scene_id, image_id)`` -> (H, W).

    With a communicator (``ctx``: one process per GPU) the TEXT is built in parallel: every rank makes the draws of all rows
    (``camera_movement_draw``: the generator ends where a single process leaves it, on every rank), formats a contiguous
    slice of the rows -- its own K4 launch, its own image-size lookups, ``transform`` (e.g. the eval form) applied -- and
    the finished JSON lines go to rank 0 through ONE ``shard.gather_bytes``.  Rank 0 gets the records as ``JsonLine``s in
    row order, the other ranks an empty list.  Without one the records are returned as dicts, as before.
    """
    import torch
    from . import engine
    n = len(rows)
    if n == 0:
        return []
    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    draws = None
    if ctx is not None:                                        # all rows, every rank: ~4 generator calls per row
        draws = [camera_movement_draw(r, question_type, templates, rng) for r in rows]
    lo, hi = (0, n) if ctx is None else _partition(n, world, rank)
    failure: Optional[BaseException] = None
    mine: List[dict] = []
    try:
        mine = _camera_movement_slice(rows, lo, hi, frame_pose, image_hw_of, question_type, templates, rng, device, draws)
        if transform is not None:
            mine = [transform(rec) for rec in mine]
    except Exception as e:                                     # the distance assert, a missing image, a backend error on THIS
        if ctx is None:                                        # rank's slice: the other ranks must not be left in the gather
            raise
        failure = e
    if ctx is None:
        return mine
    from . import shard
    shard.raise_together(ctx, failure, "camera_movement_dataset")
    parts = shard.gather_bytes("".join(json.dumps(rec) + "\n" for rec in mine).encode(), ctx, dst=0)
    if rank != 0:
        return []
    return [JsonLine(line) for p in parts for line in bytes(p).split(b"\n")[:-1]]


def _camera_movement_slice(rows, lo, hi, frame_pose, image_hw_of, question_type, templates, rng, device, draws) -> List[dict]:
    """Records of rows [lo, hi): one K4 launch over a table of the slice's distinct frames, then the per-row formatting."""
    import torch
    from . import engine
    table: Dict[Tuple[str, str], int] = {}
    poses = []
    idx = np.empty((hi - lo, 2), dtype=np.int32)
    for k in range(lo, hi):
        r = rows[k]
        for c, key in enumerate(((r["scene_id"], r["image_id1"]), (r["scene_id"], r["image_id2"]))):
            if key not in table:
                E = np.asarray(frame_pose(*key), dtype=np.float64)
                assert not np.isnan(E).any(), f"E is nan for {key[0]} {key[1]}"        # CME:160-161
                table[key] = len(poses)
                poses.append(E)
            idx[k - lo, c] = table[key]
    if hi <= lo:
        return []
    E_all = np.stack(poses)
    E_t = torch.from_numpy(E_all.reshape(-1, 16)).to(device)
    Einv_t = torch.from_numpy(np.linalg.inv(E_all).reshape(-1, 16)).to(device)       # same LAPACK call as per frame
    zeros = torch.zeros(len(poses), dtype=torch.float64, device=device)
    both = torch.from_numpy(np.concatenate([idx, idx[:, ::-1]], axis=0).copy()).to(device)
    out = engine.pair_pose(E_t, Einv_t, zeros, zeros, both).cpu().numpy()
    m = hi - lo
    return [camera_movement_record(rows[k], k, question_type, out[k - lo, 3:6], out[m + k - lo, 3:6],
                                   image_hw_of(rows[k]["scene_id"], rows[k]["image_id1"]), templates, rng,
                                   draw=None if draws is None else draws[k]) for k in range(lo, hi)]


def _partition(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# --------------------------------------------------------------------------------------------
# visual correspondence, coordinate -> coordinate (VC_C:264-394)
# --------------------------------------------------------------------------------------------
def normalised(uv_row, image_hw) -> Tuple[int, int]:
    """round(u / W * 1000), round(v / H * 1000) with Python's banker's rounding (VC_C:341-344)."""
    H, W = image_hw
    return round((uv_row[0] / W) * 1000), round((uv_row[1] / H) * 1000)


def visual_correspondence_draws(rows: Sequence[dict], n_common: Sequence[int], templates: T.TemplateSet,
                                rng=_random, max_points_per_pair: int = 1, invisible: Sequence = None):
    """Every random decision of VC_C.build_training_sample for a batch, in the reference's order, given only
    the number of common visible vertices per pair: swap coin, position(s) inside the sorted intersection,
    template picks.  Returns a list of dicts (None for pairs without common vertices, VC_C:304-309)."""
    draws = []
    for r_idx, (row, n) in enumerate(zip(rows, n_common)):
        hidden = (invisible[r_idx] if invisible is not None else None) or ()   # slots whose vertex failed the re-check:
        swap = rng.random() < 0.5                                        # VC_C:280       upstream draws no template for them
        if n == 0:
            draws.append(None)
            continue
        if n >= max_points_per_pair:                                     # VC_C:312-315
            pos = sample_indices(int(n), max_points_per_pair, rng)
        else:
            pos = [rng.choice(range(int(n))) for _ in range(max_points_per_pair)]
        picks = [None if s in hidden else (rng.choice(range(len(templates.task_description))),
                                           rng.choice(range(len(templates.questions["default"]))),
                                           rng.choice(range(len(templates.answers["default"]))))
                 for s in range(len(pos))]
        draws.append({"swap": swap, "positions": pos, "picks": picks})
    return draws


def visual_correspondence_record(row: dict, idx: int, draw: dict, uv1: np.ndarray, uv2: np.ndarray, image_hw,
                                 templates: T.TemplateSet = T.VISUAL_CORRESPONDENCE) -> dict:
    """Record stage.  uv1/uv2: [k,2] projections of the selected vertices into the (post-swap) first and
    second image."""
    scene_id, image1, image2 = row["scene_id"], row["image_id1"], row["image_id2"]
    if draw["swap"]:
        image1, image2 = image2, image1
    H, W = image_hw
    conversation, p1_list, p2_list = [], [], []
    for k, pick in enumerate(draw["picks"]):
        if pick is None:                                                   # vertex not visible after all (VC_C:327-338)
            continue
        ti, qi, ai = pick
        x1, y1 = normalised(uv1[k], image_hw)
        x2, y2 = normalised(uv2[k], image_hw)
        question = templates.questions["default"][qi].format(x1=x1, y1=y1, x2=x2, y2=y2)
        answer = templates.answers["default"][ai].format(x1=x1, y1=y1, x2=x2, y2=y2)
        if not conversation:
            conversation = [{"from": "human", "value": f"{templates.task_description[ti]}\n{question}"},
                            {"from": "gpt", "value": answer}]
        else:
            conversation += [{"from": "human", "value": question}, {"from": "gpt", "value": answer}]
        p1_list.append((x1, y1))
        p2_list.append((x2, y2))
    return {
        "id": f"{scene_id}_{image1}_{image2}_{idx}",
        "image": [f"{scene_id}/{image1}.jpg", f"{scene_id}/{image2}.jpg"],
        "conversations": conversation,
        "height_list": [H, H],
        "width_list": [W, W],
        "question_type": "visual_correspondence_coor_2_coor",
        "p1_list": p1_list,
        "p2_list": p2_list,
        "gt_value": list(p2_list[0]),
    }


def visual_correspondence_records(scene, rows: Sequence[dict], image_hw, start_idx: int = 0,
                                  templates: T.TemplateSet = T.VISUAL_CORRESPONDENCE, rng=_random) -> List[dict]:
    """Whole head for one scene: K2 intersection sizes -> host draws -> K6a selection -> K6b projection."""
    import torch
    from . import engine
    bits = scene._visibility()["bits"]
    dev = scene.device
    idx = np.array([[scene.index[r["image_id1"]], scene.index[r["image_id2"]]] for r in rows], dtype=np.int32)
    _, inter, _ = engine.pair_overlap(bits, torch.from_numpy(idx).to(dev), want_counts=True)
    n_common = inter.cpu().numpy()
    draws = visual_correspondence_draws(rows, n_common, templates, rng)
    sel, owner = [], []
    for k, dr in enumerate(draws):
        if dr is None:
            continue
        for j in dr["positions"]:
            sel.append([idx[k, 0], idx[k, 1], j])
            owner.append(k)
    records = []
    if sel:
        sel_t = torch.tensor(sel, dtype=torch.int32, device=dev)
        vert = engine.select_common_point(bits, sel_t)
        first = torch.tensor([idx[k, 1] if draws[k]["swap"] else idx[k, 0] for k in owner], dtype=torch.int32, device=dev)
        second = torch.tensor([idx[k, 0] if draws[k]["swap"] else idx[k, 1] for k in owner], dtype=torch.int32, device=dev)
        samples = torch.cat([torch.stack([vert, first], 1), torch.stack([vert, second], 1)], 0).contiguous()
        uv, _, vis = engine.project_samples(scene.xyz, scene.cam_mats, scene.depth, scene.image_hw, samples, scene.depth_scale)
        uv, vis = uv.cpu().numpy(), vis.cpu().numpy().astype(bool)
        m = len(owner)
        if not vis.all():
            raise RuntimeError("a vertex taken from both visibility lists failed the visibility re-check (IH:297-300)")
        by_row: Dict[int, List[int]] = {}
        for s, k in enumerate(owner):
            by_row.setdefault(k, []).append(s)
        for k, ss in by_row.items():
            records.append(visual_correspondence_record(rows[k], start_idx + k, draws[k], uv[ss], uv[[s + m for s in ss]],
                                                        image_hw, templates))
    return records


@quietly
def visual_correspondence_dataset(rows: Sequence, get_scene, get_bits=None, templates: T.TemplateSet = T.VISUAL_CORRESPONDENCE,
                                  rng=_random, max_points_per_pair: int = 1, on_warn=None, ctx=None,
                                  transform=None) -> List[Optional[dict]]:
    """The record loop of VC_C.build_train_dataset (VC_C:424-429) for rows that may span many scenes.

    Pass 1 (per scene, GPU): size of the common visible set of every row (K2 on the scene's bitsets).
    Pass 2 (host, global row order): every draw of VC_C.build_training_sample -- they depend on those sizes only.
    Pass 3 (per scene, GPU): the drawn positions -> vertices (K6a) -> both projections + visibility re-check (K6b).
    Returns one entry per row in row order, None where upstream returns None (no common vertex / unknown scene).
    Upstream draws no template for a vertex that fails the re-check (VC_C:327-338; only possible when the visibility index
    is stale): such a row is found after pass 3, the generator is rewound to it (checkpoints every 1024 rows) and the
    passes resume with that slot marked -- as in ``visual_correspondence_dot_dataset``.

    ``get_scene(scene_id)`` -> resident ``SceneOnDevice`` (or None if the scene is unknown) and ``get_bits(scene_id, scene)``
    -> [F, n_words] bitsets in ``scene.ids`` order (default: K1 on the resident scene); or pass a ready backend
    (``GpuCorrespondenceBackend``-like object) as ``get_scene``.

    With a communicator (``ctx``: one process per GPU) the SCENES are dealt over the ranks (longest-first by their number of
    rows): a rank reads, uploads and runs passes 1 and 3 only for its own scenes; the sizes of pass 1 are summed over the ranks
    (one all_reduce of an int64 [rows, 2] table: every row has exactly one owner), pass 2 -- all draws, microseconds per row --
    runs identically on every rank (so the generator ends where a single process leaves it), each rank builds the records of its
    scenes' rows (``transform`` applied) and ONE ``shard.gather_bytes`` brings the JSON lines to rank 0, which returns them in
    row order as ``JsonLine``s (None where upstream returns None); the other ranks return a list of Nones.  A stale visibility
    index (a drawn vertex fails the re-check: the rewind below) is a single-process affair and raises here.
    """
    backend = get_scene if hasattr(get_scene, "project") else GpuCorrespondenceBackend(get_scene, get_bits)
    warn = on_warn or (lambda message: None)
    by_scene: Dict[str, List[int]] = {}
    for k, r in enumerate(rows):
        by_scene.setdefault(r["scene_id"], []).append(k)
    n = len(rows)
    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    if ctx is not None:
        from . import shard
        names = list(by_scene)
        bins = shard.lpt_assign([float(len(by_scene[s])) for s in names], world)
        mine = {names[i] for i in bins[rank]}
        if rank != 0:
            warn = lambda message: None                                    # the warning file is rank 0's
    else:
        mine = set(by_scene)
    n_common = [0] * n
    known = [False] * n
    hw: Dict[str, Tuple[int, int]] = {}
    failure: Optional[BaseException] = None
    try:
        for scene_id, ks in by_scene.items():                              # pass 1
            if scene_id not in mine:
                continue
            counts = backend.common_counts(scene_id, [(rows[k]["image_id1"], rows[k]["image_id2"]) for k in ks])
            if counts is None:
                continue
            hw[scene_id] = backend.image_hw(scene_id)
            for k, c in zip(ks, counts):
                known[k], n_common[k] = True, c
    except Exception as e:                                                 # a rank-local failure (a missing frame, a backend
        if ctx is None:                                                    # error) must reach EVERY rank before the collective
            raise
        failure = e
    if ctx is not None:
        from . import shard
        shard.raise_together(ctx, failure, "visual_correspondence_dataset (pass 1)")
    if ctx is not None and n:
        import torch
        import torch.distributed as dist
        table = torch.tensor([[int(kn), int(c)] for kn, c in zip(known, n_common)], dtype=torch.int64, device=ctx.collective_device)
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=ctx.group)
        table = table.cpu().numpy()
        known, n_common = [bool(v) for v in table[:, 0]], [int(v) for v in table[:, 1]]

    hidden: Dict[int, set] = {}                                            # row -> slots whose vertex failed the re-check
    out: List[Optional[dict]] = [None] * n
    STRIDE = 1024

    def draw(k):
        if not known[k]:
            rng.random()                                                   # the swap coin comes before the scene check (VC_C:280)
            return None
        return visual_correspondence_draws([rows[k]], [n_common[k]], templates, rng, max_points_per_pair, [hidden.get(k)])[0]

    try:
        start = 0
        while start < n:
            draws: Dict[int, Optional[dict]] = {}
            checkpoints: Dict[int, tuple] = {}
            for k in range(start, n):                                          # pass 2
                if (k - start) % STRIDE == 0:
                    checkpoints[k] = rng.getstate()
                draws[k] = draw(k)
            proj: Dict[int, list] = {}
            for scene_id, ks in by_scene.items():                              # pass 3
                if scene_id not in mine:
                    continue
                live = [k for k in ks if k >= start and draws[k] is not None]
                if not live:
                    continue
                jobs, owner = [], []
                for k in live:
                    r, d = rows[k], draws[k]
                    i1, i2 = (r["image_id2"], r["image_id1"]) if d["swap"] else (r["image_id1"], r["image_id2"])
                    for j in d["positions"]:
                        jobs.append((i1, i2, j))
                        owner.append(k)
                for k, res in zip(owner, backend.project(scene_id, jobs)):
                    proj.setdefault(k, []).append(res)
            redo = None
            for k in range(start, n):                                          # records, until a row needs its draws corrected
                r, d = rows[k], draws[k]
                if d is None:
                    if not known[k]:
                        warn(f"[build_training_sample] Warning: Visibility info not found for scene {r['scene_id']}\n")
                    else:
                        warn(f"[build_training_sample] Warning: No common visible points for scene {r['scene_id']} "
                             f"{r['image_id1']}, {r['image_id2']}\n")
                    continue
                image1, image2 = (r["image_id2"], r["image_id1"]) if d["swap"] else (r["image_id1"], r["image_id2"])
                if r["scene_id"] not in mine:                                  # another rank's row: its record arrives as bytes;
                    if all(p is None for p in d["picks"]):                     # its warning line is rank 0's to write (the picks
                        warn(f"[build_training_sample] Warning: No conversation for scene {r['scene_id']} {image1}, {image2}\n")
                    continue                                                   # come from the replicated draws: known here)
                bad = {s for s, (_, _, _, ok1, ok2) in enumerate(proj[k]) if not (ok1 and ok2)}
                if bad - hidden.get(k, set()):
                    if ctx is not None:
                        raise RuntimeError(f"visual_correspondence_dataset: vertex {proj[k][min(bad)][0]} of scene {r['scene_id']} failed the "
                                           "visibility re-check (a visibility index that does not belong to these frames); the rewind that "
                                           "reproduces upstream's draws for such rows runs in a single process only")
                    for s in sorted(bad):
                        vertex, _, _, ok1, ok2 = proj[k][s]
                        if not ok1:
                            warn(f"Warning: Point {vertex} is not visible in image {image1} in scene {r['scene_id']}.\n")
                        if not ok2:
                            warn(f"Warning: Point {vertex} is not visible in image {image2} in scene {r['scene_id']}.\n")
                    redo = (k, bad)
                    break
                if all(p is None for p in d["picks"]):                         # VC_C:373-378
                    warn(f"[build_training_sample] Warning: No conversation for scene {r['scene_id']} {image1}, {image2}\n")
                    continue
                uv1 = np.stack([res[1] for res in proj[k]])
                uv2 = np.stack([res[2] for res in proj[k]])
                out[k] = visual_correspondence_record(r, k, d, uv1, uv2, hw[r["scene_id"]], templates)
            if redo is None:
                break
            k, bad = redo                                                      # take the generator back to the start of row k
            base = max(c for c in checkpoints if c <= k)
            rng.setstate(checkpoints[base])
            for j in range(base, k):
                draw(j)
            hidden[k] = set(bad)
            start = k
        if transform is not None:
            out = [None if rec is None else transform(rec) for rec in out]
    except Exception as e:                                                 # e.g. the stale-index error of ONE rank's rows
        if ctx is None:
            raise
        failure = e
    if ctx is None:
        return out
    from . import shard
    shard.raise_together(ctx, failure, "visual_correspondence_dataset (passes 2-3)")
    lines = "".join(f"{k}\t{json.dumps(rec)}\n" for k, rec in enumerate(out) if rec is not None).encode()
    parts = shard.gather_bytes(lines, ctx, dst=0)
    merged: List[Optional[dict]] = [None] * n
    if rank == 0:
        for p in parts:
            for line in bytes(p).split(b"\n")[:-1]:
                k, _, body = line.partition(b"\t")
                merged[int(k)] = JsonLine(body)
    return merged


def _vc_dot_row_draws(n_common: int, known: bool, image_hw, templates: T.TemplateSet, rng, correct_point=None):
    """All draws of VC_D.build_training_sample for one row (VC_D:290-386), in order.  ``correct_point`` None = assume that
    no random distractor lands exactly on the correct pixel (checked afterwards); a pixel = apply upstream's rejection;
    the string "invisible" = the drawn vertex failed the visibility re-check, upstream returns right after the pick."""
    from .annotate import generate_distinct_colors
    swap = rng.random() < 0.5                                              # VC_D:290
    if not known or n_common == 0:
        return {"swap": swap, "dead": True}
    pos = sample_indices(int(n_common), 1, rng)[0]                          # VC_D:326-327 (max_points_per_pair == 1)
    if correct_point == "invisible":                                       # VC_D:339-351: no further draw for this row
        return {"swap": swap, "dead": True, "pos": pos, "invisible": True}
    color1 = (rng.randint(0, 255), rng.randint(0, 255), rng.randint(0, 255))   # VC_D:356
    H, W = image_hw
    wrong = []
    while len(wrong) < 3:                                                  # VC_D:362-367
        p = (rng.randint(0, W - 10), rng.randint(0, H - 10))
        if correct_point is None or p != correct_point:
            wrong.append(p)
    order = [0, 1, 2, 3]                                                   # 0 = the correct point
    rng.shuffle(order)                                                     # VC_D:371
    labels = ["A", "B", "C", "D"]
    rng.shuffle(labels)                                                    # VC_D:374
    colors = generate_distinct_colors(4, rng)                              # VC_D:379
    ti = rng.choice(range(len(templates.task_description)))
    qi = rng.choice(range(len(templates.questions["default"])))
    ai = rng.choice(range(len(templates.answers["default"])))
    return {"swap": swap, "dead": False, "pos": pos, "color1": color1, "wrong": wrong, "order": order, "labels": labels,
            "colors": colors, "picks": (ti, qi, ai)}


class GpuCorrespondenceBackend:
    """Numerics of the correspondence heads for rows that span scenes: sizes of the common visible sets (K2), the
    drawn positions -> vertices (K6a) and their projections into both images (K6b), one batch per scene."""

    def __init__(self, get_scene, get_bits=None):
        self.get_scene = get_scene
        self.bits_of = get_bits or (lambda scene_id, scene: scene._visibility()["bits"])

    def image_hw(self, scene_id):
        scene = self.get_scene(scene_id)
        return None if scene is None else scene.image_hw

    def common_counts(self, scene_id, pairs: Sequence[Tuple[str, str]]) -> Optional[List[int]]:
        """len(intersect1d(points(image1), points(image2))) per pair; None if the scene is unknown."""
        import torch
        from . import engine
        scene = self.get_scene(scene_id)
        if scene is None:
            return None
        out = [0] * len(pairs)
        usable = [n for n, (a, b) in enumerate(pairs) if a in scene.index and b in scene.index]
        if usable:
            idx = np.array([[scene.index[pairs[n][0]], scene.index[pairs[n][1]]] for n in usable], dtype=np.int32)
            _, inter, _ = engine.pair_overlap(self.bits_of(scene_id, scene), torch.from_numpy(idx).to(scene.device), want_counts=True)
            for n, c in zip(usable, inter.cpu().numpy()):
                out[n] = int(c)
        return out

    def project(self, scene_id, jobs: Sequence[Tuple[str, str, int]]):
        """jobs: (first image, second image, position in their sorted common set) -> [(vertex, uv1, uv2, ok1, ok2)]."""
        import torch
        from . import engine
        scene = self.get_scene(scene_id)
        dev = scene.device
        a = [scene.index[j[0]] for j in jobs]
        b = [scene.index[j[1]] for j in jobs]
        sel = torch.tensor([[x, y, j[2]] for x, y, j in zip(a, b, jobs)], dtype=torch.int32, device=dev)
        vert = engine.select_common_point(self.bits_of(scene_id, scene), sel)
        first = torch.tensor(a, dtype=torch.int32, device=dev)
        second = torch.tensor(b, dtype=torch.int32, device=dev)
        samples = torch.cat([torch.stack([vert, first], 1), torch.stack([vert, second], 1)], 0).contiguous()
        uv, _, ok = engine.project_samples(scene.xyz, scene.cam_mats, scene.depth, scene.image_hw, samples, scene.depth_scale)
        uv, ok, vert_h = uv.cpu().numpy(), ok.cpu().numpy().astype(bool), vert.cpu().numpy()
        m = len(jobs)
        return [(int(vert_h[s]), uv[s], uv[s + m], bool(ok[s]), bool(ok[s + m])) for s in range(m)]


@quietly
def visual_correspondence_dot_dataset(rows: Sequence, backend, templates: T.TemplateSet = None, rng=_random, on_warn=None,
                                      on_mark=None, ctx=None, transform=None) -> List[Optional[dict]]:
    """Record loop of the multiple-choice correspondence head (visual_correspondence_qa_engine_dot_2_multichoice.py
    :279-433, VC_D) for rows that may span scenes.  ``backend``: ``GpuCorrespondenceBackend`` (or anything with its
    three methods).

    Pass 1 (per scene): sizes of the common visible sets.  Pass 2 (host, global row order): every draw.  Pass 3 (per
    scene): drawn positions -> vertices -> projections.  Upstream rejects a random distractor that coincides with the
    correct pixel (VC_D:366) -- a draw that depends on the projection.  The draws are made assuming no such coincidence
    and checked afterwards; at the first row where one did occur the generator is taken back to that row (replayed from
    a checkpoint kept every 1024 rows), the row is redrawn with the rejection applied, and the passes resume behind it.
    ``on_mark(row_index, scene_id, first_image, second_image, vertex, p1_pixel, colour1, labelled_points, colours)``.

    With a communicator (``ctx``: one process per GPU) the SCENES are dealt over the ranks (longest-first by rows), as in
    ``visual_correspondence_dataset``: passes 1 and 3, the records and -- the expensive part of this head -- the two annotated
    JPEGs per record (``on_mark``) only for a rank's own scenes; the sizes of pass 1 summed over the ranks; all draws on every
    rank.  A coincidence is expected a handful of times in the 500 K-row train set, so it is handled, not refused: the ranks
    agree on the FIRST row where one occurred (one all_reduce(MIN) of row << 32 | x << 16 | y), every rank rewinds to it with the
    owner's pixel, and records and images are only produced for rows in front of it -- the files of a sharded run are those of
    one process.  The finished JSON lines (``transform`` applied) travel to rank 0 in one ``shard.gather_bytes``; the other
    ranks get Nones.  Only a stale visibility index (a drawn vertex that fails the depth test) stays a one-process affair."""
    templates = templates or T.VISUAL_CORRESPONDENCE_DOT
    warn = on_warn or (lambda message: None)
    by_scene: Dict[str, List[int]] = {}
    for k, r in enumerate(rows):
        by_scene.setdefault(r["scene_id"], []).append(k)
    n = len(rows)
    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    if ctx is not None:
        from . import shard
        names = list(by_scene)
        bins = shard.lpt_assign([float(len(by_scene[s_])) for s_ in names], world)
        mine = {names[i] for i in bins[rank]}
        if rank != 0:
            warn = lambda message: None                                    # noqa: E731 -- the warning file is rank 0's
    else:
        mine = set(by_scene)
    n_common = [0] * n
    known = [False] * n
    hw: Dict[str, Tuple[int, int]] = {}
    failure: Optional[BaseException] = None
    try:
        for scene_id, ks in by_scene.items():                              # pass 1
            if scene_id not in mine:
                continue
            counts = backend.common_counts(scene_id, [(rows[k]["image_id1"], rows[k]["image_id2"]) for k in ks])
            if counts is None:
                continue
            hw[scene_id] = backend.image_hw(scene_id)
            for k, c in zip(ks, counts):
                known[k], n_common[k] = True, c
    except Exception as e:
        if ctx is None:
            raise
        failure = e
    if ctx is not None:
        import torch
        import torch.distributed as dist
        shard.raise_together(ctx, failure, "visual_correspondence_dot_dataset (pass 1)")
        if n:
            # known, size, and the image size the distractors are drawn inside (H, W): every row has exactly one owner
            table = torch.tensor([[int(kn), int(c), hw.get(rows[k]["scene_id"], (0, 0))[0] if kn else 0,
                                   hw.get(rows[k]["scene_id"], (0, 0))[1] if kn else 0]
                                  for k, (kn, c) in enumerate(zip(known, n_common))], dtype=torch.int64, device=ctx.collective_device)
            dist.all_reduce(table, op=dist.ReduceOp.SUM, group=ctx.group)
            table = table.cpu().numpy()
            known, n_common = [bool(v) for v in table[:, 0]], [int(v) for v in table[:, 1]]
            for k in range(n):
                if known[k]:
                    hw.setdefault(rows[k]["scene_id"], (int(table[k, 2]), int(table[k, 3])))

    out: List[Optional[dict]] = [None] * n
    start = 0
    forced: Dict[int, Tuple[int, int]] = {}                                # row -> correct pixel where the rejection applies
    STRIDE = 1024
    NONE = (1 << 62)

    def draw(k):
        return _vc_dot_row_draws(n_common[k], known[k], hw.get(rows[k]["scene_id"], (0, 0)), templates, rng, forced.get(k))

    try:
        while start < n:
            draws: Dict[int, dict] = {}
            checkpoints: Dict[int, tuple] = {}
            for k in range(start, n):                                      # pass 2
                if (k - start) % STRIDE == 0:
                    checkpoints[k] = rng.getstate()
                draws[k] = draw(k)
            proj: Dict[int, tuple] = {}
            for scene_id, ks in by_scene.items():                          # pass 3
                if scene_id not in mine:
                    continue
                live = [k for k in ks if k >= start and not draws[k]["dead"]]
                if not live:
                    continue
                jobs = []
                for k in live:
                    r, d = rows[k], draws[k]
                    i1, i2 = (r["image_id2"], r["image_id1"]) if d["swap"] else (r["image_id1"], r["image_id2"])
                    jobs.append((i1, i2, d["pos"]))
                for k, res in zip(live, backend.project(scene_id, jobs)):
                    proj[k] = res
            # the first row (of mine) where the assumption behind the draws fails; nothing is written before the ranks agree
            local = NONE
            for k in range(start, n):
                if k not in proj:
                    continue
                vertex, uv1, uv2, ok1, ok2 = proj[k]
                if not (ok1 and ok2):
                    local = (k << 32) | (0xFFFF << 16) | 0xFFFF
                    break
                correct = (int(uv2[0]), int(uv2[1]))
                if k not in forced and correct in draws[k]["wrong"]:
                    local = (k << 32) | ((correct[0] & 0xFFFF) << 16) | (correct[1] & 0xFFFF)
                    break
            first = local
            if ctx is not None:
                import torch
                import torch.distributed as dist
                t = torch.tensor([local], dtype=torch.int64, device=ctx.collective_device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=ctx.group)
                first = int(t.item())
            stop = n if first == NONE else first >> 32
            clash = None
            if first != NONE:
                cx, cy = (first >> 16) & 0xFFFF, first & 0xFFFF
                clash = (stop, "invisible" if (cx, cy) == (0xFFFF, 0xFFFF) else (cx, cy))
            for k in range(start, min(n, stop + (1 if clash and clash[1] == "invisible" else 0))):     # records in front of it
                r, d = rows[k], draws[k]
                image1, image2 = (r["image_id2"], r["image_id1"]) if d["swap"] else (r["image_id1"], r["image_id2"])
                if d["dead"]:
                    if d.get("invisible"):
                        pass                                               # warned when the failure was found
                    elif not known[k]:
                        warn(f"[build_training_sample] Warning: Visibility info not found for scene {r['scene_id']}\n")
                    else:
                        warn(f"[build_training_sample] Warning: No common visible points for scene {r['scene_id']} {image1}, {image2}\n")
                    continue
                if k not in proj:                                          # another rank's row: its record arrives as bytes
                    continue
                vertex, uv1, uv2, ok1, ok2 = proj[k]
                if not (ok1 and ok2):
                    # the index and the depth test disagree (a stale visibility file): upstream warns and returns before any
                    # further draw (VC_D:339-351) -- rewind to this row and redraw it that way
                    if ctx is not None:
                        raise RuntimeError(f"visual_correspondence_dot_dataset: vertex {vertex} of scene {r['scene_id']} failed the "
                                           "visibility re-check (a visibility index that does not belong to these frames); the rewind "
                                           "that reproduces upstream's draws for such rows runs in a single process only")
                    if not ok1:
                        warn(f"Warning: Point {vertex} is not visible in image {image1} in scene {r['scene_id']}.\n")
                    if not ok2:
                        warn(f"Warning: Point {vertex} is not visible in image {image2} in scene {r['scene_id']}.\n")
                    break
                correct = (int(uv2[0]), int(uv2[1]))
                H, W = hw[r["scene_id"]]
                points = [correct] + d["wrong"]
                labeled = dict(zip(d["labels"], [points[j] for j in d["order"]]))
                correct_label = [lab for lab, p in labeled.items() if p == correct][0]
                ti, qi, ai = d["picks"]
                p1_pixel = (int(uv1[0]), int(uv1[1]))
                if on_mark is not None:
                    on_mark(k, r["scene_id"], image1, image2, vertex, p1_pixel, d["color1"], labeled,
                            {lab: d["colors"][j] for j, lab in enumerate(d["labels"])})
                out[k] = {
                    "id": f"{k}_p{vertex}",
                    "image": [os.path.join(r["scene_id"], f"{k}_point{vertex}_{image1}_{image2}_img1.jpg"),
                              os.path.join(r["scene_id"], f"{k}_point{vertex}_{image1}_{image2}_img2.jpg")],
                    "conversations": [{"from": "human", "value": f"{templates.task_description[ti]}\n{templates.questions['default'][qi]}"},
                                      {"from": "gpt", "value": templates.answers["default"][ai].format(correct_label=correct_label)}],
                    "height_list": [H] * 2,
                    "width_list": [W] * 2,
                    "question_type": "visual_correspondence_multiple_choice",
                    "gt_value": correct_label,
                    "p1_list": [p1_pixel[0], p1_pixel[1]],
                    "p2_list": [correct] + d["wrong"],
                }
            if clash is None:
                break
            k, correct = clash                                             # take the generator back to the start of row k
            base = max(c for c in checkpoints if c <= k)
            rng.setstate(checkpoints[base])
            for j in range(base, k):
                draw(j)
            forced[k] = correct
            start = k
        if transform is not None:
            out = [None if rec is None else transform(rec) for rec in out]
    except Exception as e:
        if ctx is None:
            raise
        failure = e
    if ctx is None:
        return out
    shard.raise_together(ctx, failure, "visual_correspondence_dot_dataset (passes 2-3)")
    lines = "".join(f"{k}\t{json.dumps(rec)}\n" for k, rec in enumerate(out) if rec is not None).encode()
    parts = shard.gather_bytes(lines, ctx, dst=0)
    merged: List[Optional[dict]] = [None] * n
    if rank == 0:
        for p in parts:
            for line in bytes(p).split(b"\n")[:-1]:
                k, _, body = line.partition(b"\t")
                merged[int(k)] = JsonLine(body)
    return merged


# --------------------------------------------------------------------------------------------
# depth estimation by coordinate (DE_C:175-254)
# --------------------------------------------------------------------------------------------
def depth_estimation_draws(image_ids: Sequence[str], n_visible: Dict[str, int], max_samples: int,
                           templates: T.TemplateSet, rng=_random, max_n_points_per_image: int = 1, dot: bool = False):
    """Random decisions of DE_C.generate_qa_training_single_scene: which images, which visible-vertex
    positions, which templates (question, answer, task description -- in that order, DE_C:224-230)."""
    n_images = min(max_samples, len(image_ids)) if max_samples > 0 else len(image_ids)
    sampled = rng.sample(list(image_ids), n_images)                       # DE_C:187
    draws = []
    for image_id in sampled:
        n = int(n_visible[image_id])
        if n < max_n_points_per_image:                                    # DE_C:195-198
            pos = rng.choices(range(n), k=max_n_points_per_image)
        else:
            pos = sample_indices(n, max_n_points_per_image, rng)
        picks, colors = [], []
        for _ in pos:
            if dot:                                                           # DE_D:219-222: the disc colour comes first
                from .annotate import generate_distinct_colors
                colors.append(generate_distinct_colors(1, rng)[0])
            picks.append((rng.choice(range(len(templates.questions["default"]))),
                          rng.choice(range(len(templates.answers["default"]))),
                          rng.choice(range(len(templates.task_description)))))
        draws.append({"image_id": image_id, "positions": pos, "picks": picks, "colors": colors})
    return draws


def depth_estimation_record(scene_id: str, image_id: str, vertex: int, uv_row, depth_m: float, pick, image_hw,
                            templates: T.TemplateSet = T.DEPTH_ESTIMATION, dot: bool = False) -> dict:
    H, W = image_hw
    x, y = normalised(uv_row, image_hw)
    depth = round(depth_m * 1000)                                         # DE_C:218
    qi, ai, ti = pick
    question = templates.questions["default"][qi]
    if not dot:                                                           # the dot question names no coordinates (DE_D:232)
        question = question.format(x1=x, y1=y)
    answer = templates.answers["default"][ai].format(x1=x, y1=y, depth=depth)
    return {
        "id": f"{scene_id}_{image_id}_point{vertex}",
        "image": [f"{scene_id}/{image_id}_p{vertex}_annotated.jpg" if dot else f"{scene_id}/{image_id}.jpg"],
        "conversations": [{"from": "human", "value": f"{templates.task_description[ti]}\n{question}"},
                          {"from": "gpt", "value": answer}],
        "height_list": [H],
        "width_list": [W],
        "question_type": "depth_estimation_dot" if dot else "depth_estimation_coor",
        "gt_value": depth,
        "ori_coordinates": [int(uv_row[0]), int(uv_row[1])],
    }


def depth_estimation_records_fn(scene_id: str, image_ids: Sequence[str], n_visible, numeric_fn, image_hw,
                                max_samples: int = -1, templates: T.TemplateSet = T.DEPTH_ESTIMATION, rng=_random,
                                max_n_points_per_image: int = 1, on_skip=None, dot: bool = False,
                                on_mark=None, draws=None) -> List[dict]:
    """DE_C / DE_D.generate_qa_training_single_scene with the numerics behind ``numeric_fn`` (see
    ``depth_comparison_records``); ``n_visible`` is only indexed for the images that get sampled.  In ``dot`` mode
    ``on_mark(scene_id, image_id, vertex, (px, py), colour)`` receives the disc to draw for every record.
    ``draws``: the scene's random decisions when they were made ahead (``depth_estimation_draws`` -- they depend on nothing the
    kernels compute, so a sharded builder makes them for ALL scenes on every rank and builds records only for its own)."""
    if draws is None:
        draws = depth_estimation_draws(image_ids, n_visible, max_samples, templates, rng, max_n_points_per_image, dot)
    numerics = numeric_fn([(d["image_id"], j) for d in draws for j in d["positions"]])
    records, s = [], 0
    for dr in draws:
        for n, pick in enumerate(dr["picks"]):
            vertex, uv_row, depth_m = numerics[s]
            s += 1
            if uv_row is None:
                if on_skip is not None:
                    on_skip(scene_id, dr["image_id"], [int(vertex)])
                continue
            if dot and on_mark is not None:
                on_mark(scene_id, dr["image_id"], int(vertex), (int(uv_row[0]), int(uv_row[1])), dr["colors"][n])
            records.append(depth_estimation_record(scene_id, dr["image_id"], int(vertex), uv_row, float(depth_m), pick,
                                                   image_hw, templates, dot))
    return records


def depth_estimation_records(scene, scene_id: str, image_hw, max_samples: int = -1,
                             templates: T.TemplateSet = T.DEPTH_ESTIMATION, rng=_random) -> List[dict]:
    counts = scene._visibility()["count"].cpu().numpy()
    n_visible = {k: int(c) for k, c in zip(scene.ids, counts)}
    return depth_estimation_records_fn(scene_id, scene.ids, n_visible, gpu_point_numerics(scene), image_hw, max_samples,
                                       templates, rng)


# --------------------------------------------------------------------------------------------
# depth comparison by coordinate (DC_C = depth_perception/depth_comparison_coor_engine.py:231-343)
# --------------------------------------------------------------------------------------------
def _depth_comparison_images(image_ids: Sequence[str], max_samples: int, rng) -> List[str]:
    """DC_C:237-246: with replacement when more samples than images are asked for."""
    ids = list(image_ids)
    if max_samples > 0:
        if max_samples > len(ids):
            return rng.choices(ids, k=max_samples)
        return rng.sample(ids, max_samples)
    return rng.sample(ids, len(ids))


def depth_comparison_records(scene_id: str, image_ids: Sequence[str], n_visible: Dict[str, int], numeric_fn, image_hw,
                             max_samples: int = -1, templates: T.TemplateSet = None, rng=_random,
                             max_n_points_per_image: int = 1, on_skip=None, dot: bool = False, on_mark=None,
                             dry_run: bool = False) -> List[dict]:
    """Records of DC_C.generate_qa_training_single_scene (``dot``: of DC_D's, depth_comparison_dot_engine.py:240-375 --
    lettered discs instead of coordinates; ``on_mark(scene_id, image_id, vertices, points_info, colours)`` gets them).

    ``numeric_fn([(image_id, position), ...]) -> [(vertex, uv_row | None, depth_m), ...]`` resolves the
    position-th visible vertex of an image and projects it (K6a + K6b for the whole batch on the GPU).
    Upstream skips a pair whose two points round to the same depth *before* drawing its templates, so the
    position of later draws in the ``random`` stream depends on the numerics.  The loop below therefore
    speculates: it draws for all remaining pairs as if none were skipped, evaluates them in one batch,
    accepts everything up to the first skipped pair, rewinds the generator to just after that pair's vertex
    draw and goes on from there.  Skips are rare (equal millimetre depths), so this is one batch in practice.
    ``dry_run``: only the draws of a scene WITHOUT a skipped pair are made (no numerics, no records): where the generator
    stands after such a scene -- what a sharded builder needs to start the next scene before this one has been evaluated.
    """
    templates = templates or T.DEPTH_COMPARISON
    H, W = image_hw
    sampled = _depth_comparison_images(image_ids, max_samples, rng)
    slots = [(img, r) for img in sampled for r in range(max_n_points_per_image)]
    records: List[dict] = []
    start = 0
    retries_left = 10 if dot else 0          # DC_D re-draws a tied pair up to 10 more times (DC_D:263-309); DC_C gives up at once
    while start < len(slots):
        plan = []
        for img, _ in slots[start:]:
            state = rng.getstate()
            pos = sample_indices(int(n_visible[img]), 2, rng)                 # random.sample(visible_points, 2)
            after_pick = rng.getstate()
            letters = ["A", "B"]
            rng.shuffle(letters)
            order = sample_indices(2, 2, rng)                                 # random.sample(points_info, 2)
            closer_q = rng.choice([True, False])
            kind = "closer" if closer_q else "farther"
            qi = rng.choice(range(len(templates.questions[kind])))
            ai = rng.choice(range(len(templates.answers[kind])))
            ti = rng.choice(range(len(templates.task_description)))
            colors = [(rng.randint(0, 255), rng.randint(0, 255), rng.randint(0, 255)) for _ in range(2)] if dot else []  # DC_D:335-336
            plan.append({"image_id": img, "pos": pos, "after_pick": after_pick, "letters": letters, "order": order,
                         "closer_q": closer_q, "kind": kind, "picks": (qi, ai, ti), "state": state, "colors": colors})
        if dry_run:
            return []
        numerics = numeric_fn([(p["image_id"], j) for p in plan for j in p["pos"]])
        skipped_at = None
        for n, p in enumerate(plan):
            info, verts = [], []
            for i in range(2):
                vertex, uv_row, depth_m = numerics[2 * n + i]
                verts.append(int(vertex))
                if uv_row is None:                                            # DC_C:260-266
                    continue
                info.append({"x": round((uv_row[0] / W) * 1000), "y": round((uv_row[1] / H) * 1000),
                             "depth": round(depth_m * 1000), "coords": (int(uv_row[0]), int(uv_row[1])),
                             "letter": chr(65 + i)})
            if len(info) != 2 or info[0]["depth"] == info[1]["depth"]:        # DC_C:279-284
                if on_skip is not None:
                    on_skip(scene_id, p["image_id"], verts)
                skipped_at = n
                break
            shuffled = [info[k] for k in p["order"]]
            for i, pi in enumerate(shuffled):
                pi["letter"] = p["letters"][i]
            p1, p2 = shuffled
            closer = p1 if p1["depth"] <= p2["depth"] else p2
            farther = p2 if p1["depth"] <= p2["depth"] else p1
            correct = closer if p["closer_q"] else farther
            qi, ai, ti = p["picks"]
            if dot:
                question = templates.questions[p["kind"]][qi]
                answer = templates.answers[p["kind"]][ai].format(correct_label=correct["letter"])
                if on_mark is not None:
                    on_mark(scene_id, p["image_id"], verts, shuffled, p["colors"])
            else:
                question = templates.questions[p["kind"]][qi].format(x1=p1["x"], y1=p1["y"], x2=p2["x"], y2=p2["y"])
                answer = templates.answers[p["kind"]][ai].format(correct_x=correct["x"], correct_y=correct["y"])
            records.append({
                "id": f"{scene_id}_{p['image_id']}_p{verts[0]}_p{verts[1]}",
                "image": [f"{scene_id}/{p['image_id']}_p{verts[0]}_p{verts[1]}_annotated.jpg" if dot
                          else f"{scene_id}/{p['image_id']}.jpg"],
                "conversations": [{"from": "human", "value": f"{templates.task_description[ti]}\n{question}"},
                                  {"from": "gpt", "value": answer}],
                "height_list": [H],
                "width_list": [W],
                "question_type": "depth_comparison_annotated" if dot else "depth_comparison_coordinate",
                "gt_value": correct["letter"] if dot else [correct["x"], correct["y"]],
                "points_info": shuffled,
                "is_closer_question": p["closer_q"],
            })
        if skipped_at is None:
            break
        rng.setstate(plan[skipped_at]["after_pick"])       # the skipped pair consumed its vertex draw only
        if skipped_at > 0:
            retries_left = 10 if dot else 0                # a different slot than the one that was being retried
        start += skipped_at
        if retries_left > 0:
            retries_left -= 1                              # same slot again, with a fresh pair
        else:
            start += 1                                     # give up on this slot
            retries_left = 10 if dot else 0
    return records


def gpu_point_numerics(scene):
    """numeric_fn for the depth heads: position-th visible vertex of an image (K6a) and its projection (K6b)."""
    import torch
    from . import engine

    def fn(samples):
        if not samples:
            return []
        vis = scene._visibility()
        sel = torch.tensor([[scene.index[i], scene.index[i], j] for i, j in samples], dtype=torch.int32, device=scene.device)
        vert = engine.select_common_point(vis["bits"], sel)
        uv, d, ok = engine.project_samples(scene.xyz, scene.cam_mats, scene.depth, scene.image_hw,
                                           torch.stack([vert, sel[:, 0]], 1).contiguous(), scene.depth_scale)
        uv, d, ok, vert = uv.cpu().numpy(), d.cpu().numpy(), ok.cpu().numpy().astype(bool), vert.cpu().numpy()
        return [(int(vert[s]), uv[s] if ok[s] else None, float(d[s])) for s in range(len(samples))]
    return fn


def list_point_numerics(scene, visible_points):
    """numeric_fn over the on-disk visibility index: ``visible_points(image_id)`` is the image's vertex list as the
    reference reads it (VisibilityInfoHandler.get_image_to_points_info); projection + visibility check on the GPU (K6b)."""
    import torch
    from . import engine

    def fn(samples):
        if not samples:
            return []
        vert = [int(visible_points(i)[j]) for i, j in samples]
        smp = torch.tensor([[v, scene.index[i]] for v, (i, _) in zip(vert, samples)], dtype=torch.int32, device=scene.device)
        uv, d, ok = engine.project_samples(scene.xyz, scene.cam_mats, scene.depth, scene.image_hw, smp, scene.depth_scale)
        uv, d, ok = uv.cpu().numpy(), d.cpu().numpy(), ok.cpu().numpy().astype(bool)
        return [(vert[s], uv[s] if ok[s] else None, float(d[s])) for s in range(len(samples))]
    return fn


def depth_comparison_records_gpu(scene, scene_id: str, image_hw, max_samples: int = -1, templates: T.TemplateSet = None,
                                 rng=_random, on_skip=None) -> List[dict]:
    counts = scene._visibility()["count"].cpu().numpy()
    n_visible = {k: int(c) for k, c in zip(scene.ids, counts)}
    return depth_comparison_records(scene_id, scene.ids, n_visible, gpu_point_numerics(scene), image_hw, max_samples,
                                    templates, rng, on_skip=on_skip)


# --------------------------------------------------------------------------------------------
# object movement on TAPVid-3D tracks (OM_C:317-404)
# --------------------------------------------------------------------------------------------
def object_movement_record(scene_id: str, frame1: int, frame2: int, point_index: int, question_type: str,
                           numeric: dict, image_hw, templates: T.TemplateSet = T.OBJECT_MOVEMENT, rng=_random,
                           dot: bool = False, needs_annotation=None, on_mark=None) -> Optional[dict]:
    """numeric: distance, vector (camera-1 axes, metres), point_moving, cam_moving, p1n/p2n (normalised
    projections or None) -- K5's outputs for this (frame1, frame2, point).

    ``dot`` (single_object_movement_engine_dot.py:341-436, OM_D): the first frame carries a disc on the point; its colour
    is drawn after the templates and only when the annotated file is still to be made (``needs_annotation(name)``,
    OM_D:409-413), ``on_mark(frame1, frame2, point, pixel, colour_or_None)`` receives the job."""
    if numeric["p1n"] is None or numeric["p2n"] is None:                  # OM_C:360-362
        return None
    H, W = image_hw
    x1, y1 = round(numeric["p1n"][0] * 1000), round(numeric["p1n"][1] * 1000)
    x2, y2 = round(numeric["p2n"][0] * 1000), round(numeric["p2n"][1] * 1000)
    task_description = rng.choice(templates.task_description)             # OM_C:367-375
    question = rng.choice(templates.questions[question_type]).format(x1=x1, y1=y1)
    vec = numeric["vector"]
    answer_text = rng.choice(templates.answers[question_type]).format(
        total_distance=round(numeric["distance"] * 1000), x_value=round(vec[0] * 1000), y_value=round(vec[1] * 1000),
        z_value=round(vec[2] * 1000))
    if not numeric["point_moving"]:
        answer_text = "The point did not move. " + answer_text
    images = [f"{scene_id}/{frame:05d}.jpg" for frame in [frame1, frame2]]
    if dot:
        from .annotate import generate_distinct_colors
        name = f"{frame1:05d}_{point_index}_annotated.jpg"
        color = None
        if needs_annotation is None or needs_annotation(name):
            color = generate_distinct_colors(1, rng)[0]
        if on_mark is not None:
            on_mark(frame1, frame2, point_index, (int(numeric["p1n"][0] * W), int(numeric["p1n"][1] * H)), color)
        images = [f"{scene_id}/{name}", f"{scene_id}/{frame2:05d}.jpg"]
    return {
        "id": f"{scene_id}_{frame1}_{frame2}_{point_index}" + ("_ann" if dot else ""),
        "image": images,
        "conversations": [{"from": "human", "value": f"{task_description}\n{question}"},
                          {"from": "gpt", "value": answer_text}],
        "height_list": [H] * 2,
        "width_list": [W] * 2,
        # OM_D:429 tests ``== "total_distance"`` against the "tapvid3d_..." names, so its dot records always carry the vector
        "gt_value": (int(numeric["distance"] * 1000) if (question_type == "total_distance" if dot else "total_distance" in question_type)
                     else list(vec)),
        "question_type": question_type,
        "point_moving": int(numeric["point_moving"]),
        "cam_moving": int(numeric["cam_moving"]),
        "p1": (x1, y1),
        "p2": (x2, y2),
    }


def object_movement_numeric(tracks_xyz: np.ndarray, extrinsics_w2c: np.ndarray, fx_fy_cx_cy, image_hw,
                            triples: np.ndarray, device="cuda", obj_threshold: float = 0.01,
                            cam_threshold: float = 0.01) -> List[dict]:
    """GPU stage: K5a + K5b for a list of (frame1, frame2, point) triples."""
    import torch
    from . import engine
    T_, P, _ = tracks_xyz.shape
    tr = torch.from_numpy(np.ascontiguousarray(tracks_xyz, dtype=np.float64)).to(device)
    w2c_np = np.ascontiguousarray(extrinsics_w2c, dtype=np.float64)
    c2w = torch.from_numpy(np.linalg.inv(w2c_np).reshape(T_, 16)).to(device)          # OM_C:448, host LAPACK
    w2c = torch.from_numpy(w2c_np.reshape(T_, 16)).to(device)
    res = engine.track_to_world(tr, c2w, fx_fy_cx_cy, image_hw)
    trip = torch.from_numpy(np.ascontiguousarray(triples, dtype=np.int32)).to(device)
    disp, flags = engine.track_displacement(res["world"], w2c, c2w, trip, obj_threshold, cam_threshold)
    uvn, ok = res["uvn"].cpu().numpy(), res["ok"].cpu().numpy().astype(bool)
    disp, flags = disp.cpu().numpy(), flags.cpu().numpy()
    out = []
    for k, (f1, f2, p) in enumerate(np.asarray(triples)):
        out.append({"distance": float(disp[k, 0]), "vector": disp[k, 1:4].tolist(),
                    "point_moving": bool(flags[k, 0]), "cam_moving": bool(flags[k, 1]),
                    "p1n": uvn[f1, p].tolist() if ok[f1, p] else None,
                    "p2n": uvn[f2, p].tolist() if ok[f2, p] else None})
    return out


def object_movement_records(scene_id: str, tracks_xyz, extrinsics_w2c, fx_fy_cx_cy, image_hw, sample_pairs: Sequence[dict],
                            question_type: str, templates: T.TemplateSet = T.OBJECT_MOVEMENT, rng=_random,
                            device="cuda", dot: bool = False, needs_annotation=None, on_mark=None,
                            obj_threshold: float = 0.01, cam_threshold: float = 0.01) -> List[dict]:
    """format_training_samples (OM_C:317-404; OM_D:341-436 with ``dot``) for already chosen {frame1, frame2, point_index}
    samples."""
    triples = np.array([[s["frame1"], s["frame2"], s["point_index"]] for s in sample_pairs], dtype=np.int32).reshape(-1, 3)
    numeric = object_movement_numeric(tracks_xyz, extrinsics_w2c, fx_fy_cx_cy, image_hw, triples, device, obj_threshold,
                                      cam_threshold)
    records = []
    for s, num in zip(sample_pairs, numeric):
        r = object_movement_record(scene_id, int(s["frame1"]), int(s["frame2"]), int(s["point_index"]), question_type, num,
                                   image_hw, templates, rng, dot, needs_annotation, on_mark)
        if r is not None:
            records.append(r)
    return records


# --------------------------------------------------------------------------------------------
# object perception (OPE = object_perception/single_object_perception_engine.py)
# --------------------------------------------------------------------------------------------
def object_perception_records(dim_info: Dict[str, Dict], dimension_name: str, value_m, category, image_hw,
                              max_k: int = 6, templates: T.TemplateSet = T.OBJECT_PERCEPTION,
                              rng=_random) -> Dict[int, List[dict]]:
    """{k: records} from a coverage table {scene: {object: {k: [image combinations]}}} (OPE:126-213).

    ``value_m(scene_id, object_id)`` is the ground-truth size in metres, ``category(scene_id, object_id)``
    the raw category, ``image_hw`` (H, W) or a callable scene_id -> (H, W).  Draw order per record: shuffle of the combination, task line, question, answer."""
    by_k: Dict[int, List[dict]] = {k: [] for k in range(1, max_k + 1)}
    for scene_id, objects in dim_info.items():
        H, W = image_hw(scene_id) if callable(image_hw) else image_hw
        for object_id, k_table in objects.items():
            val_mm = int(round(value_m(scene_id, object_id) * 1000))
            cat = category(scene_id, object_id)
            for k_key, combos in k_table.items():
                try:
                    k = int(k_key)
                except (TypeError, ValueError):
                    continue
                if k < 1 or k > max_k:
                    continue
                for combo_idx, combo in enumerate(combos):
                    if not combo:
                        continue
                    combo = list(combo)
                    rng.shuffle(combo)
                    prefix = "\n".join(f"Image-{i}: <image>" for i in range(1, len(combo) + 1))
                    task_line = rng.choice(templates.task_description)
                    question = rng.choice(templates.questions["default"]).format(dimension=dimension_name, object_category=cat)
                    answer = rng.choice(templates.answers["default"]).format(dimension=dimension_name, value_mm=val_mm,
                                                                             object_category=cat)
                    by_k[k].append({
                        "id": f"{scene_id}_{object_id}_{k}_{combo_idx}",
                        "image": [f"{scene_id}/{img}.jpg" for img in combo],
                        "conversations": [{"from": "human", "value": f"{prefix}\n{task_line}\n{question}"},
                                          {"from": "gpt", "value": answer}],
                        "height_list": [H] * len(combo),
                        "width_list": [W] * len(combo),
                        "question_type": f"object_perception_{dimension_name}_estimation",
                        "gt_value": val_mm,
                    })
    return by_k


def object_movement_mine_pairs(visibility: np.ndarray, groups: Sequence[Sequence[int]], distance_fn,
                               npoints_per_group: int = 5, npairs_per_bin=1e8, augment: bool = True,
                               augment_ratio: float = 1.0, rng=_random, object_not_moving_threshold: float = 0.01,
                               future_frame_windows: float = 1e8) -> List[dict]:
    """Frame-pair mining of OM_C.generate_qa_training_single_scene (OM_C:465-567): per rigid group a few points,
    per point one static pair plus distance-binned moving pairs, then the swap augmentation.

    ``distance_fn(points, visible_frame_lists) -> [float64 array per point]`` gives, for each point, the world-space
    distance between every two of its visible frames (i < j, row-major) -- K5c on the GPU, one launch per group.
    Draw order, stable sort, bin edges and the carried-over ``npairs_per_bin`` follow upstream; so does its frame
    window test, which compares the *distance* and the first frame index (OM_C:505-507) and therefore never fires
    with the default window.
    """
    sample_pairs: List[dict] = []
    for group in groups:
        rng.shuffle(group)                                                   # in place, as upstream (OM_C:468)
        selected = list(group[:npoints_per_group])
        vis_frames = [np.where(visibility[:, p])[0] for p in selected]
        todo = [k for k, v in enumerate(vis_frames) if len(v) >= 2]
        dists = dict(zip(todo, distance_fn([selected[k] for k in todo], [vis_frames[k] for k in todo]))) if todo else {}
        for k, point_idx in enumerate(selected):
            if k not in dists:
                continue
            frames = vis_frames[k]
            ii, jj = np.triu_indices(len(frames), 1)                         # (i, j), i < j, row-major (OM_C:482-483)
            d = np.asarray(dists[k], dtype=np.float64)
            f1, f2 = frames[ii], frames[jj]
            keep = ~(f1 > d + future_frame_windows)                          # OM_C:505-507 (sic)
            static = np.where(keep & (d < object_not_moving_threshold))[0]
            moving = np.where(keep & ~(d < object_not_moving_threshold))[0]
            chosen: List[int] = []
            if len(static):
                chosen.append(int(static[rng.choice(range(len(static)))]))
            if len(moving):
                moving = moving[np.argsort(d[moving], kind="stable")]        # list.sort is stable
                dm = d[moving]
                edges = np.histogram_bin_edges(dm.tolist(), bins=10)
                which = np.minimum(np.digitize(dm, edges) - 1, 9)
                bins = [moving[which == b] for b in range(10)]
                npairs_per_bin = max(min(len(bins[4]), npairs_per_bin), 1)   # carried over to later points (OM_C:535-537)
                for members in bins:
                    if len(members) > npairs_per_bin:
                        members = members[sample_indices(len(members), int(npairs_per_bin), rng)]
                    chosen.extend(int(m) for m in members)
            for m in chosen:
                sample_pairs.append({"point_index": point_idx, "frame1": f1[m], "frame2": f2[m]})
    if augment:
        n_aug = int(len(sample_pairs) * augment_ratio)
        for m in sample_indices(len(sample_pairs), n_aug, rng):
            s = sample_pairs[m]
            sample_pairs.append({"point_index": s["point_index"], "frame1": s["frame2"], "frame2": s["frame1"]})
    return sample_pairs
