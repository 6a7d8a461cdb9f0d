"""Shared scaffolding of the two coordinate depth engines (reference: depth_estimation_coor_engine.py and
depth_comparison_coor_engine.py share their constructor, scene sampling and writers line for line)."""
from __future__ import annotations

import os
import random

from mspa import heads
from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler, VisibilityInfoHandler
from mspa.hostinfo import quietly


class _LazyCounts:
    """len(visible points) per image, read from the visibility index only for the images that are sampled."""

    def __init__(self, lists):
        self._lists = lists

    def __getitem__(self, image_id):
        return len(self._lists(image_id))


class DepthCoorEngineBase:
    task_name = ""
    TEMPLATE_SET = None

    def __init__(self, scene_info_path, version_num="v1_0", all_max_samples=-1, image_output_dir=None,
                 visibility_info_path=None, max_n_points_per_image=1, warning_file=None):
        self.scene_info = SceneInfoHandler(scene_info_path)
        self.version_num = version_num
        self.image_output_dir = image_output_dir
        self.all_max_samples = all_max_samples
        self.max_n_points_per_image = max_n_points_per_image
        self.warning_file = warning_file
        self.visibility_info = VisibilityInfoHandler(visibility_info_path)
        self.max_samples = -1
        self.templates = self.TEMPLATE_SET
        self.annotator = None          # dot variants: mspa.annotate.PillowAnnotator() unless the caller sets another one

    # -- helpers ------------------------------------------------------------------------------
    def _warn(self, message):
        print(message.strip())
        sink = getattr(self, "_warn_sink", None)
        if sink is not None:                       # a sharded build: the owner rank collects, rank 0 writes in scene order
            sink.append(message.strip())
            return
        if self.warning_file:
            with open(self.warning_file, "a") as wf:
                wf.write(message.strip())

    def _visible_points_of(self, scene_id):
        cache = {}

        def visible_points(image_id):
            if image_id not in cache:
                cache[image_id] = self.visibility_info.get_image_to_points_info(scene_id, image_id)
            return cache[image_id]
        return visible_points

    def _scene_inputs(self, scene_id, scene=None):
        visible_points = self._visible_points_of(scene_id)
        scene = scene if scene is not None else self.scene_info.scene_on_device(scene_id)
        return (self.scene_info.get_all_extrinsic_valid_image_ids(scene_id), _LazyCounts(visible_points),
                heads.list_point_numerics(scene, visible_points), self.scene_info.get_image_shape(scene_id))

    def _annotated_path(self, scene_id, name):
        return os.path.join(self.image_output_dir, scene_id, name)

    def _annotator(self):
        if self.annotator is None:
            from mspa.annotate import PillowAnnotator
            self.annotator = PillowAnnotator()
        return self.annotator

    def generate_qa_training_single_scene(self, scene_id):
        raise NotImplementedError

    # -- sharded build of the engines whose draws do not depend on the kernels (the two estimation engines) ------------------
    DRAWS_AHEAD = False                   # subclasses that implement _scene_draws / _scene_records_on set this

    def _scene_draws(self, scene_id):
        raise NotImplementedError

    def _scene_records_on(self, scene, scene_id, draws):
        raise NotImplementedError

    def _sharded_scene_records(self, scene_ids, ctx, num_workers=8):
        """All scenes' records on rank 0, in scene order, for a job with one process per GPU (upstream: one process, one scene
        after the other, DE_C:256-262).  Every rank makes the draws of EVERY scene in order -- they read the visibility index
        only, so the ``random`` stream ends where a single process leaves it, on every rank; the scenes are then dealt over
        the ranks (``mspa.sweep.sharded_sweep``: longest-first by draws, prefetched, depth frames decoded on the device) and
        each rank projects, checks, words -- and for the dot engine draws the annotated JPEGs of -- its own scenes' records;
        JSON lines and warning lines travel to rank 0 per window."""
        import json
        from mspa import sweep
        all_draws = [self._scene_draws(s) for s in scene_ids]
        costs = [float(sum(len(d["positions"]) for d in dr) + 1) for dr in all_draws]
        records = []

        def work_items(indices):
            return self.scene_info.prefetched_scenes([scene_ids[i] for i in indices], num_workers, ctx.device)

        def produce(index, scene):
            self._warn_sink = []
            try:
                recs = self._scene_records_on(scene, scene_ids[index], all_draws[index])
                warned = list(self._warn_sink)
            finally:
                self._warn_sink = None
            return None, ["".join(json.dumps(r) + "\n" for r in recs).encode(), "\0".join(warned).encode()]

        def consume(index, _rows, blobs):
            records.extend(heads.JsonLine(line) for line in bytes(blobs[0]).split(b"\n")[:-1])
            if blobs[1].size and self.warning_file:
                with open(self.warning_file, "a") as wf:
                    for w in bytes(blobs[1]).decode().split("\0"):
                        wf.write(w)

        sweep.sharded_sweep(costs, ctx, work_items, produce, consume)
        return records

    # -- sharded build of the engines whose draws DO depend on the kernels (the two comparison engines) -------------------------
    CHAINED = False                       # subclasses that implement _scene_records_on(scene, scene_id, None, dry_run=...) set this

    def _chained_scene_records(self, scene_ids, ctx, num_workers=8):
        """All scenes' records on rank 0, in scene order, when a scene's draws depend on its numerics: upstream skips a pair of
        equal millimetre depths BEFORE drawing its templates (DC_C:279-284), so where scene k + 1 starts in the ``random``
        stream is only known once scene k has been evaluated -- about once in ten scenes differently from the no-skip guess.

        Scenes are taken in windows of one per rank (rank r owns every ``world``-th scene and streams them through the
        prefetcher).  Per window every rank computes, on the host, where each scene would START if no pair in front of it were
        skipped (``dry_run`` draws: they read the visibility index only); each rank evaluates its own scene exactly from that
        guess; one all_gather says whose guess was wrong.  Scenes up to the first wrong one are final; its owner's true end
        state is broadcast and only the scenes behind it -- still resident on their ranks -- are redone.  Without a skip a
        window costs one small all_gather and one gather of the finished lines."""
        import json
        import torch
        import torch.distributed as dist
        from mspa import shard
        rank, world = ctx.rank, ctx.world
        my_ids = scene_ids[rank::world]
        my_scenes = iter(self.scene_info.prefetched_scenes(my_ids, num_workers, ctx.device)) if my_ids else iter(())
        records = []
        state = random.getstate()                                  # exact, the same on every rank
        for w0 in range(0, len(scene_ids), world):
            window = scene_ids[w0:w0 + world]
            mine = rank if rank < len(window) else None
            failure, scene = None, None
            try:
                scene = next(my_scenes) if mine is not None else None
            except Exception as e:                                  # a scene that cannot be read: every rank leaves together
                failure = e
            final_upto, result = 0, ([], [])
            while final_upto < len(window):
                random.setstate(state)
                starts = []
                for j in range(final_upto, len(window)):            # the no-skip guess of every open scene's start
                    starts.append(random.getstate())
                    if failure is None:
                        try:
                            self._scene_records_on(None, window[j], None, dry_run=True)
                        except Exception as e:
                            failure = e
                starts.append(random.getstate())
                flag, end_state = 0, None
                if failure is None and mine is not None and mine >= final_upto:
                    random.setstate(starts[mine - final_upto])
                    self._warn_sink = []
                    try:
                        recs = self._scene_records_on(scene, window[mine], None)
                        result = (recs, list(self._warn_sink))
                        end_state = random.getstate()
                        flag = int(end_state != starts[mine - final_upto + 1])
                    except Exception as e:
                        failure = e
                    finally:
                        self._warn_sink = None
                flags = torch.tensor([2 if failure is not None else flag], dtype=torch.int32, device=ctx.collective_device)
                gathered = [torch.zeros_like(flags) for _ in range(world)]
                dist.all_gather(gathered, flags, group=ctx.group)
                gathered = [int(g.item()) for g in gathered]
                if 2 in gathered:
                    if failure is not None:
                        raise failure
                    raise RuntimeError("depth comparison (sharded): another rank failed in this window (its own traceback says why)")
                wrong = [j for j in range(final_upto, len(window)) if gathered[j]]
                if not wrong:
                    state = starts[-1]
                    final_upto = len(window)
                else:                                               # its owner evaluated it from a correct start: its end is exact
                    state = shard.broadcast_object(end_state, ctx, src=wrong[0])
                    final_upto = wrong[0] + 1
                    if mine is not None and mine > wrong[0]:
                        result = ([], [])                           # started from a wrong guess: redone in the next round
            payload = json.dumps({"lines": [json.dumps(r) for r in result[0]], "warned": result[1]}).encode()
            parts = shard.gather_bytes(payload, ctx, dst=0)
            if rank == 0:
                for part in parts[:len(window)]:
                    got = json.loads(bytes(part).decode()) if len(part) else {"lines": [], "warned": []}
                    records.extend(heads.JsonLine(line.encode()) for line in got["lines"])
                    if got["warned"] and self.warning_file:
                        with open(self.warning_file, "a") as wf:
                            for msg in got["warned"]:
                                wf.write(msg)
        for _ in my_scenes:
            pass
        random.setstate(state)
        return records

    # -- dataset level (reference: generate_qa_training_data / generate_qa_eval_data) --------------
    @quietly
    def generate_qa_training_data(self, output_dir, save_file=True):
        from mspa import shard
        ctx = shard.context_from_env() if (self.DRAWS_AHEAD or self.CHAINED) else None
        scene_ids = self.scene_info.get_sorted_keys()
        if self.all_max_samples > 0:
            self.max_samples = max(self.all_max_samples // len(scene_ids) + 1, 1)
            if self.max_samples == 1:
                scene_ids = random.sample(scene_ids, self.all_max_samples)
        else:
            self.max_samples = -1
        self.num_used_scenes = len(scene_ids)
        if ctx is not None:
            train_data = self._chained_scene_records(scene_ids, ctx) if self.CHAINED else self._sharded_scene_records(scene_ids, ctx)
            n = int(shard.broadcast_object(len(train_data), ctx, src=0))
            order = list(range(n))                 # the sample and the shuffle as index lists: every rank's generator stays in step
            if n > self.all_max_samples:
                order = random.sample(order, self.all_max_samples)
            random.shuffle(order)
            ctx.barrier()
            if ctx.rank != 0:
                return [] if not save_file else None
            train_data = [train_data[i] for i in order]
            if not save_file:
                import json
                return [json.loads(line) for line in train_data]
        else:
            train_data = []
            for scene_id in scene_ids:
                train_data.extend(self.generate_qa_training_single_scene(scene_id))
            if len(train_data) > self.all_max_samples:
                train_data = random.sample(train_data, self.all_max_samples)
            random.shuffle(train_data)
            if not save_file:
                return train_data
        os.makedirs(output_dir, exist_ok=True)
        path = f"{output_dir}/{self.task_name}.jsonl"
        heads.write_jsonl(path, train_data)
        print(f"[Train] Training data saved to {path}. Generated {len(train_data)} samples in total.")

    def convert_train_sample_to_eval_sample(self, train_sample):
        train_sample["text"] = train_sample["conversations"][0]["value"]      # upstream keeps `conversations` here
        return train_sample

    @quietly
    def generate_qa_eval_data(self, output_dir):
        assert self.max_n_points_per_image == 1, "max_n_points_per_image should be 1 for evaluation"
        from mspa import shard
        data = [self.convert_train_sample_to_eval_sample(s) for s in self.generate_qa_training_data(output_dir, save_file=False)]
        ctx = shard.context_from_env() if (self.DRAWS_AHEAD or self.CHAINED) else None
        if ctx is not None and ctx.rank != 0:
            return
        os.makedirs(output_dir, exist_ok=True)
        path = f"{output_dir}/{self.task_name}.jsonl"
        heads.write_jsonl(path, data)
        print(f"[Eval] Evaluation data saved to {path}. Generated {len(data)} samples in total.")


def run_cli(engine_cls, task_dir, argv=None):
    """The reference scripts' __main__: eval split first, then train, same defaults and paths."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--train_scene_info_path", default="data/scannet/scannet_instance_data/scenes_train_info_i_D5.pkl")
    ap.add_argument("--val_scene_info_path", default="data/scannet/scannet_instance_data/scenes_val_info_i_D5.pkl")
    ap.add_argument("--train_all_max_samples", type=int, default=500000)
    ap.add_argument("--val_all_max_samples", type=int, default=300)
    ap.add_argument("--output_dir_train", default=f"training_data/{task_dir}")
    ap.add_argument("--output_dir_val", default=f"evaluation_data/{task_dir}")
    ap.add_argument("--version_num", default="v1_0")
    args = ap.parse_args(argv)
    out_train = os.path.join(args.output_dir_train, args.version_num)
    out_val = os.path.join(args.output_dir_val, args.version_num)
    os.makedirs(out_train, exist_ok=True)
    os.makedirs(out_val, exist_ok=True)
    root = "data/scannet/scannet_instance_data"
    engine_cls(args.val_scene_info_path, args.version_num, args.val_all_max_samples, os.path.join(out_val, "images"),
               f"{root}/val_visibility_info_D5.parquet", warning_file=f"{out_val}/val_warning.txt").generate_qa_eval_data(out_val)
    engine_cls(args.train_scene_info_path, args.version_num, args.train_all_max_samples, os.path.join(out_train, "images"),
               f"{root}/train_visibility_info_D5.parquet",
               warning_file=f"{out_train}/train_warning.txt").generate_qa_training_data(out_train)
