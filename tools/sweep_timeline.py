#!/usr/bin/env python3
"""Per-scene timeline of a one-rank from-disk sweep with the depth decode on the device: when each scene was admitted, staged,
copied, inflated, un-filtered, taken by the consumer and released -- the question being what a slot is held FOR, since the
sweep's rate is slots / hold time once no single stage is saturated.

    python tools/sweep_timeline.py [--scenes 96] [--frames 320] [--entry cfr|mvi] [--passes 3]

Hooks (this tool only; nothing in the package changes): UploadSlot.stage_and_upload / _upload_and_decode / finish_decode and
ScenePrefetcher._consume are wrapped; device times come from events recorded on the slot's stream against one base event."""
import argparse, contextlib, io, json, os, sys, tempfile, shutil, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT, os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=96)
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--points", type=int, default=131072)
    ap.add_argument("--entry", default="cfr")
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--per-rank", type=int, default=8)
    ap.add_argument("--smooth", action="store_true")
    ap.add_argument("--brief", action="store_true", help="only the summary line")
    ap.add_argument("--nogc", action="store_true", help="gc.disable() for the passes (is the once-per-pass stall the collector?)")
    ap.add_argument("--gc", default="", help="freeze: gc.freeze() before the passes; freeze+thr: and gen-0 threshold 50 000")
    ap.add_argument("--series", action="store_true", help="per-scene series of the device-side durations")
    ap.add_argument("--switch", type=float, default=None, help="sys.setswitchinterval (seconds; the interpreter's default is 0.005)")
    a = ap.parse_args()
    os.environ.setdefault("MSPA_DEPTH_DECODE", "device")
    os.environ["MSPA_WINDOW_PER_RANK"] = str(a.per_rank)
    os.environ.setdefault("MSPA_LOOKAHEAD", "2")
    if a.switch is not None:
        sys.setswitchinterval(a.switch)
    import numpy as np, torch
    import dropin_ranks
    from mspa import sweep, upload
    from mspa.scene import SceneOnDevice
    import spatial_engine.camera_movement.calculate_frames_relations as CFR
    import spatial_engine.utils.scannet_utils.make_visibility_info as MVI
    from spatial_engine.utils.scannet_utils.handler import info_handler as IH
    root = tempfile.mkdtemp(prefix="mspa_timeline_")
    try:
        paths = dropin_ranks.write_inputs(root, a.scenes, a.frames, a.points, smooth=a.smooth)
        orig_init = IH.SceneInfoHandler.__init__
        IH.SceneInfoHandler.__init__ = lambda self, info_path, *x, **k: orig_init(
            self, info_path, posed_images_root=paths["posed_images_root"], instance_data_root=paths["instance_data_root"])
        log, lock = [], threading.Lock()
        base = {"ev": None, "t": None}
        now = time.perf_counter

        def rec(**kw):
            with lock:
                log.append(kw)

        o_stage, o_up, o_fin, o_cons = (upload.UploadSlot.stage_and_upload, upload.UploadSlot._upload_and_decode,
                                        upload.UploadSlot.finish_decode, upload.ScenePrefetcher._consume)

        fr = {}
        o_fr = SceneOnDevice.from_resident                      # (bound to the class)

        def from_res(*x, **k):
            t = now()
            out = o_fr(*x, **k)
            fr["ms"] = (now() - t) * 1e3
            return out
        SceneOnDevice.from_resident = staticmethod(from_res)

        def stage(self, sc, cs):
            t0 = now()
            free_wait = 0.0
            if self.free is not None:
                self.free.synchronize()
                free_wait = now() - t0
            fr.clear()
            r = o_stage(self, sc, cs)
            self._tl = {"from_resident_ms": fr.get("ms", 0.0), "up0": self._up[0], "up1": self._up[1], "scene": getattr(sc, "scene_id", None), "stage0": t0, "stage1": now(), "free_wait": free_wait, "ev": getattr(self, "_ev", None)}
            return r

        def up(self, packed, F, stream):
            t_up0 = now()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            h, w = packed.hw
            from mspa import engine
            self._ensure_decode(F, (h, w), packed.capacity)
            self.h_off[:F] = torch.from_numpy(packed.offsets)
            self.h_nb[:F] = torch.from_numpy(np.where(packed.status == 0, packed.nbytes, 0))
            with torch.cuda.stream(stream):
                evs[0].record(stream)
                self.d_comp[:packed.capacity].copy_(packed.buf[:packed.capacity], non_blocking=True)
                self.d_off[:F].copy_(self.h_off[:F], non_blocking=True)
                self.d_nb[:F].copy_(self.h_nb[:F], non_blocking=True)
                evs[1].record(stream)
                engine.inflate_blocks_device(self.d_comp, self.d_off[:F], self.d_nb[:F], h * (2 * w + 1), self.d_raw, self.d_status)
                evs[2].record(stream)
                engine.png_unfilter_device(self.d_raw[:F], h, w, self.d_status, self.d_depth[:F])
                evs[3].record(stream)
                self.h_status[:F].copy_(self.d_status[:F], non_blocking=True)
            self.pending_decode = (packed, F)
            self._ev = (evs, now())
            self._up = (t_up0, self._ev[1])

        def fin(self):
            t0 = now()
            had = self.pending_decode is not None
            o_fin(self)
            if had:
                self._tl["fin0"], self._tl["fin1"] = t0, now()

        def cons(item, free_slots):
            scene, slot = item
            for x in o_cons(item, free_slots):
                t1 = now()
                yield x
                t2 = now()
                tl = dict(slot._tl)
                evs, t_launch = slot._ev
                tl.update(take1=t1, release=t2, launch=t_launch,
                          gpu=[base["ev"].elapsed_time(e) * 1e-3 + base["t"] for e in evs])
                tl.pop("ev", None)
                rec(**tl)

        upload.UploadSlot.stage_and_upload, upload.UploadSlot._upload_and_decode = stage, up
        upload.UploadSlot.finish_decode, upload.ScenePrefetcher._consume = fin, staticmethod(cons)
        fn, fname = (CFR.run_split, "pairs.parquet") if a.entry == "cfr" else (MVI.run_split, "vis.parquet")
        res = []
        import gc
        if a.nogc:
            gc.collect()
            gc.disable()
        if a.gc.startswith("freeze"):
            gc.collect()
            gc.freeze()
            if a.gc.endswith("thr"):
                gc.set_threshold(50000, 20, 100)
        gc_log = []
        gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get("generation"), now())))
        for rep in range(a.passes):
            gc_log.clear()
            log.clear()
            torch.cuda.synchronize()
            base["ev"] = torch.cuda.Event(enable_timing=True)
            base["ev"].record()
            base["ev"].synchronize()
            base["t"] = now()
            tm = sweep.Timings()
            from mspa import hostinfo
            th0, cpu0 = hostinfo.throttle_stats(), os.times()
            d = os.path.join(root, "out", f"p{rep}")
            with contextlib.redirect_stdout(io.StringIO()):
                t0 = now()
                fn(paths["info_path"], os.path.join(d, fname), os.path.join(d, fname + ".warn.txt"), num_workers=a.workers, keep=False,
                   ctx=None, timings=tm)
                dt = now() - t0
            th1, cpu1 = hostinfo.throttle_stats(), os.times()
            host = {"cpu_s_user_sys": round((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system), 3),
                    "cfs": {k: th1[k] - th0.get(k, 0) for k in th1}}
            res.append((dt, list(log), t0, tm.as_dict(), host))
        dt, lg, t0, tm, host = res[-1]
        lg.sort(key=lambda r: r["stage0"])
        print(f"# {a.entry}: {a.scenes} scenes x {a.frames} frames, pass {a.passes}: {dt:.3f} s = {a.scenes / dt:.1f} scenes/s; timings {tm}")
        print("# per scene, ms since the pass began: stage0 (slot obtained) | stage ms | free-wait ms | H2D begin..end | inflate end | un-filter end | consumer saw it decoded | released | slot held ms")
        rows = []
        for r in lg:
            g = [(x - t0) * 1e3 for x in r["gpu"]]
            row = dict(stage0=(r["stage0"] - t0) * 1e3, stage_ms=(r["stage1"] - r["stage0"]) * 1e3, free_wait=r["free_wait"] * 1e3, h2d0=g[0], h2d1=g[1],
                       inf1=g[2], unf1=g[3], fin0=(r.get("fin0", r["take1"]) - t0) * 1e3, seen=(r["take1"] - t0) * 1e3, rel=(r["release"] - t0) * 1e3)
            row["held"] = row["rel"] - row["stage0"]
            row["up0"], row["up1"], row["fr"] = (r["up0"] - r["stage0"]) * 1e3, (r["up1"] - r["stage0"]) * 1e3, r["from_resident_ms"]
            rows.append(row)
        for i, w in enumerate(rows):
            if a.brief:
                break
            if i < 40 or i >= len(rows) - 8:
                print(f"{i:3d} {w['stage0']:8.1f} | {w['stage_ms']:5.1f} | {w['free_wait']:5.1f} | {w['h2d0']:8.1f}..{w['h2d1']:8.1f} | {w['inf1']:8.1f} | {w['unf1']:8.1f} | {w['fin0']:8.1f} -> {w['seen']:8.1f} | {w['rel']:8.1f} | {w['held']:6.1f}")
        mid = rows[16:-8] if len(rows) > 40 else rows
        mean = lambda f: float(np.mean([f(w) for w in mid]))
        summ = {"scenes_per_s": round(a.scenes / dt, 1), "host": host, "cpus_busy": round(host["cpu_s_user_sys"] / dt, 2), "first_slot_obtained_at_ms": round(rows[0]["stage0"], 1),
                "first_release_at_ms": round(rows[0]["rel"], 1), "last_release_at_ms": round(rows[-1]["rel"], 1), "pass_ms": round(dt * 1e3, 1),
                "mid_region_scenes_per_s": round(1e3 * (len(mid) - 1) / (mid[-1]["rel"] - mid[0]["rel"]), 1),
                "slot_held_ms": round(mean(lambda w: w["held"]), 2),
                "stage_ms": round(mean(lambda w: w["stage_ms"]), 2),
                "of_which_wait_for_the_slots_previous_user_ms": round(mean(lambda w: w["free_wait"]), 2),
                "stage_helper_starts_after_ms": round(mean(lambda w: w["up0"]), 2), "stage_helper_upload_and_decode_enqueue_ms": round(mean(lambda w: w["up1"] - w["up0"]), 2),
                "stage_from_resident_ms": round(mean(lambda w: w["fr"]), 2),
                "stage0_to_h2d_begin_ms": round(mean(lambda w: w["h2d0"] - w["stage0"]), 2),
                "h2d_ms": round(mean(lambda w: w["h2d1"] - w["h2d0"]), 2),
                "inflate_ms": round(mean(lambda w: w["inf1"] - w["h2d1"]), 2),
                "unfilter_ms": round(mean(lambda w: w["unf1"] - w["inf1"]), 2),
                "decoded_to_consumer_asks_ms": round(mean(lambda w: w["fin0"] - w["unf1"]), 2),
                "consumer_waits_in_finish_decode_ms": round(mean(lambda w: w["seen"] - w["fin0"]), 2),
                "consumer_holds_ms": round(mean(lambda w: w["rel"] - w["seen"]), 2),
                "interval_between_releases_ms": round(float(np.mean(np.diff([w["rel"] for w in mid]))), 2)}
        if a.series:
            print("inflate_ms per scene:", [round(w["inf1"] - w["h2d1"], 1) for w in rows])
            print("h2d_ms per scene:", [round(w["h2d1"] - w["h2d0"], 1) for w in rows])
            pauses = [(g0[1], round((g1[2] - g0[2]) * 1e3, 1), round((g0[2] - t0) * 1e3)) for g0, g1 in zip(gc_log[::2], gc_log[1::2])]
            print("collector runs in the pass (generation, ms, at ms):", [p_ for p_ in pauses if p_[1] >= 2.0], "of", len(pauses), "total ms", round(sum(p_[1] for p_ in pauses), 1))
            big = [(i, round(b["rel"] - a_["rel"], 1), round(a_["rel"])) for i, (a_, b) in enumerate(zip(rows, rows[1:])) if b["rel"] - a_["rel"] > 20]
            print("release intervals > 20 ms (scene, ms, at ms):", big)
            print("release interval ms:", [round(b["rel"] - a_["rel"], 1) for a_, b in zip(rows, rows[1:])])
        print(json.dumps(summ))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
