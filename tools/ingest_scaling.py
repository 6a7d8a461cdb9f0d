#!/usr/bin/env python3
"""Where does the host-side depth-PNG ingest stop scaling?  No GPU work: `mspa.ingest.read_depth_frames` on 320-frame scenes
with T native threads per call, C concurrent calls per process (the loader's look-ahead) and P processes (the ranks of a
job), on hard-linked and on distinct files, with page-fault / context-switch counts per leg and optional CPU / memory binding.

    python tools/ingest_scaling.py [--frames 320] [--scenes 4] [--seconds 2.0]
"""
import argparse
import json
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def node_cpus():
    out = {}
    base = "/sys/devices/system/node"
    if not os.path.isdir(base):
        return out
    for d in sorted(os.listdir(base)):
        if d.startswith("node") and d[4:].isdigit():
            cpus = []
            for part in open(os.path.join(base, d, "cpulist")).read().strip().split(","):
                if "-" in part:
                    a, b = part.split("-")
                    cpus += list(range(int(a), int(b) + 1))
                elif part:
                    cpus.append(int(part))
            out[int(d[4:])] = cpus
    return out


def child(a):
    """One process: C caller threads, each decoding scene lists round robin with T native threads for `seconds`."""
    import numpy as np
    from mspa import ingest
    if a.bind_node >= 0:
        cpus = node_cpus().get(a.bind_node)
        if cpus:
            os.sched_setaffinity(0, cpus)
    lists = json.load(open(a.lists))
    h, w = 480, 640
    bufs = [np.zeros((len(lists[0]), h, w), dtype=np.uint16) for _ in range(a.callers)] if a.reuse else None
    for c in range(a.callers):                       # warm: threads, scratch, page cache
        ingest.read_depth_frames(lists[c % len(lists)], a.threads, out=None if bufs is None else bufs[c])
    # rendezvous with the other processes: a file per process, wait until all are there
    open(os.path.join(a.sync, f"ready{a.index}"), "w").close()
    while len([f for f in os.listdir(a.sync) if f.startswith("ready")]) < a.procs:
        time.sleep(0.001)
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    counts, per_scene = [0] * a.callers, [[] for _ in range(a.callers)]

    def run(c):
        k = c + a.index * a.callers
        while time.perf_counter() - t0 < a.seconds:
            t = time.perf_counter()
            ingest.read_depth_frames(lists[k % len(lists)], a.threads, out=None if bufs is None else bufs[c])
            per_scene[c].append(time.perf_counter() - t)
            counts[c] += len(lists[k % len(lists)])
            k += 1
    th = [threading.Thread(target=run, args=(c,)) for c in range(a.callers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    ms = sorted(x * 1e3 for c in per_scene for x in c)
    print(json.dumps({"frames": sum(counts), "seconds": dt, "scene_ms_median": ms[len(ms) // 2], "scene_ms_max": ms[-1],
                      "minflt": r1.ru_minflt - r0.ru_minflt, "nvcsw": r1.ru_nvcsw - r0.ru_nvcsw, "nivcsw": r1.ru_nivcsw - r0.ru_nivcsw,
                      "utime": r1.ru_utime - r0.ru_utime, "stime": r1.ru_stime - r0.ru_stime}))


def leg(lists_path, procs, callers, threads, seconds, reuse=True, bind_node=-1, env=None):
    sync = tempfile.mkdtemp(prefix="mspa_sync_")
    try:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "--lists", lists_path, "--procs", str(procs),
                                "--index", str(i), "--callers", str(callers), "--threads", str(threads), "--seconds", str(seconds),
                                "--sync", sync, "--reuse", str(int(reuse)), "--bind-node", str(bind_node)],
                               stdout=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {}))) for i in range(procs)]
        outs = [json.loads(p.communicate(timeout=300)[0].strip().splitlines()[-1]) for p in ps]
    finally:
        shutil.rmtree(sync, ignore_errors=True)
    fps = sum(o["frames"] / o["seconds"] for o in outs)
    return {"procs": procs, "callers": callers, "threads": threads, "reuse": reuse, "bind_node": bind_node,
            "frames_per_s": round(fps), "scenes320_per_s": round(fps / 320, 1), "GBps_out": round(fps * 614400 / 1e9, 2),
            "scene_ms_median": round(sorted(o["scene_ms_median"] for o in outs)[len(outs) // 2], 1),
            "minflt": sum(o["minflt"] for o in outs), "nivcsw": sum(o["nivcsw"] for o in outs), "nvcsw": sum(o["nvcsw"] for o in outs),
            "cpu_s_user": round(sum(o["utime"] for o in outs), 2), "cpu_s_sys": round(sum(o["stime"] for o in outs), 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--lists")
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--index", type=int, default=0)
    ap.add_argument("--callers", type=int, default=1)
    ap.add_argument("--threads", type=int, default=25)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--sync")
    ap.add_argument("--reuse", type=int, default=1)
    ap.add_argument("--bind-node", type=int, default=-1)
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--scenes", type=int, default=4)
    a = ap.parse_args()
    if a.child:
        child(a)
        return
    from mspa import synth
    root = tempfile.mkdtemp(prefix="mspa_ingest_scaling_")
    try:
        H, W = 480, 640
        base = synth.make_scene(5000, n_points=4096, n_frames=8, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0, with_color=False)
        ids = base.image_ids

        def layout(tag, link):
            scenes = []
            for s in range(a.scenes):
                depth = {f"{5 * f:05d}": (base.depth[ids[f % 8]] if link else base.depth[ids[f % 8]].copy()) for f in range(a.frames)}
                E = {k: base.E[ids[0]] for k in depth}
                scenes.append(synth.SynthScene(f"scene{s:04d}_00", base.K, base.A, E, base.points, depth, {}, base.color_hw, base.depth_hw, base.boxes))
            p = synth.write_scannet_layout(scenes, os.path.join(root, tag), compress_level=6, link_identical=link)
            lists = [[os.path.join(p["posed_images_root"], sc.scene_id, f"{k}.png") for k in sc.depth] for sc in scenes]
            path = os.path.join(root, f"{tag}.json")
            json.dump(lists, open(path, "w"))
            return path
        t0 = time.perf_counter()
        linked = layout("linked", True)
        t1 = time.perf_counter()
        distinct = layout("distinct", False)
        t2 = time.perf_counter()
        nodes = node_cpus()
        print(f"# {a.scenes} scenes x {a.frames} frames; linked inputs {t1 - t0:.1f} s, distinct inputs {t2 - t1:.1f} s; {os.cpu_count()} cpus; "
              f"NUMA nodes {({k: len(v) for k, v in nodes.items()})}; THP {open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip() if os.path.exists('/sys/kernel/mm/transparent_hugepage/enabled') else '?'}")
        try:
            print("# cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
        except Exception as e:
            print("# cgroup cpu.max: ?", e)
        print("# affinity of this process:", len(os.sched_getaffinity(0)), "cpus")
        S = a.seconds
        for name, lp in (("linked", linked), ("distinct", distinct)):
            print(f"## {name} files")
            for (p, c, t) in ((1, 1, 1), (1, 1, 8), (1, 1, 25), (1, 1, 64), (1, 1, 128), (1, 2, 25), (1, 4, 25), (2, 2, 25), (4, 2, 25), (4, 4, 25),
                              (4, 1, 25), (8, 1, 25), (4, 1, 8), (16, 1, 8)):
                print(json.dumps(leg(lp, p, c, t, S)), flush=True)
        print("## distinct files, fresh destination per call")
        for (p, c, t) in ((1, 1, 25), (1, 4, 25), (4, 4, 25)):
            print(json.dumps(leg(distinct, p, c, t, S, reuse=False)), flush=True)
        if len(nodes) > 1:
            print("## distinct files, bound to one NUMA node's cpus")
            for n in sorted(nodes):
                print(json.dumps(leg(distinct, 1, 4, 25, S, bind_node=n)), flush=True)
        print("## distinct files, glibc malloc tuned (MALLOC_ARENA_MAX=1 / mmap threshold off)")
        print(json.dumps(leg(distinct, 1, 4, 25, S, env={"MALLOC_ARENA_MAX": "1"})), flush=True)
        print(json.dumps(leg(distinct, 1, 4, 25, S, env={"MALLOC_MMAP_THRESHOLD_": "1073741824", "MALLOC_TRIM_THRESHOLD_": "1073741824"})), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
