"""Timeline of mspa.upload.ScenePrefetcher on the bench's pipeline workload: when each scene is staged / received / done."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-spatialmllm_amd"))
import numpy as np, torch
from mspa import synth, upload
H, W = 480, 640
sc = synth.make_scene(4000, n_points=131072, n_frames=8, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0, with_color=False)
ids0 = sc.valid_image_ids
class HostScene:
    K, A, color_hw, points = sc.K, sc.A, sc.color_hw, sc.points
    E = {f"{r:03d}_{i}": sc.E[i] for r in range(40) for i in ids0}
    depth = {f"{r:03d}_{i}": sc.depth[i] for r in range(40) for i in ids0}
log = []
if os.environ.get("SWITCH"): sys.setswitchinterval(float(os.environ["SWITCH"]))
orig = upload.UploadSlot.stage_and_upload
def traced(self, s, cs):
    t0 = time.perf_counter(); r = orig(self, s, cs); t1 = time.perf_counter()
    ev = torch.cuda.Event(enable_timing=False)
    log.append(("stage", t0, t1)); return r
upload.UploadSlot.stage_and_upload = traced
def run(n):
    t_prev = time.perf_counter()
    for scene in upload.ScenePrefetcher([HostScene] * n, "cuda"):
        t_got = time.perf_counter()
        torch.cuda.current_stream().synchronize()          # upload complete
        t_up = time.perf_counter()
        scene.frames_relations_arrays()
        t_done = time.perf_counter()
        log.append(("consume", t_prev, t_got, t_up, t_done)); t_prev = t_done
run(5); log.clear()
T0 = time.perf_counter(); run(12); T1 = time.perf_counter()
print("12 scenes in %.1f ms = %.1f scenes/s" % ((T1 - T0) * 1e3, 12 / (T1 - T0)))
for e in log:
    if e[0] == "stage": print("stage    %7.2f -> %7.2f  (%.2f ms)" % ((e[1] - T0) * 1e3, (e[2] - T0) * 1e3, (e[2] - e[1]) * 1e3))
    else: print("consume  wait %5.2f  upload-wait %5.2f  relations %5.2f   at %7.2f" % ((e[2] - e[1]) * 1e3, (e[3] - e[2]) * 1e3, (e[4] - e[3]) * 1e3, (e[4] - T0) * 1e3))
