import json, os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "multi-spatialmllm_amd")]
import bench, torch
dev = torch.device("cuda", 0)
def run(tag):
    d = bench.time_dropin_sweep()
    print(tag, d["scenes_per_s"], json.dumps(d["stage_busy_s"])[:160], flush=True)
run("fresh process        ")
bench.time_track_geometry(dev)
run("after K5 leg         ")
bench.time_scene_kernels(dev)
run("after scene kernels  ")
bench.time_scene_pipeline(dev)
run("after pipeline leg   ")
bench.time_scannet_shape(dev)
run("after scannet leg    ")
bench.measured_hbm_ceilings(dev)
run("after hbm ceilings   ")
