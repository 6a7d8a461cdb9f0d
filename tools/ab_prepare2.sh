cd $GRAFT_REPO_ROOT
run() {
  echo -n "prepare=$1 scenes $2: "
  MSPA_PREPARE_ON_LOADER=$1 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes $2 --workers 8 --decode device --passes 6 --per-rank 8 > gpurun_out/gc/p_$1.json 2> gpurun_out/gc/p.err
  python - <<P
import json, statistics
d = json.load(open("gpurun_out/gc/p_$1.json")); n = d["scenes"]
for k, v in d["worlds"]["1"].items():
    if isinstance(v, dict) and "passes_s" in v:
        p = v["passes_s"][1:]
        print(k.split(".")[0][:12], "median %.1f best %.1f scenes/s" % (n / statistics.median(p), n / min(p)), [round(x, 3) for x in p], end="; ")
print()
P
}
mkdir -p gpurun_out/gc
run 0 192; run 1 192; run 0 192; run 1 192; run 0 96; run 1 96
