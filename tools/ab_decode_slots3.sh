cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/gc
run() {
  echo -n "slots $1 cap $2 $3: "
  MSPA_DECODE_SLOTS=$1 MSPA_DECODE_MAX_FRAMES=$2 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes 192 --workers 8 --decode device --passes 5 --per-rank 8 $3 > gpurun_out/gc/s_$1.json 2> gpurun_out/gc/s.err
  python - <<P
import json, statistics
d = json.load(open("gpurun_out/gc/s_$1.json")); n = d["scenes"]
for k, v in d["worlds"]["1"].items():
    if isinstance(v, dict) and "passes_s" in v:
        p = v["passes_s"][1:]
        print(k.split(".")[0][:12], "median %.1f best %.1f scenes/s" % (n / statistics.median(p), n / min(p)), [round(x, 3) for x in p], end="; ")
print()
P
}
run 8 2560; run 10 3200; run 8 2560; run 10 3200; run 10 3200 --smooth
