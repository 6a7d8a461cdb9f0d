cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/slots
run() {
  echo -n "slots $1 cap $2: "
  MSPA_DECODE_SLOTS=$1 MSPA_DECODE_MAX_FRAMES=$2 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes 96 --workers 8 --decode device --passes 7 --per-rank 8 > gpurun_out/slots/t_$1_$2.json 2> gpurun_out/slots/t.err
  python - <<P
import json, statistics
d = json.load(open("gpurun_out/slots/t_$1_$2.json"))["worlds"]["1"]
for k, v in d.items():
    if isinstance(v, dict) and "passes_s" in v:
        p = v["passes_s"][1:]
        print(k.split(".")[0][:12], "median %.1f best %.1f scenes/s" % (96 / statistics.median(p), 96 / min(p)), [round(x, 3) for x in p], end="; ")
print()
P
}
run 8 2560; run 10 3200; run 11 3520; run 8 2560; run 10 3200; run 11 3520
