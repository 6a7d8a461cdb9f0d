cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/enc
run() {
  echo -n "encoders $1: "
  MSPA_ENCODE_THREADS=$1 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes 96 --workers 8 --decode device --passes 5 --per-rank 8 > gpurun_out/enc/e_$1.json 2> gpurun_out/enc/e.err
  python - <<P
import json, statistics
d = json.load(open("gpurun_out/enc/e_$1.json"))["worlds"]["1"]
for k, v in d.items():
    if isinstance(v, dict) and "passes_s" in v:
        p = v["passes_s"][1:]
        print(k.split(".")[0][:12], "median %.1f best %.1f scenes/s" % (96 / statistics.median(p), 96 / min(p)), [round(x, 3) for x in p], "encode", v.get("encode_s"), "waited", v.get("encode_wait_s"), end="; ")
print(d.get("cfs_throttling_whole_run"))
P
}
run 8; run 10; run 12; run 16; run 8
