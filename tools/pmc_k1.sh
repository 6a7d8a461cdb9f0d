#!/bin/bash
# One PMC pass over the K1 kernels (VALU and LDS counters) for every tools/ab/libmspa_*.so and the in-tree library.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
for lib in $ROOT/tools/ab/libmspa_*.so $ROOT/multi-spatialmllm_amd/libmspa.so; do
  [ -f "$lib" ] || continue
  rm -rf /tmp/k1pmc
  MSPA_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/k1pmc -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --no-sweep --also none > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/k1pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vertex_visibility" in r["Kernel_Name"] and float(r["Grid_Size"]) > 1e6:
            acc[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else float("nan")
    w = m("SQ_WAVES")
    print("%-24s %s waves %.0f VALU/wave %.0f VALUbusy/wave %.3f LDS/wave %.0f LDSactive/wavecyc %.3f LDSwait/wavecyc %.3f bankconf/LDSactive %.2f" % (
        "$lib".split("/")[-1], k[-36:], w, m("SQ_INSTS_VALU") / w, m("SQ_ACTIVE_INST_VALU") / m("SQ_WAVE_CYCLES"), m("SQ_INSTS_LDS") / w,
        m("SQ_ACTIVE_INST_LDS") / m("SQ_WAVE_CYCLES"), m("SQ_WAIT_INST_LDS") / m("SQ_WAVE_CYCLES"), m("SQ_LDS_BANK_CONFLICT") / max(1.0, m("SQ_ACTIVE_INST_LDS"))))
PY
done
