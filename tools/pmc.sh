#!/bin/bash
# Memory-path PMC passes for one kernel (run on the GPU box through gpurun).  Separate rocprofv3 runs, --kernel-trace only
# (never combined with sys/hip/hsa traces).  usage: tools/pmc.sh <tag> <kernel-name substring> <command...>
TAG=$1; PAT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1; }
CMD="$*"
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
run ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcp2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcc2 TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum
python - <<PY | tee $OUT/summary.md
import csv, glob, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    sub = f.split("/")[-2]
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if sub == "sq" and r["Counter_Name"] == "SQ_WAVES":
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# PMC of kernels matching \`$PAT\` ($TAG): mean per dispatch\n")
print(f"dispatches: {len(dur)}, duration under PMC (us): mean {sum(dur)/max(1,len(dur)):.1f}, max {max(dur) if dur else 0:.1f}\n")
print("| counter | mean | max |\n|---|---|---|")
for k in sorted(acc):
    v = acc[k]
    print(f"| {k} | {sum(v)/len(v):.4g} | {max(v):.4g} |")
PY
for d in sq sq2 ta ta2 tcp tcp2 tcc tcc2 fetch write grbm; do rm -rf $OUT/$d; done     # raw CSVs: too big to bring back
