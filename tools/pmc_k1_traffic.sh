#!/bin/bash
# K1's L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, --kernel-trace only) for every
# tools/ab/libmspa_k1*.so and the in-tree library, on tools/ab_k1.py's scene (320 images x 131 072 vertices), one library
# per process.  FETCH_SIZE is doubled (gfx950 reports half of the streamed bytes: profiles/r01_counter_calibration.md).
# usage (GPU box): bash tools/pmc_k1_traffic.sh > gpurun_out/<...>.md
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
echo "# K1 traffic per 320-image scene (vertex_visibility_compact_kernel), per library"
echo
echo "| library | order | FETCH_SIZE x 2 (MB) | WRITE_SIZE (MB) | total (MB) | x compulsory (205 MB) | dispatches |"
echo "|---|---|---|---|---|---|---|"
for lib in $ROOT/tools/ab/libmspa_k1*.so $ROOT/multi-spatialmllm_amd/libmspa.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/k1t_$c
    MSPA_AB_ONLY=$name timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/k1t_$c -o p -- python $ROOT/tools/ab_k1.py --reps 4 --rounds 1 > /dev/null 2>&1
  done
  python - <<PY
import csv, glob
def rows(c):
    out = []
    for f in glob.glob(f"/tmp/k1t_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "vertex_visibility_compact" in r["Kernel_Name"] and r["Counter_Name"] == c:
                out.append((int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    return [v for _, v in sorted(out)]
f, w = rows("FETCH_SIZE"), rows("WRITE_SIZE")
# ab_k1.py: per order (shuffled, then morton) 3 warm-up + 4 timed launches
n = len(f) // 2
for o, sl in (("shuffled", slice(0, n)), ("morton", slice(n, 2 * n))):
    ff, ww = f[sl], w[sl]
    if not ff:
        continue
    fm, wm = sum(ff) / len(ff) * 2048 / 1e6, (sum(ww) / len(ww) * 1024 / 1e6 if ww else float("nan"))
    print(f"| $name | {o} | {fm:.1f} | {wm:.1f} | {fm + wm:.1f} | {(fm + wm) / 205.0:.2f} | {len(ff)} |")
PY
done
