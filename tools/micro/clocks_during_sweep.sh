cd $GRAFT_REPO_ROOT
for m in 0 1; do
  echo "== prepare=$m"
  (MSPA_PREPARE_ON_LOADER=$m timeout 300 python tools/sweep_timeline.py --scenes 192 --passes 4 --brief 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('scenes_per_s', 'mid_region_scenes_per_s', 'slot_held_ms', 'h2d_ms', 'inflate_ms')})") &
  PID=$!
  sleep 9
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | awk '{print $NF}' | tr '\n' ' '; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1 | awk '{print $NF}'; sleep 0.7; done
  wait $PID
done
