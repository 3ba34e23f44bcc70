# Sample board power and clocks (rocm-smi) while bench.py's sustained loop runs.  usage: bash tools/power_watch.sh [ENV=VALUE ...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
env "$@" timeout 600 python bench.py --no-extras --no-cpu-baseline --sustain-steps 4000 > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 45
for i in $(seq 1 40); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|junction\|edge" | tr -s ' ' | tr '\n' '|'
  echo
  sleep 1
done
wait $BP
tail -1 gpurun_out/power_bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'sustained', d.get('sustained_ms_per_step'), d.get('sustained_window_ms'))"
