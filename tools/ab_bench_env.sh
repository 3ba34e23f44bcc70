# In-loop A/B of a tuning key through its environment variable: bench.py --no-extras once per value, prints ms_per_step and the kernel's share.
# usage: bash tools/ab_bench_env.sh RDO_WGRAD_SUB conv_wgrad_h2_rows 1 9 13
cd $GRAFT_REPO_ROOT
VAR=$1; KER=$2; shift 2
for v in "$@"; do
  env $VAR=$v timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels'].get('$KER',{})
print('$VAR=$v', 'ms_per_step', d['ms_per_step'], 'sustained', d.get('sustained_ms_per_step'), '$KER', k.get('ms_per_step'), 'per launch us', round(1e3*k.get('ms_per_step',0)/max(k.get('launches_per_step',1),1),1))
"
done
