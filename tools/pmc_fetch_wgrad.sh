# HBM-side FETCH_SIZE / WRITE_SIZE of the wgrad kernel on one shape (own --pmc passes); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_fetch_wg
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_fetch_wg/$c -o p --output-format csv -- python3 $R/tools/bench_one.py wgrad ${SHAPE:-128 192 192 3 1 1} 4 6 > $R/gpurun_out/pmc_fetch_wg/log_$c.txt 2>&1
  echo $c rc=$?
done
cd $R
python3 tools/pmc_table.py gpurun_out/pmc_fetch_wg/FETCH_SIZE wgrad
python3 tools/pmc_table.py gpurun_out/pmc_fetch_wg/WRITE_SIZE wgrad
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/pmc_fetch_wg/FETCH_SIZE/p_kernel_trace.csv')) if 'wgrad' in r['Kernel_Name']]
print("avg us", sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)/len(rows)/1e3)
PY
