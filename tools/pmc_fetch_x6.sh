# HBM-side FETCH_SIZE / WRITE_SIZE of the bf16x6 forward kernel on one shape (own --pmc passes); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_fetch
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_fetch/$c -o p --output-format csv -- python3 $R/tools/bench_one_x6.py ${SHAPE:-128 192 192 3 1 1} 4 6 > $R/gpurun_out/pmc_fetch/log_$c.txt 2>&1
  echo $c rc=$?
done
cd $R
python3 tools/pmc_table.py gpurun_out/pmc_fetch/FETCH_SIZE x6v
python3 tools/pmc_table.py gpurun_out/pmc_fetch/WRITE_SIZE x6v
