# Per-kernel rocprofv3 stats of the default bench workload with the P3 path off and on (same box, back to back).
# Run on the GPU box from the repo root: bash tools/ab_kernel_stats.sh ; outputs gpurun_out/ab_p3/{p3_0,p3_1}_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_p3
mkdir -p $OUT
for p in 0 1; do
  export RDO_USE_H2=$p
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t$p -o t --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --sustain-steps 0 > $OUT/log$p.txt 2>&1
  echo "p3=$p rc=$?"
  cp $(find $OUT/t$p -name "*kernel_stats.csv" | head -1) $OUT/p3_${p}_stats.csv
  rm -rf $OUT/t$p
done
