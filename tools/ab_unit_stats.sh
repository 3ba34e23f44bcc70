# Per-kernel rocprofv3 stats of 300 calibration iterations of one unit kind (rb | rbu) with the P3 path off and on.
# Run on the GPU box from the repo root: bash tools/ab_unit_stats.sh ; outputs gpurun_out/ab_unit/<kind>_p3_<0|1>_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_unit
mkdir -p $OUT
for u in rb rbu; do
for p in 0 1; do
  export RDO_USE_H2=$p
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python3 $R/tools/long_run_units.py --iters 320 --images 16 --units $u > $OUT/log_${u}_$p.txt 2>&1
  echo "$u p3=$p rc=$?"
  cp $(find $OUT/t -name "*kernel_stats.csv" | head -1) $OUT/${u}_p3_${p}_stats.csv
  rm -rf $OUT/t
done
done
