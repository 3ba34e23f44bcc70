# In-engine ablation of the plane-input forward kernel: 220 iterations of the RB unit at 128^2 with x6p_ablate = 0 | 3 (no DMA) |
# 4 (no MFMA) | 11 (no DMA, no fragment reads); per-kernel stats to gpurun_out/ab_unit/rb_abl<k>_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab_unit
mkdir -p $OUT
export RDO_USE_H2=1
for k in 0 3 4 11 32; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python3 $R/tools/long_run_units.py --iters 220 --images 16 --units rb --tune x6p_ablate=$k > $OUT/log_abl_$k.txt 2>&1
  echo "abl=$k rc=$?"
  cp $(find $OUT/t -name "*kernel_stats.csv" | head -1) $OUT/rb_abl${k}_stats.csv
  rm -rf $OUT/t
  grep -h "conv_fwd_x6p_kernel\|conv_wgrad_x6p" $OUT/rb_abl${k}_stats.csv | cut -d, -f2-6
done
