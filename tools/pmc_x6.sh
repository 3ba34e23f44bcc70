# clock / MFMA-busy of the bf16x6 forward kernel variants (one --pmc pass each); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_x6
for ver in 5 6; do
  export RDO_X6_VER=$ver
  tag=v$ver
  timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_x6/$tag -o $tag --output-format csv -- python3 $R/tools/bench_one_x6.py 128 192 192 3 1 1 4 6 > $R/gpurun_out/pmc_x6/log_$tag.txt 2>&1
  echo $tag rc=$?
done
