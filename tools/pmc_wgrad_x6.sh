# clock / MFMA-busy of the bf16x6 wgrad kernel (one --pmc pass); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_wg
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_wg/a -o a --output-format csv -- python3 $R/tools/bench_one.py wgrad 128 192 192 3 1 1 4 6 > $R/gpurun_out/pmc_wg/log_a.txt 2>&1
echo rc=$?
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_wg/b -o b --output-format csv -- python3 $R/tools/bench_one.py wgrad 128 192 192 3 1 1 4 6 > $R/gpurun_out/pmc_wg/log_b.txt 2>&1
echo rc=$?
