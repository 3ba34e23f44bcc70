cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_lu
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lu -o lu --output-format csv -- python3 $R/tools/bench_lu2022.py ${UNIT:-g_a1} > $R/gpurun_out/prof_lu/log.txt 2>&1
echo rc=$?
