# rocprofv3 passes over one Lu2022 unit (tools/bench_lu2022.py): kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the MFMA-busy counters each in
# their own --pmc pass (never combined with other trace domains).  usage (GPU box, repo root): UNIT=g_a1 bash tools/profile_lu2022.sh [out-tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_lu_pmc}
mkdir -p $OUT
CMD="python3 $R/tools/bench_lu2022.py ${UNIT:-g_a1}"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/log_trace.txt 2>&1; echo trace rc=$?
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/log_fetch.txt 2>&1; echo fetch rc=$?
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > $OUT/log_write.txt 2>&1; echo write rc=$?
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/util_$c -o u --output-format csv -- $CMD > $OUT/log_$c.txt 2>&1; echo $c rc=$?
done
cd $R
python3 tools/summarize_rocprof.py $OUT/trace > $OUT/summary_trace.md
python3 tools/summarize_rocprof.py $OUT/fetch --pmc FETCH_SIZE > $OUT/summary_fetch.md
python3 tools/summarize_rocprof.py $OUT/write --pmc WRITE_SIZE > $OUT/summary_write.md
python3 tools/pmc_util_table.py $OUT > $OUT/summary_util.md
rm -rf $OUT/fetch $OUT/write $OUT/util_*
find $OUT/trace -name "*kernel_trace.csv" -delete 2>/dev/null
