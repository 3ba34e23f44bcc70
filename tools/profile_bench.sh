# rocprofv3 passes over the default bench workload: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE and the MFMA-utilisation counters
# each in their own --pmc pass (never combined with other trace domains).  Run on the GPU box from the repo root;
# summaries: tools/summarize_rocprof.py, tools/pmc_util_table.py.      usage: bash tools/profile_bench.sh [out-tag] [passes: trace fetch write util]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_r04}
PASSES=${2:-"trace fetch write util"}
mkdir -p $OUT
CMD="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-steps 0 --no-extras --recon-iters 0"
for p in $PASSES; do
  case $p in
    trace) timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/log_trace.txt 2>&1; echo trace rc=$?;;
    fetch) timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/log_fetch.txt 2>&1; echo fetch rc=$?;;
    write) timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > $OUT/log_write.txt 2>&1; echo write rc=$?;;
    util)
      for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
        timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/util_$c -o u --output-format csv -- $CMD > $OUT/log_$c.txt 2>&1; echo $c rc=$?
      done;;
  esac
done
cd $R
[ -d $OUT/trace ] && python3 tools/summarize_rocprof.py $OUT/trace > $OUT/summary_trace.md
[ -d $OUT/fetch ] && python3 tools/summarize_rocprof.py $OUT/fetch --pmc FETCH_SIZE > $OUT/summary_fetch.md
[ -d $OUT/write ] && python3 tools/summarize_rocprof.py $OUT/write --pmc WRITE_SIZE > $OUT/summary_write.md
[ -d $OUT/util_GRBM_GUI_ACTIVE ] && python3 tools/pmc_util_table.py $OUT > $OUT/summary_util.md
tail -1 $OUT/log_trace.txt 2>/dev/null | cut -c1-300
# keep the merged artefacts small: the raw counter CSVs are large
rm -rf $OUT/fetch $OUT/write $OUT/util_*
find $OUT/trace -name "*kernel_trace.csv" -delete 2>/dev/null
