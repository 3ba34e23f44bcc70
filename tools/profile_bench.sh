# rocprofv3 passes over the default bench workload: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes
# (never combined with other trace domains).  Run on the GPU box from the repo root; summaries: tools/summarize_rocprof.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_r03}
mkdir -p $OUT
CMD="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-steps 0 --no-extras --recon-iters 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/log_trace.txt 2>&1; echo trace rc=$?
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/log_fetch.txt 2>&1; echo fetch rc=$?
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > $OUT/log_write.txt 2>&1; echo write rc=$?
cd $R
python3 tools/summarize_rocprof.py $OUT/trace > $OUT/summary_trace.md
python3 tools/summarize_rocprof.py $OUT/fetch --pmc FETCH_SIZE > $OUT/summary_fetch.md
python3 tools/summarize_rocprof.py $OUT/write --pmc WRITE_SIZE > $OUT/summary_write.md
tail -1 $OUT/log_trace.txt | cut -c1-300
# keep the merged artefacts small: the raw counter CSVs are large
rm -rf $OUT/fetch $OUT/write
find $OUT/trace -name "*kernel_trace.csv" -delete
