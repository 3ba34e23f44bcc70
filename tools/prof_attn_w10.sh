# BASELINE config 3 (Cheng2020-attn N=192 W10A10, 105 units) under rocprofv3: whole-process kernel stats of a 200-iteration schedule
# (loops dominate: ~2.5 s of ~4) + the per-unit wall split.   bash tools/prof_attn_w10.sh  (on the GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_attn
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_attn -o attn --output-format csv -- python3 $R/tools/full_schedule.py --arch attn --w-bits 10 --a-bits 10 --images 16 --iters 200 --no-quality --json $R/gpurun_out/prof_attn/schedule.json > $R/gpurun_out/prof_attn/log.txt 2>&1
echo rc=$?
cp $(find $R/gpurun_out/prof_attn -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06f_attn_w10_kernel_stats.csv
