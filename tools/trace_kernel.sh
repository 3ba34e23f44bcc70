# Per-dispatch durations of the kernels whose name matches $1 in a 3-step bench run (kernel trace only).
# Run on the GPU box from the repo root: bash tools/trace_kernel.sh conv_wgrad_kernel ; output gpurun_out/trace_<name>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_tmp
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustain-steps 0 > $OUT/log.txt 2>&1
echo rc=$?
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$F" "$1" > $R/gpurun_out/trace_$1.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
start = next(i for i, r in enumerate(rows) if "gather_qdrop" in r["Kernel_Name"])
sel = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r["Kernel_Name"][:70]) for r in rows[start:] if pat in r["Kernel_Name"]]
n = len(sel) // 5 if len(sel) >= 5 else len(sel)
for d in sel[-n:]:
    print(f"{d[0] / 1e3:9.1f} us  grid {d[1]}x{d[2]}x{d[3]}  {d[4]}")
PY
rm -rf $OUT
