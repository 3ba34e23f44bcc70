# clock and MFMA-busy of tools/mfma_probe.hip under rocprofv3 --pmc (one counter set per pass; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_probe
mkdir -p $OUT
hipcc -O3 -std=c++20 --offload-arch=gfx950 $R/tools/mfma_probe.hip -o /tmp/mfma_probe 2>/dev/null
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- /tmp/mfma_probe 432 > $OUT/log_$c.txt 2>&1
  echo $c rc=$?
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_probe"
res = collections.defaultdict(dict)
for c in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = collections.defaultdict(list)
        for r in rows:
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        for k, v in per.items():
            v.sort()
            v = v[1:] if len(v) > 1 else v     # skip the warm-up launch
            res[k][c] = (sum(x[1] for x in v) / len(v), sum(x[3] - x[2] for x in v) / len(v))
for k, d in res.items():
    if "GRBM_GUI_ACTIVE" not in d:
        continue
    cyc, ns = d["GRBM_GUI_ACTIVE"]
    clk = cyc / 8 / ns
    mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 1))[0] / 1024 / (cyc / 8) if "SQ_VALU_MFMA_BUSY_CYCLES" in d else float("nan")
    print(f"{k[:60]:60s} {ns/1e3:8.1f} us  clock {clk:5.2f} GHz  MFMA busy {mf*100:5.1f} %")
PY
