# rocprofv3 --pmc passes (one counter group per pass, kernel-trace only) over one command; prints per-kernel means.
# SQ counters only: passes with TA_* / TCP_* counters did not finish on this pool (each ran into its 300 s limit).
# usage: bash tools/pmc_groups.sh <tag> <kernel-substring> -- python3 tools/bench_one_h2.py wgrad 128 192 192
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; PAT=$2; shift 3
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p --output-format csv -- "$@" > $OUT/log_g$i.txt 2>&1 || echo "group $i rc=$?"
done <<'G'
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM
G
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "g*", "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        v = v[1:] if len(v) > 1 else v
        print(f"   {c:36s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
