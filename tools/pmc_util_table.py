#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and clock from three separate rocprofv3 --pmc passes over bench.py (tools/profile_bench.sh, pass `util`):

    <out>/util_SQ_VALU_MFMA_BUSY_CYCLES, <out>/util_SQ_BUSY_CYCLES, <out>/util_GRBM_GUI_ACTIVE   (each with --kernel-trace)

Only hot-loop dispatches are counted (same rule as tools/summarize_rocprof.py).  Columns, per kernel, means over its dispatches:
  * duration (us) of the dispatch in that pass,
  * clock = GRBM_GUI_ACTIVE / 8 XCDs / duration -- printed ONLY for kernels whose dispatches average >= 100 us: the counter's window is
    wider than the dispatch (it includes the command processor's work around it), so the quotient read 3-6 GHz on a 2.4 GHz part for
    every short kernel in rounds 4-5.  Even at 100 us it over-reads by a few per cent; the in-kernel
    clock64 / wall_clock64 stamps of tools/h2k_stamps.py and tools/wgrad_ablate.py are the evidence for the power-limit reading
    of the dominant kernels, this column is a cross-check,
  * MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8): the share of the kernel's cycles in which a SIMD's
    matrix pipe was busy, averaged over the chip (the counter sums cycles over all SIMDs; it counts cycles, not quad-cycles),
  * SQ busy = SQ_BUSY_CYCLES / 32 shader engines / (GRBM_GUI_ACTIVE / 8) (a sanity column: ~1 for kernels that keep every SE busy)."""
import csv
import glob
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import hot_start, short  # noqa: E402

out = sys.argv[1]
per = {}
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    fs = glob.glob(os.path.join(out, "util_" + c, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    start = hot_start(rows)
    agg = OrderedDict()
    for r in rows[start:]:
        if r["Counter_Name"] != c:
            continue
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per[c] = agg
g = per.get("GRBM_GUI_ACTIVE", {})
print("| kernel | dispatches | avg us | clock GHz | MFMA busy | SQ busy |")
print("|---|---|---|---|---|---|")
for k, (n, cyc, ns) in sorted(g.items(), key=lambda kv: -kv[1][2]):
    xcd_cyc = cyc / n / 8.0                       # cycles of the dispatch (per XCD)
    us = ns / n / 1e3
    clk = xcd_cyc / (ns / n)
    m = per.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get(k)
    b = per.get("SQ_BUSY_CYCLES", {}).get(k)
    mf = (m[1] / m[0]) / 1024.0 / xcd_cyc if m else float("nan")
    sq = (b[1] / b[0]) / 32.0 / xcd_cyc if b else float("nan")      # summed over the 32 shader engines
    print(f"| {k} | {n} | {us:.1f} | {f'{clk:.2f}' if us >= 100.0 else '-'} | {mf * 100:.1f} % | {sq:.2f} |")
print("\n(clock: only for dispatches of >= 100 us, see the docstring; MFMA busy and SQ busy are relative to the counter's own window and "
      "therefore UNDER-read on short dispatches)")
