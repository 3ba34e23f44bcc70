#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of `bench.py` for the calibration hot loop.

    python tools/summarize_rocprof.py <dir with *_kernel_trace.csv [and *_counter_collection.csv]> [--pmc NAME]

Only the hot loop is counted: dispatches from the first `gather_qdrop*` launch that FOLLOWS the last weight-plane split kernel
(`split_*_conv_kernel`: the engines fill their weight planes once, after recording) -- everything before it is model set-up, cache
building and the engines' probe iterations, which are outside the timed region of bench.py.  With --pmc the per-dispatch counter values of a
`--pmc NAME` run are aggregated per kernel instead (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports
wide coalesced reads by 2x -- MI355X_MICROARCH.md, HBM section -- so the table prints both raw and corrected bytes)."""
import csv
import glob
import os
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def hot_start(rows):
    last_split = max((i for i, r in enumerate(rows) if "_conv_kernel" in r["Kernel_Name"] and "split_" in r["Kernel_Name"]), default=-1)
    return next(i for i, r in enumerate(rows) if i > last_split and "gather_qdrop" in r["Kernel_Name"])


def main():
    d = sys.argv[1]
    pmc = sys.argv[sys.argv.index("--pmc") + 1] if "--pmc" in sys.argv else None
    if pmc:
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        start = hot_start(rows)
        agg = OrderedDict()
        for r in rows[start:]:
            if r["Counter_Name"] != pmc:
                continue
            a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print(f"| kernel | dispatches | {pmc} total (KiB) | per dispatch (MiB) |" + (" corrected x2 (MiB) |" if pmc == "FETCH_SIZE" else ""))
        print("|---|---|---|---|" + ("---|" if pmc == "FETCH_SIZE" else ""))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            line = f"| {k} | {n} | {v:.0f} | {v / n / 1024:.2f} |"
            if pmc == "FETCH_SIZE":
                line += f" {2 * v / n / 1024:.2f} |"
            print(line)
        return
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    start = hot_start(rows)
    agg = {}
    for r in rows[start:]:
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    span = int(rows[-1]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])
    print(f"hot-loop dispatches: {len(rows) - start}, kernel time {tot / 1e6:.2f} ms, wall span {span / 1e6:.2f} ms\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {n} | {t / 1e6:.3f} | {t / n / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / tot:.1f} |")


if __name__ == "__main__":
    main()
