#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of tools/profile_bench.sh: HBM-side MiB per launch of the GEMM
kernels, keyed by the bench's kernel tags (bench.py reads it for `roofline.traffic`).
usage: python tools/make_traffic_json.py <summary_fetch.md> <summary_write.md> <out.json>"""
import json
import sys

TAGS = {"conv_fwd_h2h_kernel": "conv_fwd_h2_halo", "conv_fwd_h2h_kernel<4, 2, 2, 3>": "conv_fwd_h2_halo",
        "conv_fwd_h2h_kernel<8, 1, 1, 2>": "conv_fwd_h2_halo64", "conv_wgrad_h2r_kernel": "conv_wgrad_h2_rows", "conv_fwd_h2_kernel": "conv_fwd_h2_256x192",
        "conv_wgrad_h2_kernel": "conv_wgrad_h2_192x192", "conv_fwd_x6v5_kernel": "conv_fwd_x6_128x192", "conv_wgrad_x6w8_kernel<false>": "conv_wgrad_x6_192x192",
        "ada_step_batch_kernel": "ada_step", "conv_fwd_h2k_kernel<4, 2, 4, 6, 0>": "conv_fwd_h2_halo", "conv_fwd_h2k_kernel<8, 1, 2, 3, 0>": "conv_fwd_h2_halo64",
        "conv_fwd_h2k_kernel<8, 1, 2, 4, 0>": "conv_fwd_h2_halo64", "conv_wgrad_h2r_kernel<1, 2, 2>": "conv_wgrad_h2_rows"}


def table(path):
    out = {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) >= 4 and c[1].isdigit():
            out[c[0]] = (int(c[1]), float(c[3]))
    return out


f, w = table(sys.argv[1]), table(sys.argv[2])
res = {}
PREFIX = {"conv_fwd_h2k_kernel<4, 2, 4, 6": "conv_fwd_h2_halo", "conv_fwd_h2k_kernel<8, 1, 2,": "conv_fwd_h2_halo64",
          "conv_wgrad_h2r_kernel<": "conv_wgrad_h2_rows"}           # template kernels: any trailing diagnostic parameters
for k in f:
    tag = TAGS.get(k) or next((t for p, t in PREFIX.items() if k.startswith(p)), None)
    if tag and k in w and tag not in res:
        res[tag] = {"fetch_mib_raw": f[k][1], "write_mib": w[k][1], "kernel": k, "dispatches": f[k][0]}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
