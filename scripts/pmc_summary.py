#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: launches, mean counter value."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
out = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            m = re.search(r"(\w+_kernel)", name)
            short = m.group(1) if m else name[:40]
            tw = re.search(r"kernelILi(\d+)", name)
            if tw:
                short += f"<{tw.group(1)}>"
            if "Lb1" in name:
                short += "<res>"
            # demangled names (rocprofv3 prints the r16 kernels that way): "void r16::stft_r16_kernel<24, __bf16, 1, true>(...)" -- the
            # fused STFT + first-layer kernel (last argument true) must not be averaged with the plain STFT under one name
            dm = re.search(r"(i?stft_r\d+_kernel)<([^>]*)>", name)
            if dm:
                targs = [a.strip() for a in dm.group(2).split(",")]
                short = f"{dm.group(1)}<{targs[0]}>" + ("<fused>" if targs[-1] == "true" and dm.group(1).startswith("stft") else "")
            key = f'{short} grid={int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)}'
            a = agg[key][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for k, cs in agg.items():
        out.setdefault(k, {})
        for c, (n, s) in cs.items():
            out[k][c] = {"launches": n, "mean": s / n}
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k in sorted(out):
    print(k, {c: round(v["mean"], 1) for c, v in out[k].items()})
