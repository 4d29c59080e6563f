#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): count, average and total time."""
import collections
import csv
import glob
import os
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else max(glob.glob("gpurun_out/prof/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel)", name)
    short = (m.group(1) if m else name[:40])
    tw = re.search(r"kernelILi(\d+)", name)
    if tw:
        short += f"<{tw.group(1)}>"
    if "Lb1" in name or "<true" in name:
        short += "<res>"
    key = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"{path}: {len(rows)} dispatches, {tot / 1e3:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{k[0]:34s} blocks=({k[1]:7d},{k[2]:3d}) n={v[0]:5d} avg_us={v[1] / v[0]:9.1f} tot_ms={v[1] / 1e3:8.2f} {100 * v[1] / tot:5.1f}%")
