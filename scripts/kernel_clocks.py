#!/usr/bin/env python3
"""Effective shader clock per kernel from a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace` run:
clock = (GRBM_GUI_ACTIVE / 8 XCDs) / (End - Start) per dispatch (MI355X_MICROARCH.md, DVFS give-back: within 3 % of the in-kernel clock on
dispatches of 0.3 ms and more), averaged per (kernel, grid).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles).
usage: kernel_clocks.py <dir with *counter_collection.csv and *kernel_trace.csv>"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
dur = {}
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
vals = collections.defaultdict(dict)
meta = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            d = r["Dispatch_Id"]
            vals[d][r["Counter_Name"]] = vals[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            name = r["Kernel_Name"]
            m = re.search(r"(\w+_kernel)", name)
            short = m.group(1) if m else name[:40]
            tw = re.search(r"kernelILi(\d+)", name)
            if tw:
                short += f"<{tw.group(1)}>"
            if "Lb1" in name:
                short += "<res>"
            meta[d] = f'{short} grid={int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)}'
            if "Start_Timestamp" in r and d not in dur and r.get("End_Timestamp"):
                dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d, cs in vals.items():
    if d not in dur or "GRBM_GUI_ACTIVE" not in cs or dur[d] <= 0:
        continue
    cyc = cs["GRBM_GUI_ACTIVE"] / 8
    a = agg[meta[d]]
    a[0] += 1
    a[1] += dur[d]
    a[2] += cyc
    a[3] += cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
print("kernel (grid) | launches | avg us | effective clock GHz (GRBM_GUI_ACTIVE / 8 / duration) | MFMA busy % | MFMA-busy-equivalent TFLOP/s ceiling at this clock (1024 SIMDs x 1024 flop/clk)")
for k, (n, t, cyc, mf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t / n < 20e-6:
        continue
    clk = cyc / t
    print("%-52s | %4d | %8.1f | %5.2f | %5.1f | %6.0f" % (k[:52], n, 1e6 * t / n, clk * 1e-9, 100 * mf / (1024 * cyc) if cyc else 0.0, clk * 1024 * 1024 * 1e-12))
