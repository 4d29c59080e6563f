#!/usr/bin/env python3
"""Where a launch-bound workload's wall time goes: from a rocprofv3 --kernel-trace CSV, the busy time, the idle time between consecutive
dispatches (next start - previous end, in start order) and which kernels the idle time sits in front of."""
import collections, csv, glob, os, re, sys

path = sys.argv[1] if len(sys.argv) > 1 else max(glob.glob("gpurun_out/prof/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5            # analyse the last `tail` share of the trace (the timed steps)
ev = ev[int(len(ev) * (1 - tail)):]
span = ev[-1][1] - ev[0][0]
busy = sum(e - s for s, e, _ in ev)
gaps = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
end = ev[0][1]
for s, e, name in ev[1:]:
    g = max(0, s - end)
    m = re.search(r"(\w+_kernel)", name)
    short = m.group(1) if m else name[:40]
    gaps[short][0] += 1
    gaps[short][1] += g
    hist[min(int(g / 1000) // 5 * 5, 100)] += 1
    end = max(end, e)
idle = sum(v[1] for v in gaps.values())
print(f"{path}: last {tail:.0%}: {len(ev)} dispatches over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle between dispatches {idle / 1e6:.2f} ms "
      f"({idle / len(ev) / 1e3:.1f} us per dispatch)")
print("idle in front of (kernel: dispatches, mean us, total ms):")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {k:36s} {v[0]:6d} {v[1] / v[0] / 1e3:8.1f} {v[1] / 1e6:8.2f}")
print("gap histogram (us bucket: count): " + "  ".join(f"{b}{'+' if b == 100 else ''}:{c}" for b, c in sorted(hist.items())))
