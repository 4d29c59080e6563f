#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from a pmc_summary.json (scripts/pmc_summary.py): HBM bytes per launch per
kernel class = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, launch-weighted over grids.  FETCH_SIZE is doubled as
MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950; WRITE_SIZE is exact."""
import collections
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiolab_amd.buildinfo import source_hash  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
wpl = int(sys.argv[3]) if len(sys.argv) > 3 else None       # model windows per network launch of the profiled run (bench.py --batch)
d = json.load(open(src))
agg = collections.defaultdict(lambda: {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
for key, cs in d.items():
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", key.split(" grid=")[0])
    if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    n = cs["FETCH_SIZE"]["launches"]
    a = agg[name]
    a["launches"] += n
    a["fetch_kb"] += cs["FETCH_SIZE"]["mean"] * n
    a["write_kb"] += cs["WRITE_SIZE"]["mean"] * cs["WRITE_SIZE"]["launches"]
out = {}
for name, a in agg.items():
    n = a["launches"]
    out[name] = {"launches": n, "fetch_size_kb_mean": a["fetch_kb"] / n, "write_size_kb_mean": a["write_kb"] / n,
                 "hbm_bytes_per_launch": (2 * a["fetch_kb"] + a["write_kb"]) / n * 1024,
                 "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, x1024"}
out["_build"] = {"source_hash": source_hash(), "windows_per_launch": wpl}    # bench.py refuses this file for any other kernel source
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
del out["_build"]
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
    print(f'{k:40s} n={v["launches"]:5d} hbm_MB/launch={v["hbm_bytes_per_launch"] / 1e6:9.1f}')
