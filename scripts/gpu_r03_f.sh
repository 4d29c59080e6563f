#!/bin/bash
# HBM traffic (PMC) of the half-precision GEMM / attention micro-benchmark: separate FETCH_SIZE and WRITE_SIZE passes
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 scripts/bench_gemm_h.py > $O/pmc_gemm_$c.log 2>&1
  echo "$c rc=$?"
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
    agg.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[2], k, "dispatches", len(v), "avg", sum(v) / len(v))
PY
done
