#!/bin/bash
# SQ counters of the half-precision GEMM / attention micro-benchmark
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
pass() {
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$tag -o p -- python3 scripts/bench_gemm_h.py > $O/pmc_gemm_$tag.log 2>&1
  echo "$tag rc=$?"
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if "gemm_h" not in r["Kernel_Name"] and "attn_h" not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"].split("(")[-2][-22:] if False else r["Kernel_Name"][27:52], r.get("Grid_Size", ""))
    agg.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq2 GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
