#!/bin/bash
# Round-4 evidence, call A: PMC traffic (two passes; the bench line reads the resulting file, stamped with the kernel-source hash) and SQ counters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  step 500 "pmc $c" rocprofv3 --pmc $c --output-format csv -d /tmp/pmc/$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-precision > gpurun_out/pmc_$c.log 2>&1
done
python3 scripts/pmc_summary.py /tmp/pmc > /dev/null
W=$(grep -o 'windows/launch=[0-9]*' gpurun_out/pmc_FETCH_SIZE.log | tail -1 | cut -d= -f2)
python3 scripts/pmc_traffic.py /tmp/pmc/pmc_summary.json gpurun_out/pmc_traffic.json ${W:-} | head -16
rm -rf /tmp/pmc_sq
step 500 "sq a" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_sq/a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-precision --seconds 60 > gpurun_out/pmc_sq.log 2>&1
step 500 "sq b" rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq/b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-precision --seconds 60 >> gpurun_out/pmc_sq.log 2>&1
python3 scripts/pmc_summary.py /tmp/pmc_sq > /dev/null
python3 scripts/sq_table.py /tmp/pmc_sq/pmc_summary.json > gpurun_out/sq_table.txt 2>&1; head -12 gpurun_out/sq_table.txt | cut -c1-200
