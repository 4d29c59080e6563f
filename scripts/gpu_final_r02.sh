#!/bin/bash
# Round-2 evidence on one box: PMC traffic first (two passes; the bench line reads the resulting file, stamped with the kernel-source
# hash), then GPU tests, smoke, default bench (with cpu baseline), rocprof kernel stats, SQ counters (one pass).
# Outputs under gpurun_out/; scripts/collect_r02.sh copies the summaries into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 1200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py gpurun_out/pmc > /dev/null
W=$(grep -o 'windows/launch=[0-9]*' gpurun_out/pmc_FETCH_SIZE.log | tail -1 | cut -d= -f2)
python scripts/pmc_traffic.py gpurun_out/pmc/pmc_summary.json gpurun_out/pmc_traffic.json ${W:-} | head -12
find gpurun_out/pmc -name "*counter_collection.csv" -delete
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 420 > gpurun_out/pytest_gpu_full.log 2>&1; tail -3 gpurun_out/pytest_gpu_full.log | tee gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout -k 10 900 python bench.py 2>gpurun_out/bench_default.err | tail -1 | tee gpurun_out/bench_default.log | cut -c1-400
rm -rf gpurun_out/prof
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv
python scripts/trace_summary.py "$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)" 60 > gpurun_out/trace_summary.txt
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
rm -rf gpurun_out/pmc_sq
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq/a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 > gpurun_out/pmc_sq.log 2>&1
timeout -k 10 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq/b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 >> gpurun_out/pmc_sq.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_sq > /dev/null
python scripts/sq_table.py gpurun_out/pmc_sq/pmc_summary.json > gpurun_out/sq_table.txt 2>&1; head -8 gpurun_out/sq_table.txt | cut -c1-220
find gpurun_out/pmc_sq -name "*counter_collection.csv" -delete
