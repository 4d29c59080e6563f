#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_configs.py tests/test_gpu_parity.py -m gpu -q -x --durations=5 > $O/r03_suite_b.log 2>&1
echo "suite B rc=$?"; tail -10 $O/r03_suite_b.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r03_smoke.log
timeout -k 10 400 python bench.py > $O/r03_bench_b.json 2> $O/r03_bench_b.err; echo "bench rc=$?"; cut -c1-400 $O/r03_bench_b.json
