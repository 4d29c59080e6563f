#!/bin/bash
# HIP graphs in the Demucs runner: tests, configs[2] / configs[3] lines with graphs on and off
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_htdemucs.py tests/test_engine_e2e.py -m gpu -q -x > $O/r03_o_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/r03_o_tests.log
[ $rc -eq 0 ] || exit $rc
for g in 1 0; do
  ALSEP_RUNNER_GRAPH=$g timeout -k 10 300 python bench.py --workload demucs6 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demucs6 graphs $g', d['ms_per_step'], d['value'])"
done
ALSEP_RUNNER_GRAPH=1 timeout -k 10 400 python bench.py --workload tracks --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tracks graphs 1', d['ms_per_step'], d['value'])"
