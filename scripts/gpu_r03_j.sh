#!/bin/bash
# fused front-end rework: bit-identity tests, the bf16/f16 full-size parity tests, bench (stage line)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fused_front.py tests/test_gpu_parity.py -m gpu -q -x --durations=5 > $O/r03_j_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 $O/r03_j_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-precision > $O/r03_bench_j.json 2> $O/r03_bench_j.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_j.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
for k,v in d['stages'].items(): print(k, v['frac'], v['us_per_launch'])
PY
