#!/bin/bash
# three-pass 7680 iSTFT: FFT / parity / configs tests, stage timing
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_configs.py tests/test_seam.py -m gpu -x -v > $O/r03_r_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/r03_r_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-precision --no-cpu-baseline > $O/r03_bench_r.json 2> $O/r03_bench_r.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_r.json').read().strip().splitlines()[-1])
print(d['ms_per_step'])
for k,v in d['stages'].items(): print(k, v['frac'], v['us_per_launch'], v.get('us_per_chunk'))
PY
