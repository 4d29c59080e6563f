#!/bin/bash
# wide TDF epilogue in fragment layout: bit-identity / parity tests, bench
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_parity.py tests/test_fused_front.py -m gpu -q -x --durations=5 > $O/r03_n_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r03_n_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-precision --no-cpu-baseline > $O/r03_bench_n.json 2> $O/r03_bench_n.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_n.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
for k in d.get('kernels', {}):
    v=d['kernels'][k]; print(k, v if not isinstance(v, dict) else {a: v[a] for a in list(v)[:6]})
PY
rm -rf /tmp/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precision > $O/bench_prof_n.log 2>&1
python scripts/trace_summary.py "$(find /tmp/prof -name '*kernel_trace.csv' | head -1)" 60 > $O/trace_summary_n.txt; grep "tdf_bf16" $O/trace_summary_n.txt | cut -c1-160
