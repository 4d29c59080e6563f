#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -s --timeout 420 -k "full_size" > gpurun_out/f16_tests.log 2>&1
grep -n "full-size\|passed\|failed" gpurun_out/f16_tests.log
for d in bf16 f16 bf16 f16; do
  timeout -k 10 300 python bench.py --dtype $d --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$d.log 2>&1
  python - "$d" <<'PY'
import json, sys
d = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/bench_{d}.log").read().strip().splitlines()[-1])
    print(d, "ms/step", j["ms_per_step"], "value", j["value"], "big<2>", j["roofline"]["avg_us"])
except Exception as e:
    print(d, "failed", e, open(f"gpurun_out/bench_{d}.log").read()[-800:])
PY
done 2>&1 | tee gpurun_out/f16_ab.txt
