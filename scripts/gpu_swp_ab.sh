#!/bin/bash
# Same-box A/B of the big-tile conv's software-pipelined k-loop (ALSEP_CONV_BIG_SWP) + the tests that cover it.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_parity.py -m gpu -q -s --timeout 420 -k "dispatch_orders or big_tile or full_size_mdx_bf16" > gpurun_out/swp_tests.log 2>&1
tail -3 gpurun_out/swp_tests.log
for rep in 1 2; do
for v in 0 1 2; do
  ALSEP_CONV_BIG_SWP=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/swp_bench_$v.log 2>&1
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/swp_bench_{v}.log").read().strip().splitlines()[-1])
    k = d["kernels"].get("conv3x3_bf16_big_kernel<3>", {})
    print(f"SWP={v} ms/step {d['ms_per_step']} big<2> avg_us {d['roofline']['avg_us']} TF {d['roofline']['achieved']} big<3> avg_us {k.get('avg_us')}")
except Exception as e:
    print("SWP", v, "failed", e)
PY
done
done 2>&1 | tee gpurun_out/swp_ab.txt
