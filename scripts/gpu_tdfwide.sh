#!/bin/bash
# wide-tile TDF kernel: parity first, then A/B of the dispatch modes on the bench workload
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv_variants.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/tdfwide_pytest.log
: > gpurun_out/tdfwide.log
for m in 0 1 4 8; do
  ALSEP_TDF_WIDE=$m timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tdf_wide=$m','value',d['value'],'ms/step',d['ms_per_step'])" | tee -a gpurun_out/tdfwide.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_tdfwide" -o tdfwide -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/tdfwide_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/prof_tdfwide -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/tdfwide_kernel_stats.csv && head -12 "$f" | cut -c1-150
