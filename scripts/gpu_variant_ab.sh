#!/bin/bash
# Same-box A/B of the production library against audiolab_amd/lib/libalsep_variant.so (a build with one constant changed):
# usage gpu_variant_ab.sh "<grep -E pattern of kernel names>"
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
pat=${1:-us_stream}
: > gpurun_out/variant_ab.log
timeout 900 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "streaming or net_ or full_size" 2>&1 | tail -2 | tee -a gpurun_out/variant_ab.log
grep -q failed gpurun_out/variant_ab.log && exit 1
run() {
  rm -rf gpurun_out/prof_ab
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/ab_run.log 2>&1
  python scripts/trace_summary.py "$(find gpurun_out/prof_ab -name '*kernel_trace.csv' | head -1)" 80 | grep -E "$pat|big_kernel<2>" | sed "s/^/$1: /" | cut -c1-170 | tee -a gpurun_out/variant_ab.log
  rm -rf gpurun_out/prof_ab
}
run base
cp audiolab_amd/lib/libalsep.so /tmp/libalsep_keep.so
cp audiolab_amd/lib/libalsep_variant.so audiolab_amd/lib/libalsep.so
run variant
cp /tmp/libalsep_keep.so audiolab_amd/lib/libalsep.so
run base2
