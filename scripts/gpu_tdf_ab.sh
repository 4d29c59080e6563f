#!/bin/bash
# Same-box A/B of the wide TDF kernel: dispatch order, residual prefetch, staging stride (variant library), via kernel traces
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/tdf_ab.log
run() {  # label, env...
  label=$1; shift
  rm -rf gpurun_out/prof_ab
  env "$@" ALSEP_AB=1 true
  ( export "$@"; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/ab_run.log 2>&1 )
  python scripts/trace_summary.py "$(find gpurun_out/prof_ab -name '*kernel_trace.csv' | head -1)" 60 | grep -E "tdf_bf16_wide_kernel<4><res> blocks=\( *(8192|512, 16|4096|512,  8)|big_kernel<2>|istft" | sed "s/^/$label: /" | cut -c1-170 | tee -a gpurun_out/tdf_ab.log
  rm -rf gpurun_out/prof_ab
}
run base ALSEP_X=0
run yfast0 ALSEP_TDF_YFAST=0
run rpf0 ALSEP_TDF_RPF=0
if [ -f audiolab_amd/lib/libalsep_ss48.so ]; then
  cp audiolab_amd/lib/libalsep.so /tmp/libalsep_keep.so
  cp audiolab_amd/lib/libalsep_ss48.so audiolab_amd/lib/libalsep.so
  run ss48 ALSEP_X=0
  cp /tmp/libalsep_keep.so audiolab_amd/lib/libalsep.so
fi
run base2 ALSEP_X=0
