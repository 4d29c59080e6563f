#!/bin/bash
# round 4, call e: what bounds tdf_bf16_wide_kernel<4><res>?  timing-only ablations (variant libraries, scripts/dbg/libalsep_tdfablN.so:
# bit 0 weight fragments loaded once, bit 1 no residual loads, bit 2 no X tiles beyond the first two), same box, kernel traces
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/r04_tdf_ablation.txt
cp audiolab_amd/lib/libalsep.so /tmp/libalsep_keep.so
run() {
  label=$1; lib=$2
  cp "$lib" audiolab_amd/lib/libalsep.so
  rm -rf gpurun_out/prof_ab
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision > gpurun_out/ab_run.log 2>&1
  python3 scripts/trace_summary.py "$(find gpurun_out/prof_ab -name '*kernel_trace.csv' | head -1)" 60 | grep -E "tdf_bf16_wide_kernel" | sed "s/^/$label: /" | cut -c1-150 | tee -a gpurun_out/r04_tdf_ablation.txt
  rm -rf gpurun_out/prof_ab
}
run "product            " /tmp/libalsep_keep.so
run "abl1 (W once)      " scripts/dbg/libalsep_tdfabl1.so
run "abl2 (no residual) " scripts/dbg/libalsep_tdfabl2.so
run "abl4 (no X stream) " scripts/dbg/libalsep_tdfabl4.so
run "abl7 (all three)   " scripts/dbg/libalsep_tdfabl7.so
run "product again      " /tmp/libalsep_keep.so
cp /tmp/libalsep_keep.so audiolab_amd/lib/libalsep.so
