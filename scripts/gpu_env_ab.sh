#!/bin/bash
# Same-box A/B of environment switches via kernel traces: usage gpu_env_ab.sh "<kernel grep -E pattern>" "VAR=val" ["VAR=val" ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
pat=$1; shift
: > gpurun_out/env_ab.log
run() {
  label=$1; shift
  rm -rf gpurun_out/prof_ab
  ( [ $# -gt 0 ] && export "$@"; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/ab_run.log 2>&1 )
  python scripts/trace_summary.py "$(find gpurun_out/prof_ab -name '*kernel_trace.csv' | head -1)" 80 | grep -E "$pat|big_kernel<2>" | sed "s/^/$label: /" | cut -c1-170 | tee -a gpurun_out/env_ab.log
  tail -1 gpurun_out/ab_run.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label: ms/step', d['ms_per_step'])" | tee -a gpurun_out/env_ab.log
  rm -rf gpurun_out/prof_ab
}
run base
for combo in "$@"; do run "$combo" $combo; done
run base2
