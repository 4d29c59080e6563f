#!/bin/bash
# round 4, call f: non-temporal streaming stores A/B (variant library scripts/dbg/libalsep_nt.so), same box, bench step + kernel traces
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/r04_nt_stores_ab.txt
cp audiolab_amd/lib/libalsep.so /tmp/libalsep_keep.so
run() {
  label=$1; lib=$2
  cp "$lib" audiolab_amd/lib/libalsep.so
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label: step %.2f ms | stft_first_conv %.1f us frac %.3f | roofline %s %.0f us frac %.3f' % (d['ms_per_step'], d['stages']['stft_first_conv']['us_per_launch'], d['stages']['stft_first_conv']['frac'], d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac']))" | tee -a gpurun_out/r04_nt_stores_ab.txt
  rm -rf gpurun_out/prof_ab
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-precision > gpurun_out/ab_run.log 2>&1
  python3 scripts/trace_summary.py "$(find gpurun_out/prof_ab -name '*kernel_trace.csv' | head -1)" 60 | grep -E "tdf_bf16_wide_kernel<4><res>|us_stream|ds48|ds_split|m0_kernel|mq_kernel|final_conv|stft_r16" | sed "s/^/$label: /" | cut -c1-150 | tee -a gpurun_out/r04_nt_stores_ab.txt
  rm -rf gpurun_out/prof_ab
}
run "plain stores" /tmp/libalsep_keep.so
run "nt stores   " scripts/dbg/libalsep_nt.so
run "plain again " /tmp/libalsep_keep.so
run "nt again    " scripts/dbg/libalsep_nt.so
cp /tmp/libalsep_keep.so audiolab_amd/lib/libalsep.so
