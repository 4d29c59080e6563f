#!/bin/bash
# round 4, call j: kernel trace of the Mel-Band Roformer workload (half precision) -> busy / idle split
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload model --model vocals_mel_band_roformer.ckpt --dtype f16 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_j_bench.log 2>&1
rc=$?; echo "prof rc $rc"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/prof/*/*_kernel_trace.csv | head -1)
python3 scripts/trace_gaps.py $f 0.6 | tee gpurun_out/r04_mel_half_trace_gaps.txt
cp $(ls /tmp/prof/*/*_kernel_stats.csv | head -1) gpurun_out/r04_roformer_mel_half_kernel_stats.csv
tail -1 gpurun_out/r04_j_bench.log | cut -c1-300
exit $rc
