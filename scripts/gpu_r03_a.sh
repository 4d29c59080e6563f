#!/bin/bash
# round 3, first GPU call: the new configs[3]/[4] tests + per-depth storage table, the bench line with precision/accuracy,
# and rocprofv3 kernel stats of the fp32 model families (one model each, 120 s of audio)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_configs.py "tests/test_gpu_parity.py::test_half_storage_error_per_depth" -m gpu -q -s -x > $O/r03_configs_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/r03_configs_tests.log
tail -5 $O/r03_configs_tests.log
timeout -k 10 600 python bench.py > $O/r03_bench_a.log 2> $O/r03_bench_a.err
echo "bench rc=$?"; tail -c 3000 $O/r03_bench_a.log
for m in vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt MDX23C-8KFFT-InstVoc_HQ.ckpt htdemucs_6s.yaml 17_HP-Wind_Inst-UVR.pth; do
  tag=${m%%.*}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python3 scripts/bench_lanes.py $m > $O/prof_$tag.log 2>&1
  echo "prof $m rc=$?"; grep "ms for" $O/prof_$tag.log
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/r03_${tag}_kernel_stats.csv
  find $O/prof_$tag -name "*.csv" ! -name "*stats*" -delete
done
ls -la $O/*.csv
