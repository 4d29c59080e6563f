#!/bin/bash
# round 3: per-depth storage table (tests print it), rocprofv3 kernel stats of the fp32 model families (one model each, 120 s of audio)
# and of the MDX bench step in fp32.  Raw traces go to /tmp (gpurun_out is copied back and capped at 64 MiB); only the stats travel.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest "tests/test_gpu_parity.py::test_half_storage_error_per_depth" -m gpu -q -s > $O/r03_per_depth.log 2>&1
echo "per-depth rc=$?"; grep -A8 "storage per depth" $O/r03_per_depth.log
prof() {  # tag, then the program and its arguments
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" > $O/prof_$tag.log 2>&1
  echo "prof $tag rc=$?"; grep "ms for\|\"value\"" $O/prof_$tag.log | cut -c1-300
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/r03_${tag}_kernel_stats.csv
}
for m in vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt MDX23C-8KFFT-InstVoc_HQ.ckpt htdemucs_6s.yaml 17_HP-Wind_Inst-UVR.pth UVR-DeNoise.pth; do
  prof ${m%%.*} python3 scripts/bench_lanes.py $m
done
prof mdx_f32 python3 bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-precision --batch 13
ls -la $O/*.csv
