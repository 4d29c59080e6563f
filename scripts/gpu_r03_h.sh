#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
prof() {
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" > $O/prof_$tag.log 2>&1
  echo "prof $tag rc=$?"; grep "ms for" $O/prof_$tag.log | cut -c1-300
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/r03_${tag}_kernel_stats.csv
}
ALSEP_RUNNER_LANES=1 prof melband_half_1lane python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03_melband_half_1lane_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms (2 runs of 120 s)", tot/1e6)
for r in rows[:14]:
    print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>7s} total_ms {float(r["TotalDurationNs"])/1e6:9.1f} avg_us {float(r["AverageNs"])/1e3:9.1f} {float(r["Percentage"]):5.1f}%')
PY
