#!/bin/bash
# round 4, call k: one vs two graph captures per runner lane (Mel-Band / BS Roformer half precision; htdemucs_6s with graphs)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r04_graphs_per_lane.txt; : > $out
run() {   # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/k_tmp.json 2> gpurun_out/k_tmp.err
  rc=$?
  [ $rc -eq 0 ] || { echo "$label rc $rc"; tail -5 gpurun_out/k_tmp.err; exit $rc; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/k_tmp.json').read().strip().splitlines()[-1])
print('$label', d['ms_per_step'], 'ms', d.get('realtime_factor'))" | tee -a $out
}
run "mel f16 1 capture/lane " ALSEP_RUNNER_GRAPHS_PER_LANE=1 -- --workload model --model vocals_mel_band_roformer.ckpt --dtype f16
run "mel f16 2 captures/lane" ALSEP_RUNNER_GRAPHS_PER_LANE=2 -- --workload model --model vocals_mel_band_roformer.ckpt --dtype f16
run "mel f16 3 captures/lane" ALSEP_RUNNER_GRAPHS_PER_LANE=3 -- --workload model --model vocals_mel_band_roformer.ckpt --dtype f16
run "bs  f16 2 captures/lane" ALSEP_RUNNER_GRAPHS_PER_LANE=2 -- --workload model --model model_bs_roformer_ep_368_sdr_12.9628.ckpt --dtype f16
run "mdx23c f16 2 captures/lane" ALSEP_RUNNER_GRAPHS_PER_LANE=2 -- --workload model --model MDX23C-8KFFT-InstVoc_HQ.ckpt --dtype f16
run "demucs6 plain launches, 4 lanes" ALSEP_DEMUCS_GRAPH=0 -- --workload demucs6
run "demucs6 graphs, 2 captures/lane, 4 lanes" ALSEP_DEMUCS_GRAPH=1 ALSEP_RUNNER_GRAPHS_PER_LANE=2 -- --workload demucs6
run "demucs6 graphs, 2 captures/lane, 2 lanes" ALSEP_DEMUCS_GRAPH=1 ALSEP_RUNNER_GRAPHS_PER_LANE=2 ALSEP_DEMUCS_LANES=2 -- --workload demucs6
run "demucs6 graphs, 3 captures/lane, 1 lane" ALSEP_DEMUCS_GRAPH=1 ALSEP_RUNNER_GRAPHS_PER_LANE=3 ALSEP_DEMUCS_LANES=1 -- --workload demucs6
