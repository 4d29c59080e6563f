#!/bin/bash
# round 4, call a: f16 MFMA subnormal probe, cross-stream discriminating experiments, long-track resampler tests, per-kernel clocks
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/dbg/f16_denorm > gpurun_out/r04_f16_denorm.txt 2>&1; cat gpurun_out/r04_f16_denorm.txt
timeout -k 10 400 python3 scripts/dbg/discriminate.py > gpurun_out/r04_xstream_matrix.txt 2>&1; echo "discriminate rc $?"; tail -12 gpurun_out/r04_xstream_matrix.txt | cut -c1-400
timeout -k 10 300 python3 -m pytest tests/test_vr_frontend.py -m gpu -x -q -k "resamplers" > gpurun_out/r04_a_pytest.txt 2>&1; tail -3 gpurun_out/r04_a_pytest.txt
rm -rf gpurun_out/clk
timeout -k 10 500 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/clk -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-precision > gpurun_out/r04_clk.log 2>&1; echo "clk rc $?"
python3 scripts/kernel_clocks.py gpurun_out/clk > gpurun_out/r04_kernel_clocks.txt 2>&1; head -30 gpurun_out/r04_kernel_clocks.txt | cut -c1-200
find gpurun_out/clk -name "*.csv" -size +4M -delete
