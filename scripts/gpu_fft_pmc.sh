#!/bin/bash
# SQ counters of the STFT / iSTFT kernels (stage micro-benchmark), two passes of 8 counters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_fft
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_fft/a -- python3 scripts/bench_fft.py > gpurun_out/pmc_fft.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_fft/b -- python3 scripts/bench_fft.py >> gpurun_out/pmc_fft.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_fft | grep -E "stft" | cut -c1-600 | tee gpurun_out/pmc_fft_summary.txt
find gpurun_out/pmc_fft -name "*counter_collection.csv" -size +8M -delete
