#!/bin/bash
# SQ counters (one pass, 8 slots) for the bench kernels.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq/a -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 > gpurun_out/pmc_sq.log 2>&1
tail -2 gpurun_out/pmc_sq.log | cut -c1-200
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_sq/b -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 >> gpurun_out/pmc_sq.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_sq | grep -E "conv3x3_bf16_kernel<64>|tdf_bf16|stft|pix" | cut -c1-400
find gpurun_out/pmc_sq -name "*counter_collection.csv" -size +8M -delete
