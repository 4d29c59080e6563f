#!/bin/bash
# round 4, call b: cross-stream matrix with the instruction-class victim, the no-MFMA aggressor, and the FFT kernels compiled without packed-f32 SLP
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/dbg/discriminate.py > gpurun_out/r04_xstream_matrix2.txt 2>&1; echo "rc $?"; grep -E "^aggressor|part" gpurun_out/r04_xstream_matrix2.txt | cut -c1-500
echo "--- libalsep with fft.hip compiled -fno-slp-vectorize (no compiler-made v_pk_*_f32 in stft_kernel<8192 / 2048>; the 4096 kernel's are hand-written)"
DBG_LIB=libalsep_noslp.so XS_VICTIMS=stft8192,stft2048,stft4096 XS_AGGRESSORS=none,neutral:65536:1,lib_conv_hh timeout -k 10 200 python3 scripts/dbg/discriminate.py > gpurun_out/r04_xstream_matrix2_noslp.txt 2>&1; echo "rc $?"; grep -E "^aggressor|part" gpurun_out/r04_xstream_matrix2_noslp.txt | cut -c1-300
