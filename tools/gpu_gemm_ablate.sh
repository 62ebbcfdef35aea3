#!/bin/bash
# Timing-only ablations of the 8-wave conv-GEMM K loop (wrong results): which part of a K step is what.
# Libraries built on the build host with -DPF_ABL_* (see the Makefile-free recipe in DESIGN.md §3.1) and shipped in panfusion_amd/abl/.
export TMPDIR=/tmp
SH=${1:-conv64,conv64cat,conv32,conv16}
for lib in "" panfusion_amd/abl/lib_abl_NOBARRIER.so panfusion_amd/abl/lib_abl_NODMA.so panfusion_amd/abl/lib_abl_NOLDS.so panfusion_amd/abl/lib_abl_NODMA_NOLDS.so panfusion_amd/abl/lib_abl_NODMA_NOLDS_NOBARRIER.so ""; do
  echo "== ${lib:-current}"
  PF_HIP_LIB=$lib python tools/gemm_bench.py --reps 20 --shapes $SH 2>&1 | grep -v amdgpu.ids | cut -c1-70
  PF_GEMM8_PERSIST=0 PF_HIP_LIB=$lib python tools/gemm_bench.py --reps 3 --phases --shapes conv64 2>&1 | grep "phases" | cut -c1-200
done
