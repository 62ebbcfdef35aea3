#!/bin/bash
# A/B builds of the library (PF_HIP_LIB=path) on the GEMM microbench, with per-block phase stamps
# usage: gpu_lib_ab.sh [lib.so ...]   ("" = the in-tree build is always run first)
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$PF_AB_TESTS" ]; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear or conv or split or gemm or ff or geglu" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 5 gpurun_out/pytest_k.log
fi
SH=${PF_AB_SHAPES:-conv64,conv32,conv16,lin320,ff1_320,ff2_320,ff1_640,lin640,pano_conv32}
for lib in "" "$@"; do
  echo "== lib: ${lib:-current}"
  PF_HIP_LIB=$lib python tools/gemm_bench.py --reps 20 --shapes $SH 2>&1 | grep -v amdgpu.ids
  PF_GEMM8_PERSIST=0 PF_HIP_LIB=$lib python tools/gemm_bench.py --reps 5 --phases --shapes conv64,lin320,ff1_320 2>&1 | grep -v amdgpu.ids | grep -v "per-wave"
done 2>&1 | tee gpurun_out/lib_ab.txt
