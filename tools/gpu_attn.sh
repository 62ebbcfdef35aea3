#!/bin/bash
# attention parity tests, then the attention microbench for the in-tree build and any other builds given
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "attention or attn or epa" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 5 gpurun_out/pytest_k.log
for lib in "" "$@"; do
  echo "== lib: ${lib:-current}"
  PF_HIP_LIB=$lib python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/attn_bench.txt
