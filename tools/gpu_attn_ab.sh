#!/bin/bash
# attention microbench across library builds: gpu_attn_ab.sh lib.so ...
mkdir -p gpurun_out
for lib in "" "$@"; do
  echo "== lib: ${lib:-current}"
  PF_HIP_LIB=$lib python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "self64|self32|pano64"
done | tee gpurun_out/attn_ab.txt
