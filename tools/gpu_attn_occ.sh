#!/bin/bash
# attention occupancy variants: parity tests under the non-default switches, then the microbench per setting
mkdir -p gpurun_out
export TMPDIR=/tmp
for kv in "PF_ATTENTION_OCC32=4" "PF_ATTENTION_OCC32=5" "PF_ATTENTION_OCC=2"; do
  env $kv timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "attention or attn or epa" 2>&1 | tail -n 2 | cut -c1-200
done
for kv in "PF_ATTENTION_OCC32=3" "PF_ATTENTION_OCC32=4" "PF_ATTENTION_OCC32=5"; do
  echo "== $kv"
  env $kv python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "self64|epa"
done | tee gpurun_out/attn_occ.txt
