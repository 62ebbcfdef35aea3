#!/bin/bash
# Round 3: attention backward (dK/dV kernel: lse / delta staged through LDS) -- kernel tests, then a same-box A/B of the
# training step's kernel families between the previous library build and this one (PF_HIP_LIB).
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
TAG=${1:-r3u}
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q -p no:cacheprovider -n 4 -k "attention_lse or warpattn or denoiser_training or full_width_training" 2>&1 | tail -n 4
for lib in panfusion_amd/libpanfusion_hip_prev.so "" panfusion_amd/libpanfusion_hip_prev.so ""; do
  echo "== lib: ${lib:-current}"
  PF_HIP_LIB=$lib timeout 300 python tools/train_bench.py --steps 3 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-200
done 2>&1 | tee gpurun_out/${TAG}_ab_attn_bwd.txt
