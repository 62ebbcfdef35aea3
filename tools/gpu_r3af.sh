#!/bin/bash
# LoRA gradients by weighted column sums: tests, then same-box A/B of the training step (previous commit's host code in tools/scratch/prev)
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; TAG=${1:-r3af}
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -n 4 -x 2>&1 | tail -n 8 | cut -c1-300
for rep in 1 2; do
for side in prev cur; do
  if [ $side = prev ]; then D=$R/tools/scratch/prev; else D=$R; fi
  echo "== $side"
  (cd $D && PF_HIP_LIB=$R/panfusion_amd/libpanfusion_hip.so timeout 300 python tools/train_bench.py --steps 4 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-230)
done
done 2>&1 | tee gpurun_out/${TAG}_ab_lora_grads.txt
