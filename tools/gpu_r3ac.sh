#!/bin/bash
# Round 3: LoRA re-fold in place by one HIP launch per projection -- tests, refold timing, training step.
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
TAG=${1:-r3ac}
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_model.py tests/test_gpu_mixed.py -m gpu -q -p no:cacheprovider -n 4 2>&1 | tail -n 8 | cut -c1-300
timeout 300 python tools/scratch/refold_time.py 2>&1 | grep -v amdgpu.ids | tail -n 3
timeout 300 python tools/train_bench.py --steps 4 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-250 | tee gpurun_out/${TAG}_train.txt
timeout 300 python tools/train_bench.py --steps 4 --layout-cond 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-250 | tee gpurun_out/${TAG}_train_layout.txt
