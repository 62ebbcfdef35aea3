#!/bin/bash
# keep-activations training forward: tests, then same-box A/B PF_TRAIN_KEEP=0/1 for both training modes
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
TAG=${1:-r3au}
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -p no:cacheprovider -n 4 -s 2>&1 | grep -E "passed|failed|outputs|full-width|trainable ControlNet|whole training" | cut -c1-260
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 3
for rep in 1 2; do for k in 0 1; do
  echo "== PF_TRAIN_KEEP=$k"
  PF_TRAIN_KEEP=$k timeout 300 python tools/train_bench.py --steps 4 --no-trace 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c100-260
  PF_TRAIN_KEEP=$k timeout 300 python tools/train_bench.py --layout-cond --steps 4 --no-trace 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c100-260
done; done 2>&1 | tee gpurun_out/${TAG}_ab_keep.txt
