#!/bin/bash
# two-stream training step: tests (3 passes of the parity tests to catch cross-stream races), then same-box A/B PF_TRAIN_STREAMS=0/1
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
TAG=${1:-r3ay}
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -p no:cacheprovider -n 4 -k "denoiser_training or full_width or whole_training or trainable_controlnet" 2>&1 | tail -n 1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2
for rep in 1 2; do for k in 0 1; do
  echo "== PF_TRAIN_STREAMS=$k"
  PF_TRAIN_STREAMS=$k timeout 300 python tools/train_bench.py --steps 6 --no-trace 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c100-260
  PF_TRAIN_STREAMS=$k timeout 300 python tools/train_bench.py --layout-cond --steps 6 --no-trace 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c100-260
done; done 2>&1 | tee gpurun_out/${TAG}_ab_train_streams.txt
