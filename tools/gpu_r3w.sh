#!/bin/bash
# Round 3: query-split keys-stationary attention backward -- tests, then per-shape table and same-box A/B (PF_ATTN_BWD_QSPLIT).
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
TAG=${1:-r3w}
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q -p no:cacheprovider -n 4 -x -k "attention_lse or warpattn or denoiser_training or full_width_training or trainable_controlnet_training" 2>&1 | tail -n 6 | cut -c1-300
for q in 0 1 0 1; do
  echo "== PF_ATTN_BWD_QSPLIT=$q"
  PF_ATTN_BWD_QSPLIT=$q timeout 300 python tools/train_bench.py --steps 3 --shapes 2>&1 | grep -v amdgpu.ids | grep -E "training step|k_attention_bwd" | head -n 9 | cut -c1-160
done 2>&1 | tee gpurun_out/${TAG}_ab_qsplit.txt
