#!/bin/bash
# same-box A/B: previous commit's Python (tools/scratch/prev, current library) vs the tree, training step with / without layout cond
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; TAG=${1:-r3ad}
for rep in 1 2; do
for side in prev cur; do
  if [ $side = prev ]; then D=$R/tools/scratch/prev; else D=$R; fi
  for extra in "" "--layout-cond"; do
    echo "== $side $extra"
    (cd $D && PF_HIP_LIB=$R/panfusion_amd/libpanfusion_hip.so timeout 300 python tools/train_bench.py --steps 4 --no-trace $extra 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c1-260)
  done
done
done 2>&1 | tee gpurun_out/${TAG}_ab_refold.txt
