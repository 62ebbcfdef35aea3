#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
TAG=${1:-r3n}
{
for L in even pano_rank; do
  PF_SHARD_LAYOUT=$L python tools/sim_rank.py --world 8 --ranks 0,1,3 2>&1 | grep "^world"
  PF_SHARD_LAYOUT=$L python tools/sim_rank.py --world 4 --ranks 0,1 2>&1 | grep "^world"
done
PF_SHARD_LAYOUT=pano_rank PF_SHARD_SPLIT=2,6,6,6 python tools/sim_rank.py --world 8 --ranks 0,1 2>&1 | grep "^world"
PF_SHARD_LAYOUT=pano_rank PF_SHARD_SPLIT=1,7,6,6 python tools/sim_rank.py --world 8 --ranks 0,1 2>&1 | grep "^world"
} | tee gpurun_out/${TAG}_sim_layouts.txt
