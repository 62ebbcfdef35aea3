#!/bin/bash
# Functional dry run of the sharded path: N ranks on ONE GPU over gloo (the timing means nothing).
#   NS="2 4 8" ARGS="--cfg5" bash tools/gpu_dist_dry.sh
mkdir -p gpurun_out
export TMPDIR=/tmp PF_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1
for N in ${NS:-2 4}; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline $ARGS > gpurun_out/dist_dry_$N.log 2>&1
  echo "N=$N $ARGS rc=$?"; tail -n 1 gpurun_out/dist_dry_$N.log | cut -c1-330
done
