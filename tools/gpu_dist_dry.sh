#!/bin/bash
# Functional dry run of the sharded path: N ranks on ONE GPU over gloo (the timing means nothing).
#   NS="2 4 8" ARGS="--cfg5" bash tools/gpu_dist_dry.sh
mkdir -p gpurun_out
export TMPDIR=/tmp PF_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1
for N in ${NS:-2 4}; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29511 + N)) bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline $ARGS > gpurun_out/dist_dry_$N.log 2>&1
  echo "N=$N $ARGS rc=$?"; tail -n 1 gpurun_out/dist_dry_$N.log | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read()); d = r.get('distributed') or {}
    print('  value %.2f steps/s (meaningless here)  parallelism: %s' % (r['value'], r['config'].get('parallelism')))
    print('  distributed: backend %s  world_size_initialised %s  rank ms/step min %.1f max %.1f  graphs_in_use %s  async_collectives %s' % (d.get('backend'), d.get('world_size_initialised'), d['rank_ms_per_step']['min'], d['rank_ms_per_step']['max'], d.get('graphs_in_use'), d.get('async_collectives')))
    for k, v in d.get('collectives_rank0', {}).items():
        print('    %-72s %5.1f calls/step  %12d bytes/rank/call' % (k, v['calls_per_step'], v['bytes_per_rank_per_call']))
except Exception as e:
    print('  (no JSON line)', e)
"
done
