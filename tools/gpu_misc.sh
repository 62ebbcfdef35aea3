#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== cfg5 (layout-conditioned)"
timeout 900 python bench.py --cfg5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg5.log 2>&1; tail -n 3 gpurun_out/bench_cfg5.log | cut -c1-500
