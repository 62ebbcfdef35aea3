#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke()"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 4
echo "== cfg4 (1024x2048 pano)"
timeout 900 python bench.py --cfg4 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg4.log 2>&1; tail -n 1 gpurun_out/bench_cfg4.log | cut -c1-700
