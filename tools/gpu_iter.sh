#!/bin/bash
# usage: gpu_iter.sh "<pytest -k expr for test_gpu_kernels>" [model]   -- targeted tests + bench with per-shape trace
mkdir -p gpurun_out
export TMPDIR=/tmp
K="$1"
if [ -n "$K" ]; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "$K" --durations=5 2>&1 | tail -n 40 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 12 gpurun_out/pytest_k.log
fi
if [ "$2" = "model" ]; then
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -n 4 --durations=5 2>&1 | tail -n 40 | cut -c1-300 > gpurun_out/pytest_m.log; tail -n 12 gpurun_out/pytest_m.log
fi
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --trace-out gpurun_out/trace_shapes.txt > gpurun_out/bench_trace.log 2>&1
tail -n 1 gpurun_out/bench_trace.log | cut -c1-1500
head -n 45 gpurun_out/trace_shapes.txt
