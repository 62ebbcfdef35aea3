#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --trace-out gpurun_out/trace_shapes.txt > gpurun_out/bench_trace.log 2>&1
tail -n 1 gpurun_out/bench_trace.log | cut -c1-600
head -n 70 gpurun_out/trace_shapes.txt
