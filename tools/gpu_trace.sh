#!/bin/bash
mkdir -p gpurun_out
PF_STREAMS=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --trace-out gpurun_out/trace_shapes.txt > gpurun_out/bench_trace.log 2>&1
tail -n 1 gpurun_out/bench_trace.log | cut -c1-300
