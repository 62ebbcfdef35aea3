#!/bin/bash
# Round-1 iteration 4 (re-entry): full gpu test suite, bench, rocprof kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== gpu tests"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | tail -n 60 | cut -c1-300 > gpurun_out/pytest_gpu.log; tail -n 5 gpurun_out/pytest_gpu.log
echo "== bench (with cpu baseline)"
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log | cut -c1-3000
echo "== rocprof kernel stats"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs > $R/gpurun_out/rocprof.log 2>&1
find $R/gpurun_out/prof_r1 -type f | head -20
find $R/gpurun_out/prof_r1 -type f -size +2M -delete
head -n 40 $(find $R/gpurun_out/prof_r1 -name '*kernel_stats.csv' | head -1) | cut -c1-200
