#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
R=$GRAFT_REPO_ROOT
TAG=${1:-r3f}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_e -o t -- python $R/bench.py --no-graphs --steps 3 --warmup 1 --no-cpu-baseline --no-training-leg > $R/gpurun_out/${TAG}_rocprof.log 2>&1
T=$(find $R/gpurun_out/${TAG}_e -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_streams.py $T $R/gpurun_out/${TAG}_streams.txt
rm -rf $R/gpurun_out/${TAG}_e
cat $R/gpurun_out/${TAG}_streams.txt
