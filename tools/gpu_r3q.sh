#!/bin/bash
# Round 3: what the panorama stream runs in the segments where the view stream waits for it (tools/trace_streams.py).
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; TAG=${1:-r3q}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_e -o t -- python $R/bench.py --no-graphs --steps 3 --warmup 1 --no-cpu-baseline --no-training-leg > $R/gpurun_out/${TAG}_rocprof_eager.log 2>&1
python $R/tools/trace_streams.py $(find $R/gpurun_out/${TAG}_e -name '*kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_streams.txt
rm -rf $R/gpurun_out/${TAG}_e
tail -n 90 $R/gpurun_out/${TAG}_streams.txt | cut -c1-160
