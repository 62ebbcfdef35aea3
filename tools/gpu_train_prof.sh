#!/bin/bash
# Training-step evidence: time per step + rocprofv3 kernel stats of the same command.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2_train}
timeout 600 python tools/train_bench.py --steps 5 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|% (args" > gpurun_out/${TAG}_bench.txt; cat gpurun_out/${TAG}_bench.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o t -- python $R/tools/train_bench.py --steps 3 > $R/gpurun_out/${TAG}_rocprof.log 2>&1
cp $(find $R/gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
find $R/gpurun_out/${TAG}_prof -type f -size +1M -delete
head -n 30 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
