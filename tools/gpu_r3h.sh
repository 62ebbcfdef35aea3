#!/bin/bash
# VAE decode kernel breakdown (what the 107 ms are made of)
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
R=$GRAFT_REPO_ROOT
TAG=${1:-r3h}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_v -o t -- python $R/tools/vae_bench.py --reps 2 > $R/gpurun_out/${TAG}_vae_rocprof.log 2>&1
T=$(find $R/gpurun_out/${TAG}_v -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $T $R/gpurun_out/${TAG}_vae_kernels.txt 3
rm -rf $R/gpurun_out/${TAG}_v
head -n 60 $R/gpurun_out/${TAG}_vae_kernels.txt
tail -n 3 $R/gpurun_out/${TAG}_vae_rocprof.log
