#!/bin/bash
# rocprofv3 kernel trace of an eager single-stream bench run, condensed to a per-(kernel, grid) table; optional PMC pass.
mkdir -p gpurun_out
export TMPDIR=/tmp
export PF_STREAMS=1
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs > $R/gpurun_out/${TAG}_rocprof.log 2>&1
T=$(find $R/gpurun_out/$TAG -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $T $R/gpurun_out/${TAG}_kernels.txt 10
find $R/gpurun_out/$TAG -type f -size +2M -delete
head -n 24 $R/gpurun_out/${TAG}_kernels.txt
if [ "$2" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$C -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graphs > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
    P=$(find $R/gpurun_out/${TAG}_pmc_$C -name '*counter_collection.csv' | head -1)
    python $R/tools/prof_summary.py pmc $P $R/gpurun_out/${TAG}_pmc_$C.txt
    find $R/gpurun_out/${TAG}_pmc_$C -type f -size +2M -delete
    grep -E "k_conv_gemm|k_attention" $R/gpurun_out/${TAG}_pmc_$C.txt | cut -c1-200
  done
fi
