#!/bin/bash
# Round-end evidence: full gpu test suite, default bench (with cpu baseline), rocprofv3 kernel stats of the same
# command, serial per-kernel table, PMC traffic passes.   usage: gpu_final.sh <tag>
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-final}
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | tail -n 40 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu.log; tail -n 3 gpurun_out/${TAG}_pytest_gpu.log
echo "== bench (default)"
timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
cd /tmp
echo "== rocprofv3 --kernel-trace --stats of the default bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_g -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof_graphs.log 2>&1
cp $(find $R/gpurun_out/${TAG}_g -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_kernel_stats_default.csv 2>/dev/null
find $R/gpurun_out/${TAG}_g -type f -size +1M -delete
head -n 8 $R/gpurun_out/${TAG}_kernel_stats_default.csv | cut -c1-160
echo "== serial (one stream, no graphs) per-kernel table"
PF_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_s -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs > $R/gpurun_out/${TAG}_rocprof_serial.log 2>&1
T=$(find $R/gpurun_out/${TAG}_s -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $T $R/gpurun_out/${TAG}_kernels_serial.txt 10
find $R/gpurun_out/${TAG}_s -type f -size +1M -delete
head -n 16 $R/gpurun_out/${TAG}_kernels_serial.txt
echo "== PMC traffic passes"
for C in FETCH_SIZE WRITE_SIZE; do
  PF_STREAMS=1 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$C -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graphs > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
  P=$(find $R/gpurun_out/${TAG}_pmc_$C -name '*counter_collection.csv' | head -1)
  python $R/tools/prof_summary.py pmc $P $R/gpurun_out/${TAG}_pmc_$C.txt
  find $R/gpurun_out/${TAG}_pmc_$C -type f -size +1M -delete
  grep -E "k_conv_gemm|k_attention" $R/gpurun_out/${TAG}_pmc_$C.txt | cut -c1-200
done
cd $R
echo "== microbenchmarks"
python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_gemm_microbench.txt
PF_GEMM8_PERSIST=0 python tools/gemm_bench.py --reps 5 --phases --shapes conv64,lin320,ff1_320 2>&1 | grep -v amdgpu.ids | grep -v per-wave > gpurun_out/${TAG}_gemm_phases.txt
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_attn_microbench.txt
tail -n 4 gpurun_out/${TAG}_gemm_microbench.txt
