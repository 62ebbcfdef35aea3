#!/bin/bash
# Round 2, first GPU pass: new mixed-precision tests first (fail fast), then the whole gpu suite, the bench in the
# three precision configurations, and a serial kernel trace of the mixed scheme.   usage: gpu_r2a.sh <tag>
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2a}
echo "== new tests (mixed scheme kernels, exact grids)"
timeout 600 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "not full_width" 2>&1 | tail -n 30 | cut -c1-400 > gpurun_out/${TAG}_pytest_new.log; tail -n 12 gpurun_out/${TAG}_pytest_new.log
echo "== full-width e2e vs oracle (per-block table)"
timeout 900 python -m pytest tests/test_gpu_mixed.py -m gpu -q --tb=short -p no:cacheprovider -k "full_width" -s 2>&1 | grep -v amdgpu.ids | cut -c1-3000 > gpurun_out/${TAG}_fullwidth.log; grep -E "full-width|passed|failed|Error|assert" gpurun_out/${TAG}_fullwidth.log | cut -c1-300
echo "== whole gpu suite"
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 --deselect tests/test_gpu_mixed.py::test_full_width_denoiser_vs_oracle 2>&1 | tail -n 40 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu.log; tail -n 15 gpurun_out/${TAG}_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 3
echo "== bench fp16 mixed (default)"
timeout 900 python bench.py --trace-out gpurun_out/${TAG}_shapes_fp16_mixed.txt > gpurun_out/${TAG}_bench_fp16_mixed.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench_fp16_mixed.log > gpurun_out/${TAG}_bench_fp16_mixed.json; cut -c1-600 gpurun_out/${TAG}_bench_fp16_mixed.json
echo "== bench fp16 fast"
timeout 600 python bench.py --precision fast --no-cpu-baseline --trace-out gpurun_out/${TAG}_shapes_fp16_fast.txt > gpurun_out/${TAG}_bench_fp16_fast.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench_fp16_fast.log | cut -c1-300
echo "== bench bf16 fast"
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16_fast.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench_bf16_fast.log | cut -c1-300
cd /tmp
echo "== serial (one stream, no graphs) per-kernel table, fp16 mixed"
PF_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_s -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs > $R/gpurun_out/${TAG}_rocprof_serial.log 2>&1
T=$(find $R/gpurun_out/${TAG}_s -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $T $R/gpurun_out/${TAG}_kernels_serial.txt 10
find $R/gpurun_out/${TAG}_s -type f -size +1M -delete
head -n 24 $R/gpurun_out/${TAG}_kernels_serial.txt
