#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear or conv or split" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 3 gpurun_out/pytest_k.log
python tools/gemm_bench.py --reps 20 --shapes conv16,conv32,ff1_1280,ff1_640,lin1280 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench.txt
PF_GEMM_TAIL_SPLIT=0 python tools/gemm_bench.py --reps 20 --shapes conv16,conv32,ff1_1280,ff1_640,lin1280 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench_old.txt
