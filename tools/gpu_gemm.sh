#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
PF_GEMM8_WAVES=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear or conv or split" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 3 gpurun_out/pytest_k.log
PF_GEMM8_WAVES=4 python tools/gemm_bench.py --reps 20 --phases --shapes conv64 2>&1 | grep -v amdgpu.ids
PF_GEMM8_WAVES=4 python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench_w4.txt
python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench.txt
