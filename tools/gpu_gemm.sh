#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear or conv or split" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 3 gpurun_out/pytest_k.log
for D in 0 1; do echo "== PF_GEMM_DEBUG=$D"; PF_GEMM_DEBUG=$D python tools/gemm_bench.py --reps 10 --phases --shapes conv64,lin1280 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_bench.txt; done
python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_bench.txt
