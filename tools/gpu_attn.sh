#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "attention" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 5 gpurun_out/pytest_k.log
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_bench.txt
