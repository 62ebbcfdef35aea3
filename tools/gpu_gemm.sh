#!/bin/bash
# conv-GEMM parity tests, then the GEMM table of the step's layer shapes (+ phase stamps of three representative ones)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear or conv or split or gemm or ff or geglu or persistent" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 3 gpurun_out/pytest_k.log
python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench.txt
PF_GEMM8_PERSIST=0 python tools/gemm_bench.py --reps 5 --phases --shapes conv64,lin320,ff1_320 2>&1 | grep -v amdgpu.ids | grep -v per-wave | tee gpurun_out/gemm_phases.txt
