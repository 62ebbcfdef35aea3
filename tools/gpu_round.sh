#!/bin/bash
# One GPU-box session: kernel + model parity tests (crash-isolated via xdist), smoke, bench.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [quick]'
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
echo "== pytest -m gpu" 
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 -rfEs 2>&1 | tail -n 400 > gpurun_out/pytest_gpu.log
tail -n 5 gpurun_out/pytest_gpu.log
if [ "$1" != "quick" ]; then
  echo "== smoke"
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -n 3 gpurun_out/smoke.log
  echo "== bench small"
  timeout 300 python bench.py --small --steps 3 --warmup 1 > gpurun_out/bench_small.log 2>&1; tail -n 3 gpurun_out/bench_small.log
  echo "== bench"
  timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; tail -n 3 gpurun_out/bench.log
fi
