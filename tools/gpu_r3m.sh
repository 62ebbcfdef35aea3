#!/bin/bash
# Round 3: secondary bench lines in the parity-passing scheme (cfg 4, cfg 5), per-rank compute simulation of the sharded loop
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
TAG=${1:-r3m}
echo "== kernel tests (after default changes)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-300
echo "== cfg4 / cfg5 bench lines (fp16 mixed)"
timeout 900 python bench.py --cfg4 --steps 10 --warmup 2 --no-cpu-baseline --no-training-leg --trace-out gpurun_out/${TAG}_shapes_cfg4.txt 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench_cfg4.json; cut -c1-330 gpurun_out/${TAG}_bench_cfg4.json
timeout 900 python bench.py --cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench_cfg5.json; cut -c1-330 gpurun_out/${TAG}_bench_cfg5.json
echo "== per-rank simulation"
bash tools/gpu_sim_ranks.sh 2>&1 | tail -n 12
cp gpurun_out/sim_ranks.txt gpurun_out/${TAG}_sim_ranks.txt
