#!/bin/bash
# Round 2 regression pass: whole gpu suite, cfg4 / cfg5 lines in the mixed scheme, sharded dry runs over gloo on one GPU.
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r2d}
echo "== whole gpu suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | tail -n 25 | cut -c1-300 | tee gpurun_out/${TAG}_pytest_gpu.log
echo "== cfg4 / cfg5 (fp16 mixed)"
for f in --cfg4 --cfg5; do timeout 600 python bench.py $f --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f %.2f steps/s %.2f ms  %s' % (d['value'], d['ms_per_step'], d['config']['precision'][:40]))"; done | tee gpurun_out/${TAG}_cfg45.txt
echo "== sharded dry runs (gloo, one GPU)"
for N in 2 4 8; do
  PF_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E '^\{|Error|error' | cut -c1-400
done | tee gpurun_out/${TAG}_dist_dry.txt
