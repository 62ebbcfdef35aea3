#!/bin/bash
# usage: gpu_ab.sh "<pytest -k expr>" VAR=val [VAR=val ...]   -- targeted kernel tests, then bench default vs each env setting
mkdir -p gpurun_out
export TMPDIR=/tmp
K="$1"; shift
if [ -n "$K" ]; then
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "$K" 2>&1 | tail -n 30 | cut -c1-300 > gpurun_out/pytest_k.log; tail -n 8 gpurun_out/pytest_k.log
fi
summ() { tail -n 1 $1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read()); rf = r['roofline']
    print('$2', round(r['value'],2), 'steps/s', round(r['ms_per_step'],2), 'ms | gemm', round(rf['achieved']), 'TF/s', round(rf['share_of_step_time']*r['ms_per_step'],1), 'ms | other', {k:(round(v['tflops']), round(v['ms_per_step'],1)) for k,v in rf['other'].items()})
except Exception as e:
    print('$2 bench failed', e)
"; }
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --trace-out gpurun_out/trace_shapes.txt > gpurun_out/bench_default.log 2>&1; summ gpurun_out/bench_default.log default
for kv in "$@"; do
  timeout 600 env $kv python bench.py --steps 5 --warmup 2 --no-cpu-baseline --trace-out gpurun_out/trace_shapes_$kv.txt > gpurun_out/bench_$kv.log 2>&1; summ gpurun_out/bench_$kv.log $kv
done
head -n 40 gpurun_out/trace_shapes.txt
