#!/bin/bash
# usage: gpu_model_ab.sh VAR=val ...   -- model-level gpu tests, then bench default vs each env setting
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -n 4 --durations=3 2>&1 | tail -n 40 | cut -c1-300 > gpurun_out/pytest_m.log; tail -n 8 gpurun_out/pytest_m.log
summ() { tail -n 1 $1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read()); rf = r['roofline']
    print('$2', round(r['value'],2), 'steps/s', round(r['ms_per_step'],2), 'ms | gemm', round(rf['achieved']), 'TF/s', round(rf['share_of_step_time']*r['ms_per_step'],1), 'ms | other', {k:(round(v['tflops']), round(v['ms_per_step'],1)) for k,v in rf['other'].items()})
except Exception as e:
    print('$2 bench failed', e)
"; }
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_default.log 2>&1; summ gpurun_out/bench_default.log default
for kv in "$@"; do
  timeout 600 env $kv python bench.py --steps 8 --warmup 2 --no-cpu-baseline > "gpurun_out/bench_${kv//\//_}.log" 2>&1; summ "gpurun_out/bench_${kv//\//_}.log" $kv
done
