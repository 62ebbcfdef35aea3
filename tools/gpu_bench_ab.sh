#!/bin/bash
# usage: gpu_bench_ab.sh VAR=val ...   -- bench default vs each env setting (no tests)
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { tail -n 1 $1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read()); rf = r['roofline']
    print('$2', round(r['value'],2), 'steps/s', round(r['ms_per_step'],2), 'ms | gemm', round(rf['achieved']), 'TF/s', round(rf['share_of_step_time']*r['ms_per_step'],1), 'ms | other', {k:(round(v['tflops']), round(v['ms_per_step'],1)) for k,v in rf['other'].items()})
except Exception as e:
    print('$2 bench failed', e)
"; }
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_default.log 2>&1; summ gpurun_out/bench_default.log default
for kv in "$@"; do
  timeout 600 env $kv python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "gpurun_out/bench_${kv//\//_}.log" 2>&1; summ "gpurun_out/bench_${kv//\//_}.log" $kv
done
