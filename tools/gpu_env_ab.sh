#!/bin/bash
# Same-box A/B of environment switches on the default bench (fp16 mixed).  usage: gpu_env_ab.sh "VAR=val VAR2=val" "VAR=val" ...
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { env $1 python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %.2f steps/s  %.2f ms' % ('$1', d['value'], d['ms_per_step']))"; }
run "PF_NOP=0"
for cfg in "$@"; do run "$cfg"; done
run "PF_NOP=1"
