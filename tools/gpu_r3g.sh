#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
TAG=${1:-r3g}
run() {
  N=$(echo "$1 $2" | tr ' =-' '___')
  env $1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-training-leg $2 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench_$N.json
  python - "$1 $2" gpurun_out/${TAG}_bench_$N.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-60s %.3f steps/s  %.2f ms  gemm %.1f TF/s (%d launches, %.2f ms)  attn %.2f ms" % (sys.argv[1], d["value"], d["ms_per_step"],
          d["roofline"]["achieved"], d["roofline"]["launches_per_step"], d["roofline"]["launches_per_step"] * d["roofline"]["avg_launch_us"] / 1e3,
          d["roofline"]["other"]["k_attention"]["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-300:])
PY
}
for i in 1 2; do
run "PF_PANO_PRIORITY=0" ""
run "PF_PANO_PRIORITY=1" ""
run "PF_PANO_PRIORITY=0" "--no-graphs"
run "PF_PANO_PRIORITY=1" "--no-graphs"
done | tee gpurun_out/${TAG}_ab.txt
