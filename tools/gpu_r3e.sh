#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
TAG=${1:-r3e}
echo "== kernel tests (moments)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "moments or raw_pair or groupnorm" 2>&1 | grep -v amdgpu.ids | tail -n 6 | cut -c1-300 | tee gpurun_out/${TAG}_kernel_tests.log
echo "== A/B of the step"
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
run "PF_GN_EPILOGUE=1" ""
run "PF_GN_EPILOGUE=0" ""
run "PF_GN_EPILOGUE=1 PF_VIEW_PRIORITY=1" ""
run "PF_GN_EPILOGUE=1" "--no-graphs"
run "PF_GN_EPILOGUE=1 PF_VIEW_PRIORITY=1" "--no-graphs"
done | tee gpurun_out/${TAG}_ab.txt
