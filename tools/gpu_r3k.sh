#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
TAG=${1:-r3k}
echo "== kernel tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | tail -n 12 | cut -c1-300 | tee gpurun_out/${TAG}_kernel_tests.log
echo "== model parity"
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_mixed.py tests/test_gpu_training.py tests/test_gpu_fullsize.py -m gpu -q -s --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | grep -E "rel-L2|passed|failed|Error|error|final" | cut -c1-300 | tee gpurun_out/${TAG}_parity.log
echo "== step"
for i in 1 2; do for V in "PF_SPLITK_INKERNEL=1" "PF_SPLITK_INKERNEL=0"; do
  env $V timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 > gpurun_out/${TAG}_b.json
  python - "$V" gpurun_out/${TAG}_b.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-40s %.3f steps/s  %.2f ms  gemm %.1f TF/s (%.2f ms)  attn %.2f ms" % (sys.argv[1], d["value"], d["ms_per_step"],
      d["roofline"]["achieved"], d["roofline"]["launches_per_step"] * d["roofline"]["avg_launch_us"] / 1e3, d["roofline"]["other"]["k_attention"]["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "FAILED", open(sys.argv[2]).read()[-400:])
PY
done; done | tee gpurun_out/${TAG}_ab.txt
