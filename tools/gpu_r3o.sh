#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
TAG=${1:-r3o}
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | tail -n 6 | cut -c1-300
echo "== microbench FF1"
python tools/gemm_bench.py --reps 20 --phases --shapes ff1_320,ff1_640,ff1_1280 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ff1.txt
echo "== step"
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-training-leg --trace-out gpurun_out/${TAG}_shapes.txt 2>&1 | tail -n 1 > gpurun_out/${TAG}_b.json
  python - gpurun_out/${TAG}_b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%.3f steps/s  %.2f ms  gemm %.1f TF/s (%.2f ms)  attn %.2f ms" % (d["value"], d["ms_per_step"],
      d["roofline"]["achieved"], d["roofline"]["launches_per_step"] * d["roofline"]["avg_launch_us"] / 1e3, d["roofline"]["other"]["k_attention"]["ms_per_step"]))
PY
done | tee gpurun_out/${TAG}_ab.txt
grep -E "N2560 K320|N5120 K640|N10240 K1280" gpurun_out/${TAG}_shapes.txt
