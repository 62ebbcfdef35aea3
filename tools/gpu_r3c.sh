#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
TAG=${1:-r3c}
echo "== kernel tests (moments)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "moments or raw_pair or groupnorm or linear or conv_gemm or split_k or persistent" 2>&1 | grep -v amdgpu.ids | tail -n 12 | cut -c1-300 | tee gpurun_out/${TAG}_kernel_tests.log
echo "== microbench with / without moments, with phase stamps"
for G in "" "--gn"; do
  echo "-- gemm_bench $G"
  python tools/gemm_bench.py --reps 20 --phases --stream32 --shapes conv64,conv32,lin320,lin640,pano_conv64 $G 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/${TAG}_gemm_gn.txt
echo "== model parity"
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_vae.py -m gpu -q -s --tb=short -p no:cacheprovider -n 3 2>&1 | grep -v amdgpu.ids | grep -E "rel-L2|passed|failed|Error|error|drift|final" | cut -c1-300 | tee gpurun_out/${TAG}_parity.log
echo "== A/B of the step"
for V in "PF_GN_EPILOGUE=1" "PF_GN_EPILOGUE=0" "PF_GN_EPILOGUE=1" "PF_GN_EPILOGUE=0"; do
  N=$(echo $V | tr ' =' '__')
  env $V timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-training-leg --trace-out gpurun_out/${TAG}_shapes_$N.txt 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench_$N.json
  python - "$V" gpurun_out/${TAG}_bench_$N.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-40s %.3f steps/s  %.2f ms  gemm %.1f TF/s (%d launches, %.2f ms)  attn %.2f ms" % (sys.argv[1], d["value"], d["ms_per_step"],
      d["roofline"]["achieved"], d["roofline"]["launches_per_step"], d["roofline"]["launches_per_step"] * d["roofline"]["avg_launch_us"] / 1e3,
      d["roofline"]["other"]["k_attention"]["ms_per_step"]))
PY
done | tee gpurun_out/${TAG}_ab.txt
echo "== VAE"
for V in "PF_GN_EPILOGUE=1" "PF_GN_EPILOGUE=0"; do env $V python tools/vae_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$V  /"; done | tee gpurun_out/${TAG}_vae.txt
