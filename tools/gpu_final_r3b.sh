#!/bin/bash
# Round-end re-check after the training-path work that followed tools/gpu_final_r3.sh (inference kernels unchanged since):
# full gpu suite, smoke, the default bench line (with cpu baseline + training leg), the two training benches.
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
TAG=${1:-r3_final}
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | tail -n 30 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu.log; tail -n 3 gpurun_out/${TAG}_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee gpurun_out/${TAG}_smoke.log
echo "== bench (default = fp16 mixed)"
timeout 900 python bench.py --trace-out gpurun_out/${TAG}_shapes.txt > gpurun_out/${TAG}_bench.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
echo "== training step: LoRA + EPA, and layout-conditioned (the ControlNet trains)"
timeout 400 python tools/train_bench.py --steps 4 2>&1 | grep -v amdgpu.ids | tail -n 6 > gpurun_out/${TAG}_train_lora.txt; head -n 1 gpurun_out/${TAG}_train_lora.txt | cut -c1-300
timeout 400 python tools/train_bench.py --layout-cond --steps 4 2>&1 | grep -v amdgpu.ids | tail -n 6 > gpurun_out/${TAG}_train_layout_cond.txt; head -n 1 gpurun_out/${TAG}_train_layout_cond.txt | cut -c1-300
