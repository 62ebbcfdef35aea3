timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_mixed.py tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -x 2>&1 | grep -v amdgpu.ids | tail -n 12 | cut -c1-250
bash tools/gpu_run.sh r4h "env:PF_LINEAR_WS=0" "bench:--no-cpu-baseline --no-training-leg"
