timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_mixed.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear_ws or model or cfg or width or denoiser" 2>&1 | grep -v amdgpu.ids | tail -n 8 | cut -c1-250
bash tools/gpu_run.sh r4r "env:PF_LINEAR_LN=0"
