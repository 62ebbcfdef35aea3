timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_mixed.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "groupnorm or model or cfg or width or denoiser or loop" 2>&1 | grep -v amdgpu.ids | tail -n 6 | cut -c1-250
bash tools/gpu_run.sh r4o "env:PF_EPA_LATE_JOIN=0;PF_GN_DIRECT_MAX=0;PF_EPA_LATE_JOIN=0 PF_GN_DIRECT_MAX=0"
