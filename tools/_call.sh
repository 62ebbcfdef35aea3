timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "conv_in or boundary or model" 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-250
bash tools/gpu_run.sh r4q serial > /dev/null 2>&1
grep -n "conv_in\|conv_out" gpurun_out/r4q_kernels_serial.txt | head
