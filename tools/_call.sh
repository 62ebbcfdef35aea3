timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear_ws" 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-250
python tools/lws_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4m_lws_bench.txt
