timeout 1500 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "split or exact or mixed or stream or precision" 2>&1 | grep -v amdgpu.ids | tail -n 4 | cut -c1-300
bash tools/gpu_run.sh r4v "bench:--no-cpu-baseline --no-training-leg"
grep -E "M163840 N320 K640 |M40960 N640 K1280 |M10240 N1280 K2560 |M163840 N320 K1920|M40960 N640 K3840|M40960 N320 K5760|M10240 N640 K11520" gpurun_out/r4v_shapes___no_cpu_baseline___no_training_leg.txt
