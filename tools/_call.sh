timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | tail -n 12 | cut -c1-300
bash tools/gpu_run.sh r4t smoke
