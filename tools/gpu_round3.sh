#!/bin/bash
# Round-1 iteration 3: new DMA-staged GEMM + LDS-staged attention.  Kernel tests, A/B benches, rocprof.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== kernel tests (defaults: dma + lds)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -n 4 -k "linear or conv or attention" 2>&1 | tail -n 60 | cut -c1-300 > gpurun_out/pytest_k3.log; tail -n 4 gpurun_out/pytest_k3.log
echo "== model tests"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | tail -n 40 | cut -c1-300 > gpurun_out/pytest_m3.log; tail -n 3 gpurun_out/pytest_m3.log
b() { name=$1; shift; echo "== bench $name"; timeout 600 env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$name.log 2>&1; tail -n 1 gpurun_out/bench_$name.log | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    rf = r['roofline']
    print(r['value'], 'steps/s', r['ms_per_step'], 'ms | gemm', round(rf['achieved']), 'TF/s', round(rf['share_of_step_time']*r['ms_per_step'],1), 'ms | attn', {k:(round(v['tflops']), round(v['ms_per_step'],1)) for k,v in rf['other'].items()})
except Exception as e:
    print('bench failed', e)
"; }
b dma_lds A=1
b reg_lds PF_GEMM_STAGING=reg
b dma_direct PF_ATTENTION_IMPL=direct
echo "== bench with cpu baseline (final line)"
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log | cut -c1-2500
echo "== rocprof kernel stats"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs > $R/gpurun_out/rocprof.log 2>&1
find $R/gpurun_out/prof_r1 -type f | head -20
find $R/gpurun_out/prof_r1 -type f -size +2M -delete
