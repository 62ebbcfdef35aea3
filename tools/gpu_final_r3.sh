#!/bin/bash
# Round-end evidence (round 3): full gpu suite, default bench (with cpu baseline), rocprofv3 kernel stats of the SAME
# default command, serial per-kernel table, PMC traffic passes of the dominant kernel family, secondary bench lines.
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_final}
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 2>&1 | grep -v amdgpu.ids | tail -n 30 | cut -c1-300 > gpurun_out/${TAG}_pytest_gpu.log; tail -n 3 gpurun_out/${TAG}_pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2 | tee gpurun_out/${TAG}_smoke.log
echo "== bench (default = fp16 mixed)"
timeout 900 python bench.py --trace-out gpurun_out/${TAG}_shapes.txt > gpurun_out/${TAG}_bench.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
echo "== bench secondary lines (all-16-bit fp16 / bf16)"
timeout 600 python bench.py --precision fast --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench_fp16_fast.json; cut -c1-200 gpurun_out/${TAG}_bench_fp16_fast.json
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench_bf16_fast.json; cut -c1-200 gpurun_out/${TAG}_bench_bf16_fast.json
cd /tmp
echo "== rocprofv3 --kernel-trace --stats of the default bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_g -o bench -- python $R/bench.py --no-cpu-baseline --no-training-leg > $R/gpurun_out/${TAG}_rocprof_graphs.log 2>&1
cp $(find $R/gpurun_out/${TAG}_g -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_kernel_stats_default_cmd.csv 2>/dev/null
find $R/gpurun_out/${TAG}_g -type f -size +1M -delete
head -n 8 $R/gpurun_out/${TAG}_kernel_stats_default_cmd.csv | cut -c1-160
echo "== serial (one stream, no graphs) per-kernel table"
PF_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_s -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-training-leg --no-graphs > $R/gpurun_out/${TAG}_rocprof_serial.log 2>&1
T=$(find $R/gpurun_out/${TAG}_s -name '*kernel_trace.csv' | head -1)
python $R/tools/prof_summary.py trace $T $R/gpurun_out/${TAG}_kernels_serial.txt 10
find $R/gpurun_out/${TAG}_s -type f -size +1M -delete
head -n 14 $R/gpurun_out/${TAG}_kernels_serial.txt
echo "== PMC traffic passes (separate passes, --kernel-trace only)"
for C in FETCH_SIZE WRITE_SIZE; do
  PF_STREAMS=1 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$C -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-training-leg --no-graphs > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
  P=$(find $R/gpurun_out/${TAG}_pmc_$C -name '*counter_collection.csv' | head -1)
  python $R/tools/prof_summary.py pmc $P $R/gpurun_out/${TAG}_pmc_$C.txt
  find $R/gpurun_out/${TAG}_pmc_$C -type f -size +1M -delete
  grep -E "k_conv_gemm|k_attention" $R/gpurun_out/${TAG}_pmc_$C.txt | cut -c1-200
done
python $R/tools/prof_summary.py traffic $R/gpurun_out/${TAG}_pmc_FETCH_SIZE.txt $R/gpurun_out/${TAG}_pmc_WRITE_SIZE.txt $R/gpurun_out/${TAG}_traffic.json
echo "== eager two-stream critical path (tools/trace_streams.py)"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_e -o t -- python $R/bench.py --no-graphs --steps 3 --warmup 1 --no-cpu-baseline --no-training-leg > $R/gpurun_out/${TAG}_rocprof_eager.log 2>&1
python $R/tools/trace_streams.py $(find $R/gpurun_out/${TAG}_e -name '*kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_streams.txt
rm -rf $R/gpurun_out/${TAG}_e
tail -n 9 $R/gpurun_out/${TAG}_streams.txt
echo "== cfg 4 (1024x2048 panorama): bench line + PMC traffic per kernel"
timeout 600 python $R/bench.py --cfg4 --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 > $R/gpurun_out/${TAG}_bench_cfg4.json; cut -c1-200 $R/gpurun_out/${TAG}_bench_cfg4.json
timeout 600 python $R/bench.py --cfg5 --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 > $R/gpurun_out/${TAG}_bench_cfg5.json; cut -c1-200 $R/gpurun_out/${TAG}_bench_cfg5.json
for C in FETCH_SIZE WRITE_SIZE; do
  PF_STREAMS=1 timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc4_$C -o bench -- python $R/bench.py --cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-training-leg --no-graphs > $R/gpurun_out/${TAG}_pmc4_$C.log 2>&1
  P=$(find $R/gpurun_out/${TAG}_pmc4_$C -name '*counter_collection.csv' | head -1)
  python $R/tools/prof_summary.py pmc $P $R/gpurun_out/${TAG}_cfg4_pmc_$C.txt
  find $R/gpurun_out/${TAG}_pmc4_$C -type f -size +1M -delete
done
PF_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_s4 -o bench -- python $R/bench.py --cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-training-leg --no-graphs > $R/gpurun_out/${TAG}_rocprof_serial_cfg4.log 2>&1
python $R/tools/prof_summary.py trace $(find $R/gpurun_out/${TAG}_s4 -name '*kernel_trace.csv' | head -1) $R/gpurun_out/${TAG}_kernels_serial_cfg4.txt 6
find $R/gpurun_out/${TAG}_s4 -type f -size +1M -delete
head -n 12 $R/gpurun_out/${TAG}_kernels_serial_cfg4.txt
cd $R
echo "== VAE / microbenchmarks"
python tools/vae_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_vae_bench.txt
python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_gemm_microbench.txt
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_attn_microbench.txt
python tools/elem_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_elem_microbench.txt
tail -n 3 gpurun_out/${TAG}_gemm_microbench.txt
echo "== training step: LoRA + EPA, and layout-conditioned (the ControlNet trains)"
timeout 400 python tools/train_bench.py --steps 3 2>&1 | grep -v amdgpu.ids | tail -n 6 > gpurun_out/${TAG}_train_lora.txt; head -n 1 gpurun_out/${TAG}_train_lora.txt | cut -c1-300
timeout 400 python tools/train_bench.py --layout-cond --steps 3 2>&1 | grep -v amdgpu.ids | tail -n 6 > gpurun_out/${TAG}_train_layout_cond.txt; head -n 1 gpurun_out/${TAG}_train_layout_cond.txt | cut -c1-300
echo "== MFMA utilisation in the step (PMC)"
bash tools/gpu_mfma_instep.sh ${TAG} 2>&1 | tail -n 14
