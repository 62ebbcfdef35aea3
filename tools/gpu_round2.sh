#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== failing tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py::test_epa_tables_golden_and_flags tests/test_gpu_model.py::test_warpattn_identity_at_init_and_per_sample_cameras -q --tb=short -p no:cacheprovider 2>&1 | tail -n 30 > gpurun_out/pytest_gpu2.log; tail -n 3 gpurun_out/pytest_gpu2.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -n 3 gpurun_out/smoke.log
echo "== bench small"
timeout 300 python bench.py --small --steps 3 --warmup 1 > gpurun_out/bench_small.log 2>&1; tail -n 2 gpurun_out/bench_small.log | cut -c1-600
echo "== bench (graphs)"
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/bench.log 2>&1; tail -n 2 gpurun_out/bench.log | cut -c1-1500
echo "== bench fp16 no graphs"
timeout 600 python bench.py --steps 4 --warmup 1 --dtype fp16 --no-graphs --no-cpu-baseline > gpurun_out/bench_fp16_eager.log 2>&1; tail -n 1 gpurun_out/bench_fp16_eager.log | cut -c1-700
echo "== rocprof"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; tail -n 2 $GRAFT_REPO_ROOT/gpurun_out/rocprof.log | cut -c1-300
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof_r1 | head; find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name "*stats*" | head
# keep only the small summaries
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -type f -size +3M -delete
