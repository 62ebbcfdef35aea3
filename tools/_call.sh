bash tools/gpu_run.sh r4i tests "bench:--no-cpu-baseline --no-training-leg"
head -n 12 gpurun_out/r4i_shapes___no_cpu_baseline___no_training_leg.txt
echo "== sim ranks"
{ python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('world 1  cfg2 %.2f ms/step' % r['ms_per_step'])"
  python tools/sim_rank.py --world 2 --ranks 0 2>&1 | grep "^world"
  python tools/sim_rank.py --world 4 --ranks 0,1 2>&1 | grep "^world"
  python tools/sim_rank.py --world 8 --ranks 0,1,3 2>&1 | grep "^world"
  python tools/sim_rank.py --world 8 --ranks 1 --split 1,7,6,6 2>&1 | grep "^world"
  python tools/sim_rank.py --world 4 --ranks 1 --split 2,18 2>&1 | grep "^world"
  python bench.py --cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-training-leg 2>&1 | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('world 1  cfg4 %.2f ms/step' % r['ms_per_step'])"
  python tools/sim_rank.py --cfg4 --world 8 --ranks 0,1 --split 0,7,7,6 2>&1 | grep "^world"
  python tools/sim_rank.py --cfg4 --world 4 --ranks 0 --split 6,14 2>&1 | grep "^world"
  python tools/sim_rank.py --cfg4 --world 2 --ranks 0 2>&1 | grep "^world"
  python tools/sim_rank.py --cfg5 --world 8 --ranks 0 2>&1 | grep "^world"; } | tee gpurun_out/r4i_sim_ranks.txt
NS="4" ARGS="--cfg5" bash tools/gpu_dist_dry.sh 2>&1 | tail -3
