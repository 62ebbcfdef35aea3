#!/bin/bash
# Round 6: board power and shader clock WHILE a conv kernel runs on N(0,1) vs zero operands (the power-cap finding of DESIGN.md 3.1b from the
# platform's own sensors).  Samples every card the box exposes through sysfs hwmon every 100 ms during a loop of one kernel and keeps,
# per sample, the card that draws the most power (the job's GPU: the others idle or belong to other jobs).
#   bash tools/power_probe.sh > gpurun_out/r6_power_probe.txt
export TMPDIR=/tmp
sample() {   # -> "<power W> <sclk MHz>" of the busiest card
  best_p=0; best_f=0
  for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
    p=$(cat $h/power1_input 2>/dev/null || cat $h/power1_average 2>/dev/null || echo 0)
    f=$(cat $h/freq1_input 2>/dev/null || echo 0)
    if [ "$p" -gt "$best_p" ]; then best_p=$p; best_f=$f; fi
  done
  echo "$((best_p / 1000000)) $((best_f / 1000000))"
}
for data in randn zeros randn zeros; do
  for k in 1 0; do
    PF_GEMM32=$k python tools/gemm_bench.py --reps 20000 --data $data --shapes conv64_2r > /tmp/pp_${data}_$k.log 2>&1 &
    pid=$!
    sleep 4.0            # import + warm-up
    pw=(); fq=()
    for i in $(seq 1 15); do read p f <<< "$(sample)"; pw+=($p); fq+=($f); sleep 0.1; done
    wait $pid
    res=$(grep conv64_2r /tmp/pp_${data}_$k.log | tail -1)
    echo "operands $data  kernel $([ $k = 1 ] && echo 32x32x16 || echo 16x16x32)  power W: ${pw[*]}"
    echo "                                                   sclk MHz: ${fq[*]}"
    echo "    $res"
  done
done
# the benchmark step itself (graph replay, two streams), 300 steps
python bench.py --no-cpu-baseline --no-training-leg --steps 300 --warmup 5 > /tmp/pp_step.log 2>&1 &
pid=$!
sleep 14.0
pw=(); fq=()
for i in $(seq 1 30); do read p f <<< "$(sample)"; pw+=($p); fq+=($f); sleep 0.1; done
wait $pid
echo "bench.py --steps 300 (cfg 2, fp16 mixed)  power W: ${pw[*]}"
echo "                                          sclk MHz: ${fq[*]}"
tail -1 /tmp/pp_step.log | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('    %.2f steps/s  %.2f ms/step' % (r['value'], r['ms_per_step']))"
