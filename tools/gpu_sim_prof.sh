#!/bin/bash
# rocprofv3 kernel stats of one simulated rank (default: world 8, rank 1): where a small rank's step time goes
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-8}; RK=${2:-1}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/simprof -o sim -- python $R/tools/sim_rank.py --world $W --ranks $RK --steps 10 --warmup 2 > $R/gpurun_out/simprof.log 2>&1
grep "^world" $R/gpurun_out/simprof.log
S=$(find $R/gpurun_out/simprof -name '*kernel_stats.csv' | head -1)
cp $S $R/gpurun_out/sim_kernel_stats_w${W}_r${RK}.csv
find $R/gpurun_out/simprof -type f -size +1M -delete
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/sim_kernel_stats_w${W}_r${RK}.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("all kernels: %.1f ms over the run, %d launches" % (tot/1e6, calls))
for r in rows[:16]:
    print("%-70s calls %6s  total %8.2f ms  avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
