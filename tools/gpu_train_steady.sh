#!/bin/bash
# Steady-state kernel table of ONE training step: rocprofv3 --kernel-trace --stats of tools/train_bench.py at 3 and at 7 timed
# steps; the difference of the two tables / 4 is one step without the run's set-up (weight init, first packing, warm-up).
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; TAG=${1:-r3}; EXTRA=${2:-}
cd /tmp
for n in 3 7; do
  rm -rf /tmp/ts$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts$n -o t -- python $R/tools/train_bench.py --steps $n --no-trace $EXTRA > /tmp/ts$n.log 2>&1
  tail -n 2 /tmp/ts$n.log | cut -c1-200
done
python - $(find /tmp/ts3 -name '*kernel_stats.csv' | head -1) $(find /tmp/ts7 -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${TAG}_train_steady.txt <<'PY'
import csv, re, sys
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0.0))
    if cb - ca > 0:
        rows.append(((tb - ta) / 4e6, (cb - ca) / 4.0, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
native = sum(r[0] for r in rows if "at::" in r[2] or "rocclr" in r[2] or "rocblas" in r[2].lower() or "Cijk" in r[2])
with open(sys.argv[3], "w") as fh:
    fh.write("one training step, steady state (difference of a 7-step and a 3-step run / 4): %.2f ms of kernels, %d launches; torch / runtime / BLAS kernels %.2f ms\n" % (tot, sum(r[1] for r in rows), native))
    for ms, calls, k in rows[:60]:
        k = re.sub(r"\(.*", "", k.replace("void ", "").replace("pf::", ""))[:96]
        fh.write("%-98s %7.1f calls %8.3f ms\n" % (k, calls, ms))
print(open(sys.argv[3]).read()[:6000])
PY
