#!/bin/bash
# Is the training step host-bound?  Kernel trace of tools/train_bench.py on ONE stream: GPU busy time / wall time over the timed steps.
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT; TAG=${1:-r3}
cd /tmp; rm -rf /tmp/tb
PF_TRAIN_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -o t -- python $R/tools/train_bench.py --steps 6 --no-trace > /tmp/tb.log 2>&1
tail -n 1 /tmp/tb.log | cut -c100-200
python - $(find /tmp/tb -name '*kernel_trace.csv' | head -1) <<'PY' | tee $R/gpurun_out/${TAG}_train_busy.txt
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the optimizer's multi-tensor kernels mark step ends: take windows between consecutive AdamW bursts
marks = [s for s, e, n in rows if "multi_tensor_apply" in n or "FusedAdam" in n or "adam" in n.lower()]
ends = []
for m in marks:
    if not ends or m - ends[-1] > 20e6:
        ends.append(m)
print("optimizer bursts:", len(ends))
for a, b in list(zip(ends[:-1], ends[1:]))[-5:]:
    busy, last = 0, a
    n = 0
    for s, e, _ in rows:
        if e <= a or s >= b:
            continue
        s2 = max(s, last)
        if e > s2:
            busy += e - s2
            last = e
        n += 1
    print("window %.1f ms: GPU busy %.1f ms (%.0f %%), %d kernels" % ((b - a) / 1e6, busy / 1e6, 100.0 * busy / (b - a), n))
PY
