#!/bin/bash
# Round 3, first pass: parity at the exact BASELINE configs (tests/test_gpu_configs.py), the default bench line of the
# round's starting point, and a kernel trace of the GRAPH-REPLAYED step (per-stream, for the critical-path analysis).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3a}
echo "== parity at the exact configs"
timeout 1200 python -m pytest tests/test_gpu_configs.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | cut -c1-400 > gpurun_out/${TAG}_configs_parity.log; tail -n 40 gpurun_out/${TAG}_configs_parity.log
echo "== bench (default = fp16 mixed)"
timeout 900 python bench.py --trace-out gpurun_out/${TAG}_shapes.txt > gpurun_out/${TAG}_bench.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
cd /tmp
echo "== rocprofv3 --kernel-trace of graph replay (3 steps)"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_g -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-training-leg > $R/gpurun_out/${TAG}_rocprof_graphs.log 2>&1
T=$(find $R/gpurun_out/${TAG}_g -name '*kernel_trace.csv' | head -1)
ls -la $T
# keep the last ~12000 rows (the timed steps + the instrumented eager step), columns that matter
python - "$T" "$R/gpurun_out/${TAG}_trace_tail.csv" <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-14000:]
cols = [c for c in ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if c in rows[0]]
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(cols)
for r in keep:
    w.writerow([r[c][:90] if c == "Kernel_Name" else r[c] for c in cols])
print(len(rows), "rows,", len(keep), "kept; columns", list(rows[0].keys()))
EOF
rm -rf $R/gpurun_out/${TAG}_g
cd $R
ls -la gpurun_out | tail -n 8
