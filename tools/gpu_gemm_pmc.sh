#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench.txt
PF_GEMM8_MIN_TILES=0 python tools/gemm_bench.py --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench_old.txt
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/gemm_pmc$i -o g -- python $R/tools/gemm_bench.py --reps 2 --shapes conv64,conv16,lin320,ff1_320 > $R/gpurun_out/gemm_pmc$i.log 2>&1
  P=$(find $R/gpurun_out/gemm_pmc$i -name '*counter_collection.csv' | head -1)
  python - "$P" <<'PY' | tee -a $R/gpurun_out/gemm_pmc.txt
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
# dispatch order: each shape runs 3 warm-up + 2 timed launches; report the LAST dispatch of each distinct (kernel, grid)
last = {}
for r in rows:
    if "conv_gemm" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X")))
    last.setdefault(key, {})
    last[key][r["Counter_Name"]] = float(r["Counter_Value"])
for k, c in last.items():
    print(k[0][9:45], "grid", k[1], " ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items())))
PY
  rm -rf $R/gpurun_out/gemm_pmc$i
done
