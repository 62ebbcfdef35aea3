#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_bench.txt
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc$i -o g -- python $R/tools/attn_bench.py --reps 1 --shapes self64,epa_e > $R/gpurun_out/attn_pmc$i.log 2>&1
  P=$(find $R/gpurun_out/attn_pmc$i -name '*counter_collection.csv' | head -1)
  python - "$P" <<'PY' | tee -a $R/gpurun_out/attn_pmc.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last = {}
for r in rows:
    if "attention" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X")))
    last.setdefault(key, {})
    last[key][r["Counter_Name"]] = float(r["Counter_Value"])
for k, c in last.items():
    print(k[0][9:45], "grid", k[1], " ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items())))
PY
  rm -rf $R/gpurun_out/attn_pmc$i
done
