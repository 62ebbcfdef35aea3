#!/bin/bash
# MFMA-pipe utilisation of the two MFMA kernels by hardware counters, one shape per process (clean attribution):
#   util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs)
# SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD's matrix pipe (= 32 x #MFMA for 32x32x16, 16 x for 16x16x32:
# checked against the FLOP count); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (checked: SQ_WAVE_CYCLES x 4 /
# (GRBM_GUI_ACTIVE / 8) = the resident wave count).  Separate --pmc passes, --kernel-trace only.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r2}_mfma_util.txt
: > $OUT
cd /tmp
one() {   # $1 = tool, $2 = shape, $3 = kernel substring
  rm -rf /tmp/mu1 /tmp/mu2
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/mu1 -o g -- python $R/tools/$1 --reps 1 --shapes $2 > /tmp/mu1.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mu2 -o g -- python $R/tools/$1 --reps 1 --shapes $2 > /tmp/mu2.log 2>&1
  python - "$2" "$3" $(find /tmp/mu1 -name '*counter_collection.csv' | head -1) $(find /tmp/mu2 -name '*counter_collection.csv' | head -1) $(find /tmp/mu2 -name '*kernel_trace.csv' | head -1) <<'PY' | tee -a $OUT
import csv, sys
shape, sub = sys.argv[1], sys.argv[2]
def last(path):
    d, order = {}, []
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]: continue
        k = r["Dispatch_Id"]
        if k not in d: order.append(k)
        d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return d[order[-1]]
a, b = last(sys.argv[3]), last(sys.argv[4])
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(sys.argv[5])) if sub in r["Kernel_Name"]][-1]
cyc = b["GRBM_GUI_ACTIVE"] / 8
print("%-10s %-12s %8.1f us  clock %.2f GHz  MFMA busy %5.1f %%  VALU busy %5.1f %%  resident waves/SIMD %.2f"
      % (shape, sub, dur, cyc / dur * 1e-3, 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
         100 * 4 * a["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4 * a["SQ_WAVE_CYCLES"] / cyc / 1024))
PY
}
for s in self64 self32 pano64 text64 epa_e epa_p; do one attn_bench.py $s k_attention; done
for s in conv64 conv64cat conv32 conv16 conv8 lin320 qk320 ff1_320 ff2_320 lin640 ff1_640 lin1280 ff1_1280; do one gemm_bench.py $s k_conv_gemm; done
cat $OUT > /dev/null
