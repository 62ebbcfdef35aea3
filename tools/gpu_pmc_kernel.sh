#!/bin/bash
# SQ counters of every dispatch of one kernel family under a microbenchmark command:
#   gpu_pmc_kernel.sh <tag> <kernel substring> <command ...>
# pass 1: MFMA / VALU / LDS / wait counters (8 SQ slots), pass 2: GRBM_GUI_ACTIVE (cycles); joined per dispatch in order.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; SUB=$2; shift 2
cd /tmp; rm -rf /tmp/pk1 /tmp/pk2
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pk1 -o g -- "$@" > /tmp/pk1.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pk2 -o g -- "$@" > /tmp/pk2.log 2>&1
python - "$SUB" $(find /tmp/pk1 -name '*counter_collection.csv' | head -1) $(find /tmp/pk2 -name '*counter_collection.csv' | head -1) $(find /tmp/pk2 -name '*kernel_trace.csv' | head -1) <<'PY' | tee $R/gpurun_out/${TAG}_pmc_kernel.txt
import csv, re, sys
sub = sys.argv[1]
def load(path):
    d, order = {}, []
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]: continue
        k = int(r["Dispatch_Id"])
        if k not in d:
            order.append(k); d[k] = {"name": re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("pf::", ""))[:48]}
        d[k][r["Counter_Name"]] = float(r["Counter_Value"])
    return [d[k] for k in order]
a, b = load(sys.argv[2]), load(sys.argv[3])
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(sys.argv[4])) if sub in r["Kernel_Name"]]
n = min(len(a), len(b), len(dur))
print("%-48s %8s %6s %7s %7s %9s %9s %8s %9s" % ("kernel", "us", "GHz", "MFMA %", "VALU %", "wait_any%", "wait_inst%", "LDS act%", "LDS confl%"))
for x, y, t in zip(a[-n:], b[-n:], dur[-n:]):
    cyc = y["GRBM_GUI_ACTIVE"] / 8
    wc = x["SQ_WAVE_CYCLES"]
    print("%-48s %8.1f %6.2f %7.1f %7.1f %9.1f %9.1f %8.1f %9.1f" % (
        x["name"], t, cyc / t * 1e-3, 100 * x["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 100 * 4 * x["SQ_ACTIVE_INST_VALU"] / (cyc * 1024),
        100 * x["SQ_WAIT_ANY"] / wc, 100 * x["SQ_WAIT_INST_ANY"] / wc, 100 * 4 * x["SQ_LDS_IDX_ACTIVE"] / (cyc * 256) if cyc else 0,
        100 * x["SQ_LDS_BANK_CONFLICT"] / max(x["SQ_LDS_IDX_ACTIVE"], 1)))
PY
