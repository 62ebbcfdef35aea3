#!/bin/bash
# MFMA-pipe utilisation of every MFMA kernel INSIDE the benchmark step ("measured in-step"), by hardware
# counters over one eager, single-stream denoiser step (every dispatch of the step, not one shape per process):
#   util(kernel) = sum SQ_VALU_MFMA_BUSY_CYCLES / sum ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs)   over the kernel's dispatches
# Two separate --pmc passes with --kernel-trace only (SQ counters; GRBM_GUI_ACTIVE), joined per dispatch in order.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
cd /tmp
rm -rf /tmp/mi1 /tmp/mi2
PF_STREAMS=1 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/mi1 -o g -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-training-leg --no-graphs > $R/gpurun_out/${TAG}_mfma_instep_1.log 2>&1
PF_STREAMS=1 timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mi2 -o g -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-training-leg --no-graphs > $R/gpurun_out/${TAG}_mfma_instep_2.log 2>&1
python - $(find /tmp/mi1 -name '*counter_collection.csv' | head -1) $(find /tmp/mi2 -name '*counter_collection.csv' | head -1) $R/gpurun_out/${TAG}_mfma_instep.txt $R <<'PY'
import csv, re, sys
from collections import defaultdict
def short(n):
    n = n.replace("void ", "").replace("pf::", "")
    return re.sub(r"\(.*", "", n)[:60]
def load(path):
    d, order = {}, []
    for r in csv.DictReader(open(path)):
        k = int(r["Dispatch_Id"])
        if k not in d:
            order.append(k)
            d[k] = {"name": short(r["Kernel_Name"]), "grid": r.get("Grid_Size", "")}
        d[k][r["Counter_Name"]] = float(r["Counter_Value"])
    return [d[k] for k in order]
a, b = load(sys.argv[1]), load(sys.argv[2])
# the two runs issue the same dispatch sequence; keep the LAST denoiser pass (the instrumented eager step) = tail of both
MF = ("k_conv_gemm", "k_linear_ws", "k_attention")
mf = [x for x in a if x["name"].startswith(MF)]
gb = [x for x in b if x["name"].startswith(MF)]
n = min(len(mf), len(gb))
mf, gb = mf[-n:], gb[-n:]
assert all(x["name"] == y["name"] for x, y in zip(mf, gb)), "dispatch sequences differ"
agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for x, y in zip(mf, gb):
    fam = "attention D=64 (self / text)" if "k_attention_lds<F16, 64" in x["name"] else "attention D=32 (EPA)" if "k_attention" in x["name"] else x["name"]
    t = agg[fam]
    t[0] += x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    t[1] += y.get("GRBM_GUI_ACTIVE", 0.0) / 8 * 1024
    t[2] += 4 * x.get("SQ_ACTIVE_INST_VALU", 0.0)
    t[3] += 1
sys.path.insert(0, sys.argv[4])
from panfusion_amd import _lib
with open(sys.argv[3], "w") as fh:
    fh.write("csrc_sha256: %s\n" % _lib.source_hash())
    fh.write("MFMA-pipe utilisation inside the step (all dispatches of the last denoiser passes of `bench.py --steps 1 --no-graphs`, PF_STREAMS=1)\n")
    fh.write("%-44s %9s %12s %12s\n" % ("kernel", "launches", "MFMA busy %", "VALU busy %"))
    tot_a, tot_g = [0.0, 0.0], [0.0, 0.0]
    for k, (m, c, v, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fh.write("%-44s %9d %12.1f %12.1f\n" % (k, cnt, 100 * m / c, 100 * v / c))
        tot = tot_a if k.startswith("attention") else tot_g
        tot[0] += m
        tot[1] += c
    if tot_a[1]:
        fh.write("attention, all launches, cycle-weighted: MFMA busy %.1f %%\n" % (100 * tot_a[0] / tot_a[1]))
    if tot_g[1]:
        fh.write("GEMM family (tile kernels + weight-stationary linear), all launches, cycle-weighted: MFMA busy %.1f %%\n" % (100 * tot_g[0] / tot_g[1]))
print(open(sys.argv[3]).read())
PY
