#!/bin/bash
# L2 <-> fabric bytes (FETCH_SIZE, WRITE_SIZE: separate --pmc passes) of every dispatch of one kernel family under a microbenchmark command,
# in dispatch order (the command's shapes in order, 3 warm-up launches + --reps each):
#   gpu_fetch_kernel.sh <tag> <kernel substring> <command ...>
# FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide reads at 64 B: MI355X_MICROARCH.md, HBM); Infinity-Cache hits are counted.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; SUB=$2; shift 2
cd /tmp; rm -rf /tmp/fk1 /tmp/fk2
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fk1 -o g -- "$@" > /tmp/fk1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/fk2 -o g -- "$@" > /tmp/fk2.log 2>&1
grep -E "TF/s" /tmp/fk1.log | sed 's/^/  bench: /' | tee $R/gpurun_out/${TAG}_fetch_kernel.txt
python - "$SUB" $(find /tmp/fk1 -name '*counter_collection.csv' | head -1) $(find /tmp/fk2 -name '*counter_collection.csv' | head -1) <<'PY' | tee -a $R/gpurun_out/${TAG}_fetch_kernel.txt
import csv, re, sys
sub = sys.argv[1]
def load(path, name):
    out = []
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"] and r["Counter_Name"] == name:
            out.append((re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("pf::", ""))[:52], r.get("Grid_Size", r.get("Grid_Size_X", "")), float(r["Counter_Value"])))
    return out
f, w = load(sys.argv[2], "FETCH_SIZE"), load(sys.argv[3], "WRITE_SIZE")
print("%-52s %10s %14s %14s" % ("kernel (dispatch order)", "grid", "fetch MB (x2)", "write MB"))
prev = None
for (n, g, fv), (_, _, wv) in zip(f, w):
    key = (n, g, round(fv / 1e3), round(wv / 1e3))
    if key == prev:
        continue
    prev = key
    print("%-52s %10s %14.1f %14.1f" % (n, g, 2 * fv * 1024 / 1e6, wv * 1024 / 1e6))
PY
