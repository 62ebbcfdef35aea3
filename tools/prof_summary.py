"""Condense rocprofv3 CSV output (too large to keep) into small per-kernel tables.

    python tools/prof_summary.py trace  <kernel_trace.csv>  <out.txt>  [denoiser passes in the run]
    python tools/prof_summary.py trace_steady <kernel_trace.csv> <out.txt> <steps> <warmup>     (the timed steps only)
    python tools/prof_summary.py pmc    <counter_collection.csv> <out.txt>
"""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def csrc_hash():
    from panfusion_amd import _lib
    return _lib.source_hash()


def short(name):
    name = name.replace("void ", "").replace("pf::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:70]


SETUP = ("spin_kernel", "distribution_elementwise", "MulFunctor", "FillFunctor")   # bench set-up: stream hold, weight init


def trace(path, out, passes):
    agg = defaultdict(lambda: [0, 0.0])
    setup = 0.0
    for r in csv.DictReader(open(path)):
        if any(s in r["Kernel_Name"] for s in SETUP):
            setup += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            continue
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    tot = sum(v[1] for v in agg.values())
    fam = defaultdict(float)
    for k, v in agg.items():
        fam[k[0]] += v[1]
    with open(out, "w") as fh:
        fh.write("total kernel time %.2f ms over %d denoiser passes = %.2f ms / pass"
                 "   (excluded: %.1f ms of run set-up kernels -- weight init, the bench's stream-hold spin kernel)\n\n"
                 % (tot, passes, tot / passes, setup))
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
            fh.write("%-72s %9.3f ms/pass %5.1f %%\n" % (k, v / passes, 100 * v / tot))
        fh.write("\nper (kernel, grid): calls/pass, ms/pass, us/call\n")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
            fh.write("%-60s grid %8s %4s %4s  calls %6.1f  %8.3f ms  %8.1f us\n" % (k[0][:60], k[1], k[2], k[3], v[0] / passes, v[1] / passes, 1e3 * v[1] / v[0]))


def trace_steady(path, out, steps, warmup):
    """Per-kernel table of the TIMED steps only (the whole-run table mixes in weight packing, table building
    and warm-up).  The loop ends every step with two k_cfg_ddim_rows launches (views, panorama): the window runs from the end of
    the last warm-up step's second one to the end of the last timed step's -- exactly `steps` steps, nothing else."""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ddim = [i for i, r in enumerate(rows) if "k_cfg_ddim" in r["Kernel_Name"]]
    need = 2 * (warmup + steps)
    if len(ddim) < need:
        raise SystemExit("trace_steady: %d DDIM launches in the trace, %d expected" % (len(ddim), need))
    lo = ddim[2 * warmup - 1] + 1 if warmup else 0
    hi = ddim[need - 1] + 1
    if not warmup:      # no warm-up step: start at the first kernel after the set-up (the first step's first launch is unknown): refuse
        raise SystemExit("trace_steady needs at least one warm-up step")
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    tot = sum(v[1] for v in agg.values())
    n = sum(v[0] for v in agg.values())
    fam = defaultdict(lambda: [0, 0.0])
    for k, v in agg.items():
        fam[k[0]][0] += v[0]
        fam[k[0]][1] += v[1]
    foreign = {k: v for k, v in fam.items() if not k.startswith("k_")}
    mfma = ("k_conv_gemm", "k_linear_ws", "k_attention")
    tail = sum(v[1] for k, v in fam.items() if not k.startswith(mfma))
    tail_n = sum(v[0] for k, v in fam.items() if not k.startswith(mfma))
    with open(out, "w") as fh:
        fh.write("csrc_sha256: %s\n" % csrc_hash())
        fh.write("steady-state window: %d timed steps (one stream, no graphs) = %.2f ms of kernels and %.0f launches per step\n" % (steps, tot / steps, n / steps))
        fh.write("non-MFMA tail (everything but k_conv_gemm* / k_linear_ws / k_attention*): %.2f ms and %.0f launches per step\n" % (tail / steps, tail_n / steps))
        fh.write("kernels that are not this library's (torch / runtime copies, fills, casts) inside the window: %s\n\n"
                 % (", ".join("%s x%.1f %.3f ms" % (k, v[0] / steps, v[1] / steps) for k, v in sorted(foreign.items())) or "NONE"))
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            fh.write("%-72s %7.1f launches %9.3f ms/step %5.1f %%\n" % (k, v[0] / steps, v[1] / steps, 100 * v[1] / tot))
        fh.write("\nper (kernel, grid): calls/step, ms/step, us/call\n")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
            fh.write("%-60s grid %8s %4s %4s  calls %6.1f  %8.3f ms  %8.1f us\n" % (k[0][:60], k[1], k[2], k[3], v[0] / steps, v[1] / steps, 1e3 * v[1] / v[0]))


def pmc(path, out):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
    with open(out, "w") as fh:
        for k, cs in sorted(agg.items()):
            for c, v in sorted(cs.items()):
                fh.write("%-72s %-28s total %.6g  per-dispatch %.6g  dispatches %d\n" % (k, c, v, v / calls[(k, c)], calls[(k, c)]))


def traffic(fetch_txt, write_txt, out, family=("k_conv_gemm", "k_linear_ws")):
    """profiles/r<N>_traffic.json (read by bench.py for roofline.traffic) from the two PMC summaries: the GEMM family
    (tile kernels + the weight-stationary linear kernel), and every kernel of the run together (`all_kernels`)."""
    import json
    tot, n, everything, tail, tail_n = {}, {}, {}, {}, {}
    mfma = family + ("k_attention",)
    for path in (fetch_txt, write_txt):
        for line in open(path):
            f = line.split()
            if "total" not in f:
                continue
            c = f[f.index("total") - 1]
            everything[c] = everything.get(c, 0.0) + float(f[f.index("total") + 1])
            if line.startswith("k_") and not line.startswith(mfma):        # this library's non-MFMA kernels: the tail
                tail[c] = tail.get(c, 0.0) + float(f[f.index("total") + 1])
                tail_n[c] = tail_n.get(c, 0) + int(f[f.index("dispatches") + 1])
            if not line.startswith(family):
                continue
            tot[c] = tot.get(c, 0.0) + float(f[f.index("total") + 1])
            n[c] = n.get(c, 0) + int(f[f.index("dispatches") + 1])
    assert n["FETCH_SIZE"] == n["WRITE_SIZE"], n
    doc = {
        "csrc_sha256": csrc_hash(),
        "tail_launches": tail_n.get("FETCH_SIZE", 0),
        "tail_bytes_per_launch": (2.0 * tail.get("FETCH_SIZE", 0.0) + tail.get("WRITE_SIZE", 0.0)) * 1024.0 / max(tail_n.get("FETCH_SIZE", 0), 1),
        "kernel": "k_conv_gemm (all tile variants) + k_linear_ws",
        "all_kernels_bytes_total": (2.0 * everything["FETCH_SIZE"] + everything["WRITE_SIZE"]) * 1024.0,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, PF_STREAMS=1 bench.py --steps 1 --warmup 0 --no-graphs",
        "launches": n["FETCH_SIZE"],
        "fetch_size_kb_total": tot["FETCH_SIZE"],
        "write_size_kb_total": tot["WRITE_SIZE"],
        "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); "
                      "WRITE_SIZE uncorrected; Infinity-Cache hits are counted, so this is L2<->fabric traffic, an upper bound on HBM bytes",
        "bytes_per_launch": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / n["FETCH_SIZE"],
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(doc["launches"], "launches,", round(doc["bytes_per_launch"] / 1e6, 1), "MB per launch")


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "trace_steady":
        trace_steady(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    elif sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    else:
        pmc(sys.argv[2], sys.argv[3])
