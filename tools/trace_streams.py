"""Critical-path view of an EAGER two-stream step from a rocprofv3 kernel trace (streams are preserved only without hipGraphs):

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --no-graphs --steps 3 --warmup 1 --no-cpu-baseline --no-training-leg
    python tools/trace_streams.py out/.../t_kernel_trace.csv [summary.txt]

Takes the LAST complete two-stream pass, cuts it at the EPA attention kernels (D = 32, biased: one per direction and block) and
prints, per segment between two fusions, the kernel-busy time of the view stream and of the panorama (side) stream, the time
either stream sat idle inside the segment, and the overlap -- i.e. which branch the joins wait for."""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("void ", "").replace("pf::", "")
    return re.sub(r"\(.*", "", n)[:46]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    rows = [r for r in rows if "spin_kernel" not in r["Kernel_Name"]]
    # passes: split at gaps > 300 us with no kernel running
    passes, cur, end = [], [rows[0]], rows[0]["e"]
    for r in rows[1:]:
        if r["s"] - end > 300000:
            passes.append(cur)
            cur = []
        cur.append(r)
        end = max(end, r["e"])
    passes.append(cur)
    two = [p for p in passes if len(set(r["Stream_Id"] for r in p)) >= 2 and len(p) > 1000]
    if not two:
        print("no two-stream pass found", file=out)
        return
    p = two[-1]
    t0, t1 = p[0]["s"], max(r["e"] for r in p)
    by = collections.defaultdict(list)
    for r in p:
        by[r["Stream_Id"]].append(r)
    main_id = max(by, key=lambda k: sum(r["e"] - r["s"] for r in by[k]))
    print("pass: %d kernels, wall %.2f ms; streams %s (view stream = %s)" % (len(p), (t1 - t0) / 1e6, {k: len(v) for k, v in by.items()}, main_id), file=out)
    for sid, ks in by.items():
        busy = sum(r["e"] - r["s"] for r in ks) / 1e6
        fam = collections.defaultdict(lambda: [0, 0.0])
        for r in ks:
            f = fam[short(r["Kernel_Name"])]
            f[0] += 1
            f[1] += (r["e"] - r["s"]) / 1e6
        print("stream %s: %d kernels, busy %.2f ms" % (sid, len(ks), busy), file=out)
        for n, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:12]:
            print("    %-48s %4d %8.3f" % (n, v[0], v[1]), file=out)
    # segments on the view stream: cut after each EPA attention
    cuts = [r["e"] for r in by[main_id] if "k_attention_lds<pf::F16, 32" in r["Kernel_Name"] or "k_attention_lds<F16, 32" in short(r["Kernel_Name"])]
    bounds = [t0] + cuts + [t1]
    print("\nsegment  start..end ms   view busy  pano busy  view idle  both-running", file=out)
    segs = []
    for a, b in zip(bounds, bounds[1:]):
        def busy_in(ks):
            return sum(max(0, min(r["e"], b) - max(r["s"], a)) for r in ks) / 1e6
        vb = busy_in(by[main_id])
        pb = sum(busy_in(v) for k, v in by.items() if k != main_id)
        ev = []
        for r in p:
            s, e = max(r["s"], a), min(r["e"], b)
            if e > s:
                ev += [(s, 1), (e, -1)]
        ev.sort()
        n, last, both = 0, a, 0
        for t, d in ev:
            if n >= 2:
                both += t - last
            n += d
            last = t
        print("  %6.2f .. %6.2f   %8.2f   %8.2f   %8.2f   %8.2f" % ((a - t0) / 1e6, (b - t0) / 1e6, vb, pb, (b - a) / 1e6 - vb, both / 1e6), file=out)
        segs.append((a, b, (b - a) / 1e6 - vb))
    # the segments where the view stream waits: what the panorama stream is doing there (kernel families, time = end - start
    # of each kernel, i.e. including the wait for compute units)
    for a, b, idle in segs:
        if idle < 0.4 or len(segs) - segs.index((a, b, idle)) > 9:      # last step only
            continue
        fam = collections.defaultdict(lambda: [0, 0.0])
        for k, v in by.items():
            if k == main_id:
                continue
            for r in v:
                if r["s"] >= a and r["e"] <= b:
                    f = fam[short(r["Kernel_Name"]) + " grid " + r.get("Grid_Size_X", "")]
                    f[0] += 1
                    f[1] += (r["e"] - r["s"]) / 1e3
        print("\nsegment %.2f .. %.2f ms (view idle %.2f ms): panorama-stream kernels" % ((a - t0) / 1e6, (b - t0) / 1e6, idle), file=out)
        for n, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
            print("    %-70s %3d  %8.1f us" % (n, v[0], v[1]), file=out)


if __name__ == "__main__":
    main()
