"""Idle time of the GPU inside the timed steps of bench.py (graph replay, two streams): from a rocprofv3 kernel trace,
the union of all kernel intervals vs the span they cover, and per-stream (queue) busy time.
    python tools/gap_analysis.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "0"))))
    rows.sort()
    # the timed region: the longest stretch of k_cfg_ddim kernels (2 per step) -- take the span of the last 16 of them
    ddim = [i for i, r in enumerate(rows) if "k_cfg_ddim" in r[2]]
    if len(ddim) < 18:
        print("not enough steps in the trace")
        return
    i0, i1 = ddim[-17], ddim[-1]                      # 8 whole steps between them
    t0, t1 = rows[i0][1], rows[i1][1]
    sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    span = (t1 - t0) * 1e-6
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per_q = defaultdict(float)
    for s, e, _, q in sel:
        per_q[q] += (e - s) * 1e-6
    steps = 8
    print("span %.2f ms = %.2f ms/step; union of kernel intervals %.2f ms/step; idle %.2f ms/step (%.1f %%)"
          % (span, span / steps, busy * 1e-6 / steps, (span - busy * 1e-6) / steps, 100 * (1 - busy * 1e-6 / span)))
    for q, v in sorted(per_q.items(), key=lambda kv: -kv[1]):
        print("   queue %s: %.2f ms/step of kernels" % (q, v / steps))
    # gaps on the busiest queue
    q0 = max(per_q, key=per_q.get)
    ks = [(s, e) for s, e, _, q in sel if q == q0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    gaps = [g for g in gaps if g > 0]
    print("   busiest queue: %d kernels/step, sum of gaps %.2f ms/step, median gap %.2f us, gaps > 20 us: %d/step"
          % (len(ks) / steps, sum(gaps) * 1e-6 / steps, sorted(gaps)[len(gaps) // 2] * 1e-3, sum(g > 20000 for g in gaps) / steps))


if __name__ == "__main__":
    main(sys.argv[1])
