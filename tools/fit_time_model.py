"""Re-measure the constants of panfusion_amd.sharding.TIME_MODEL at HEAD on one MI355X.

    python tools/fit_time_model.py [--out gpurun_out/r6_time_model.json]

Runs tools/sim_rank.py (one rank of the sharded loop in a single process, collectives replaced by local stand-ins: kernels + graph
segments, no wire time) on the ranks the model is fitted from, per configuration:
    a view-only rank with 7 and with 14 views   -> per_view = (t14 - t7) / 7, base = t7 - 7 per_view
    the panorama owner without views            -> pano_only = t - base
    the panorama owner with 6 views             -> pano = t - base - 6 per_view
    cfg 4 only: the owner without views with the query split of the 32 768-token self-attentions OFF and ON (4 ranks per half)
                                                -> attn = (t_off - t_on) / (1 - 1 / 4)
and writes the raw timings next to the fitted constants.  Paste the constants into sharding.TIME_MODEL and commit the file it printed."""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sim(world, rank, split, extra=(), env=None):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "sim_rank.py"), "--world", str(world), "--ranks", str(rank), "--split", split, *extra]
    e = dict(os.environ, **(env or {}))
    out = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=900).stdout
    m = re.search(r"([0-9.]+) ms/step\s+\(graphs (\w+)\)", out)
    if not m:
        raise SystemExit("sim_rank printed no timing:\n" + out[-2000:])
    print("  world %d rank %d split %-10s %-8s %s -> %s ms (graphs %s)" % (world, rank, split, " ".join(extra), env or "", m.group(1), m.group(2)), flush=True)
    return float(m.group(1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r6_time_model.json"))
    doc = {}
    ap.add_argument("--only", default="cfg2,cfg5,cfg4")
    args = ap.parse_args()
    for name, extra in (("cfg2", ()), ("cfg5", ("--cfg5",)), ("cfg4", ("--cfg4",))):
        if name not in args.only.split(","):
            continue
        print(name, flush=True)
        nosplit = {"PF_SHARD_ATTN_MIN_TOKENS": "1000000000"}
        raw = {"t7": sim(8, 1, "0,7,7,6", extra, env=nosplit), "t14": sim(4, 1, "6,14", extra, env=nosplit),
               "owner0": sim(8, 0, "0,7,7,6", extra, env={"PF_SHARD_ATTN_MIN_TOKENS": "1000000000"}),
               "owner6": sim(4, 0, "6,14", extra, env={"PF_SHARD_ATTN_MIN_TOKENS": "1000000000"})}
        per_view = (raw["t14"] - raw["t7"]) / 7.0
        base = raw["t7"] - 7.0 * per_view
        fit = dict(base=round(base, 3), per_view=round(per_view, 4), pano=round(raw["owner6"] - base - 6.0 * per_view, 3),
                   pano_only=round(raw["owner0"] - base, 3))
        if name == "cfg4":
            raw["owner0_split"] = sim(8, 0, "0,7,7,6", extra)               # the query split on (default thresholds), G = 4
            raw["t7_split"] = sim(8, 1, "0,7,7,6", extra)                   # a view rank's share of the split attentions
            fit["attn"] = round((raw["owner0"] - raw["owner0_split"]) / 0.75, 3)
            fit["attn_helper_measured"] = round(raw["t7_split"] - raw["t7"], 3)    # (the model charges attn / G = attn / 4)
        doc[name] = {"raw_ms": raw, "fit": fit}
        print("  ->", fit, flush=True)
    from panfusion_amd import _lib
    doc["csrc_sha256"] = _lib.source_hash()
    doc["how"] = "tools/fit_time_model.py: tools/sim_rank.py per rank, fp16 mixed, hipGraph segments, collectives replaced by local stand-ins (no wire time)"
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(doc, open(args.out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
