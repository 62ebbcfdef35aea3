"""Attention microbenchmark on the shapes of the benchmark step (one MI355X)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panfusion_amd import ops  # noqa: E402

SHAPES = {  # name: (B, H, D, nq, nk)
    "self64": (40, 5, 64, 4096, 4096),
    "self32": (40, 10, 64, 1024, 1024),
    "self16": (40, 20, 64, 256, 256),
    "pano64": (2, 5, 64, 8192, 8192),
    "text64": (40, 5, 64, 4096, 77),
    "text32": (40, 10, 64, 1024, 77),
    "text16": (40, 20, 64, 256, 77),
    "textpano": (2, 5, 64, 8192, 77),
    "epa_e": (2, 20, 32, 2048, 20480),
    "epa_p": (2, 20, 32, 20480, 2048),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--pp-timing", action="store_true", help="with PF_HIP_LIB=panfusion_amd/abl/lib_attn_pp_timing.so (make -C panfusion_amd/csrc "
                    "attn_pp_timing): per-wave clocks per key tile of the ping-pong kernel's vector / matrix segments and barrier waits")
    args = ap.parse_args()
    dev = "cuda"
    for name in args.shapes.split(","):
        B, H, D, nq, nk = SHAPES[name]
        g = torch.Generator(device=dev).manual_seed(1)
        C = H * D
        q = torch.randn(B * nq, C, device=dev, generator=g).to(torch.bfloat16)
        k = torch.randn(B * nk, C, device=dev, generator=g).to(torch.bfloat16)
        ld = (nk + 31) // 32 * 32
        vt = torch.randn(B, C, ld, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(B, nq, C, device=dev, dtype=torch.bfloat16)
        kw = dict(q_ld=C, k_ld=C, vt_ld=ld, q_bs=nq * C, k_bs=nk * C, vt_bs=C * ld, out=out)
        for _ in range(2):
            ops.attention(q, k, vt, B, H, D, nq, nk, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(0.02 * 2.4e9))
        e0.record()
        for _ in range(args.reps):
            ops.attention(q, k, vt, B, H, D, nq, nk, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        fl = 4.0 * B * H * nq * nk * D
        gb = 2.0 * (2 * B * nq * C + B * nk * C + B * C * ld)          # q + out, k + vt: the bytes that must move once
        print("%-8s B%-3d H%-3d D%-3d nq%-6d nk%-6d %9.1f us  %7.1f TF/s  %5.2f TB/s algorithmic" % (name, B, H, D, nq, nk, us, fl / us / 1e6, gb / us / 1e6), flush=True)
        if args.pp_timing and D == 64 and nk % 8 == 0 and nk >= 128:
            lse = torch.zeros(B, H, nq, device=dev, dtype=torch.float32)
            ops.attention(q, k, vt, B, H, D, nq, nk, lse=lse, **kw)
            torch.cuda.synchronize()
            r = lse.view(-1, 32)[:, :6].cpu()                     # one record per wave (32 query rows)
            for grp in (0, 1):
                g = r[r[:, 3] == grp]
                if len(g):
                    print("   group %s: %5d waves  per key tile: vector segment %6.0f  matrix segment %6.0f  barrier wait %6.0f  (s_memtime ticks)   waves %s on SIMDs %s"
                          % ("AB"[grp], len(g), g[:, 0].mean(), g[:, 1].mean(), g[:, 2].mean(), sorted(set(int(v) for v in g[:64, 5])), sorted(set(int(v) for v in g[:64, 4]))))


if __name__ == "__main__":
    main()
