"""Time of one EPA block (WarpAttn) forward + backward on the GPU -- the trainable block of the reference's
training_step (PanFusion.py:64-98: bs 1 per GPU, 20 views; EPA at s = 2, 4, 8), per kernel family.

    python tools/epa_train_bench.py [--dtype fp16|bf16] [--views-latent 32|64] [--reps 5]

--views-latent 32 is the reference's training resolution (256^2 views, README.md:199), 64 the inference geometry.
The panorama latent is 64 x 128.  Levels: (C, s) = (320, 2), (640, 4), (1280, 8).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--views-latent", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--levels", default="320:2,640:4,1280:8")
    args = ap.parse_args()
    from panfusion_amd import ops
    from panfusion_amd.models.pano import WarpAttn
    from panfusion_amd.utils.pano import icosahedron_sample_camera
    dev = torch.device("cuda")
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    th, ph_ = icosahedron_sample_camera()
    m = len(th)
    cams = {"FoV": torch.full((m,), 90), "theta": torch.tensor(np.degrees(th), dtype=torch.float64),
            "phi": torch.tensor(np.degrees(ph_), dtype=torch.float64)}
    for lv in args.levels.split(","):
        Cc, s = (int(v) for v in lv.split(":"))
        ph, eh = args.views_latent // s, 64 // s
        torch.manual_seed(Cc)
        blk = WarpAttn(Cc, compute_dtype=dtype).to(dev)
        with torch.no_grad():
            for p in blk.transformer.parameters():
                p.copy_(torch.randn_like(p) * (0.1 if p.dim() == 1 else p.shape[-1] ** -0.5))
            blk.transformer.norm1.weight.add_(1.0)
            blk.transformer.norm2.weight.add_(1.0)
        xp = torch.randn(m, Cc, ph, ph, device=dev, requires_grad=True)
        xe = torch.randn(1, Cc, eh, 2 * eh, device=dev, requires_grad=True)
        gp, ge = torch.randn_like(xp) * 1e-5, torch.randn_like(xe) * 1e-5
        P, E = ph * ph, eh * 2 * eh
        T = E + m * P
        H = Cc // 32
        attn_fwd = 2 * 4.0 * H * E * m * P * 32
        lin_fwd = 2.0 * T * Cc * Cc * (3 + 1 + 8 + 4)
        fwd_flop, bwd_flop = attn_fwd + lin_fwd, 2.5 * attn_fwd + attn_fwd + 2 * lin_fwd + lin_fwd    # bwd incl. the recompute

        def step():
            op, oe = blk(xp, xe, cams)
            torch.autograd.backward([op, oe], [gp, ge])

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.reps

        t_step = timed(step)
        with torch.no_grad():
            t_fwd = timed(lambda: blk(xp, xe, cams))
        # kernel families of one forward + backward
        ops.TRACE = []
        step()
        torch.cuda.synchronize()
        fam = {}
        for name, fl, e0, e1, tag in ops.TRACE:
            a = fam.setdefault(name, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        ops.TRACE = None
        print("C=%d s=%d  views %dx%d^2, panorama %dx%d: E=%d mP=%d  forward %.2f ms  forward+backward %.2f ms  "
              "(algorithmic %.2f / %.2f TFLOP -> %.0f TF/s over the step)  peak memory %.2f GB"
              % (Cc, s, m, ph, eh, 2 * eh, E, m * P, t_fwd * 1e3, t_step * 1e3, fwd_flop / 1e12, (fwd_flop + bwd_flop) / 1e12,
                 (fwd_flop + bwd_flop) / t_step / 1e12, torch.cuda.max_memory_allocated() / 2 ** 30))
        for name, (fl, sec, n) in sorted(fam.items()):
            print("    %-18s launches %3d  %8.3f ms  %7.1f TF/s" % (name, n, sec * 1e3, fl / sec / 1e12))


if __name__ == "__main__":
    main()
