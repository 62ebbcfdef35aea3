"""Achieved HBM bandwidth of the element-wise / normalisation kernels at the step's level-0 and level-1 sizes.
    python tools/elem_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panfusion_amd import ops  # noqa: E402

DEV = "cuda"


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(0.02 * 2.4e9))
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    for n, hw, C in ((40, 4096, 320), (40, 1024, 640), (40, 256, 1280), (2, 8448, 320)):
        x32 = torch.randn(n, hw, C, device=DEV)
        x16 = x32.half()
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        sc, sh = ops.groupnorm_scale_shift(x32, None, n, hw, 32, 1e-5, g, b)
        E = n * hw * C
        rows = [
            ("gn stats fp32", lambda: ops.groupnorm_scale_shift(x32, None, n, hw, 32, 1e-5, g, b), 4 * E),
            ("gn stats fp16", lambda: ops.groupnorm_scale_shift(x16, None, n, hw, 32, 1e-5, g, b), 2 * E),
            ("apply fp32->fp16 silu", lambda: ops.scale_shift_act(x32, None, n, hw, sc, sh, 1, out_dtype=torch.float16), 6 * E),
            ("apply fp16->fp16 silu", lambda: ops.scale_shift_act(x16, None, n, hw, sc, sh, 1), 4 * E),
            ("apply fp32->split", lambda: ops.scale_shift_act(x32, None, n, hw, sc, sh, 0, out_dtype=torch.float16, split=True), 8 * E),
            ("identity fp32->split", lambda: ops.scale_shift_act(x32, None, 1, n * hw, None, None, 0, out_dtype=torch.float16, split=True), 8 * E),
            ("layernorm fp32->fp16", lambda: ops.layernorm(x32.view(-1, C), g, b, 1e-5, out_dtype=torch.float16), 6 * E),
            ("layernorm fp16->fp16", lambda: ops.layernorm(x16.view(-1, C), g, b, 1e-5), 4 * E),
        ]
        print("n %d hw %d C %d  (%.0f MB fp32)" % (n, hw, C, 4 * E / 1e6))
        for name, fn, byt in rows:
            us = timed(fn)
            print("   %-26s %8.1f us   %5.2f TB/s" % (name, us, byt / us / 1e6))

    # the UNets' head (GroupNorm-apply + SiLU + conv_out 320 -> 4) and conv_in (4 -> 320) at the benchmark's sizes
    for n, h, w, wrap in ((40, 64, 64, False), (2, 64, 128, True), (2, 128, 256, True)):
        C = 320
        x = torch.randn(n, h, w, C, device=DEV)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        sc, sh = ops.groupnorm_scale_shift(x, None, n, h * w, 32, 1e-5, g, b)
        wo = torch.randn(4, C, 3, 3, device=DEV) * 0.02
        w_old, w_t, bo = wo.permute(0, 2, 3, 1).contiguous(), ops.conv_out_weight_t(wo), torch.zeros(4, device=DEV)
        lat = torch.randn(n, 4, h, w, device=DEV)
        wi, bi = torch.randn(3, 3, 4, C, device=DEV) * 0.1, torch.zeros(C, device=DEV)
        E = n * h * w * C

        def two():
            y = ops.scale_shift_act(x, None, n, h * w, sc, sh, 1, out_dtype=torch.float32).view(n, h, w, C)
            return ops.conv_out(y, w_old, bo, 4, wrap=wrap)
        print("head / conv_in  n %d  %d x %d  C %d  wrap %d" % (n, h, w, C, wrap))
        for name, fn, byt in (("head: apply + conv_out", two, 12 * E), ("head: conv_out_gn (fused)", lambda: ops.conv_out_gn(x, sc, sh, 1, w_t, bo, 4, wrap=wrap), 4 * E),
                              ("conv_in -> fp32 stream", lambda: ops.conv_in(lat, wi, bi, C, torch.float32, wrap=wrap), 4 * E)):
            us = timed(fn)
            print("   %-26s %8.1f us   %5.2f TB/s (algorithmic bytes)" % (name, us, byt / us / 1e6))


if __name__ == "__main__":
    main()
