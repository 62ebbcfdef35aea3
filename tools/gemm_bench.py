"""GEMM / conv microbenchmark on the shapes of the benchmark step (one MI355X).
    python tools/gemm_bench.py [--reps 20] [--shapes conv64,ff1,...]
Prints TF/s per shape; under rocprofv3 --pmc the per-kernel counters give the stall breakdown."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panfusion_amd import ops  # noqa: E402

SHAPES = {
    # name: (n_img, h, w, cin, cout, ksize, extras)
    "conv64": (40, 64, 64, 320, 320, 3, {}),
    "conv64cat": (40, 64, 64, 640, 320, 3, {}),
    "conv32": (40, 32, 32, 640, 640, 3, {}),
    "conv16": (40, 16, 16, 1280, 1280, 3, {}),
    "conv8": (40, 8, 8, 1280, 1280, 3, {}),
    "lin320": (1, 1, 163840, 320, 320, 1, {"res": True}),
    "qk320": (1, 1, 163840, 320, 640, 1, {}),
    "ff1_320": (1, 1, 163840, 320, 2560, 1, {"geglu": True}),
    "ff2_320": (1, 1, 163840, 1280, 320, 1, {"res": True}),
    "lin640": (1, 1, 40960, 640, 640, 1, {"res": True}),
    "ff1_640": (1, 1, 40960, 640, 5120, 1, {"geglu": True}),
    "lin1280": (1, 1, 10240, 1280, 1280, 1, {"res": True}),
    "ff1_1280": (1, 1, 10240, 1280, 10240, 1, {"geglu": True}),
    "pano_conv64": (2, 64, 132, 320, 320, 3, {}),
    "pano_conv8": (2, 8, 20, 1280, 1280, 3, {}),
    "pano_conv16": (2, 16, 36, 1280, 1280, 3, {}),
    "pano_conv32": (2, 32, 68, 640, 640, 3, {}),
    "conv8cat": (40, 8, 8, 2560, 1280, 3, {}),
    "lin_mid": (1, 1, 2560, 1280, 1280, 1, {"res": True}),
    # resnet conv2: + residual (fp32 stream in / out with --stream32: the 17-25 k-clock epilogue of DESIGN.md section 3.1)
    # whole rounds of the 32x32x16 kernel's 256 x 320 tiles (no split-K tail launch): 2 rounds, 5 rounds, 1 round
    "conv64_2r": (32, 64, 64, 320, 320, 3, {}),
    "conv64_2r_res": (32, 64, 64, 320, 320, 3, {"res": True}),
    "conv64_n640": (40, 64, 64, 640, 640, 3, {}),
    "conv32_1r": (32, 32, 32, 640, 640, 3, {}),
    # small-M linears of the panorama branch / the deep levels (180 launches per step at 0.106 of peak)
    "sm1024_1280": (1, 1, 1024, 1280, 1280, 1, {}),
    "sm1024_2560": (1, 1, 1024, 2560, 1280, 1, {}),
    "sm256_1280": (1, 1, 256, 1280, 1280, 1, {}),
    "sm256_5120": (1, 1, 256, 5120, 1280, 1, {}),
    "sm4096_640": (1, 1, 4096, 640, 640, 1, {}),
    "sm4096_1280": (1, 1, 4096, 1280, 640, 1, {}),
    "sm1024_ff1": (1, 1, 1024, 1280, 10240, 1, {"geglu": True}),
    "sm2560_1280": (1, 1, 2560, 1280, 1280, 1, {}),
    "conv64res": (40, 64, 64, 320, 320, 3, {"res": True}),
    "conv32res": (40, 32, 32, 640, 640, 3, {"res": True}),
    "conv16res": (40, 16, 16, 1280, 1280, 3, {"res": True}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--phases", action="store_true", help="per-block phase stamps (pf_debug_gemm_profile)")
    ap.add_argument("--timeline", action="store_true", help="with PF_HIP_LIB=panfusion_amd/abl/lib_timeline.so (make -C panfusion_amd/csrc timeline): "
                    "phases of every block's SECOND tile in the persistent 8-wave kernel, shader clocks")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--gn", action="store_true", help="ask the epilogue for GroupNorm moments (pf_conv_desc.gn_partial)")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "ones", "small"],
                    help="operand values: the chip clocks to its power budget (MI355X_MICROARCH.md, DVFS give-back), so the same kernel runs faster on "
                         "zero / constant operands -- the difference is what the power cap costs")
    ap.add_argument("--stream32", action="store_true", help="fp32 residual in / fp32 out where the shape has a residual (mixed scheme)")
    args = ap.parse_args()
    dev = "cuda"
    T16 = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    for name in args.shapes.split(","):
        n, h, w, cin, cout, ks, ex = SHAPES[name]
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(n, h, w, cin, device=dev, generator=g).to(T16)
        wt = (torch.randn(cout, ks * ks * cin, device=dev, generator=g) / (ks * ks * cin) ** 0.5).to(T16)
        if args.data == "zeros":
            x.zero_(); wt.zero_()
        elif args.data == "ones":
            x.fill_(1.0); wt.fill_(1.0 / (ks * ks * cin))
        elif args.data == "small":                                 # few significant bits toggling: +-1 activations, +-2^-k weights
            x.sign_(); wt = (wt.sign() / 64).to(T16)
        b = torch.randn(cout, device=dev, generator=g)
        M = n * h * w
        n_store = cout // 2 if ex.get("geglu") else cout
        TS = torch.float32 if (args.stream32 and ex.get("res")) else T16
        res = torch.randn(M, n_store, device=dev, generator=g).to(TS) if ex.get("res") else None
        out = torch.empty(M, n_store, device=dev, dtype=TS)
        kw = dict(n_img=n, h_in=h, w_in=w, ksize=ks, pad=ks // 2, bias=b, residual=res, out=out, geglu=bool(ex.get("geglu")),
                  gn_stats=args.gn and not ex.get("geglu"))
        for _ in range(3):
            ops.conv_gemm(x, wt, cout, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(0.02 * 2.4e9))
        e0.record()
        for _ in range(args.reps):
            ops.conv_gemm(x, wt, cout, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        fl = 2.0 * M * cout * ks * ks * cin
        byt = 2.0 * (M * cin + cout * ks * ks * cin + M * n_store * (2 if res is not None else 1))
        if args.timeline:
            from panfusion_amd import _lib
            cap = 1 << 12
            buf = torch.zeros(cap * 32, dtype=torch.int64, device=dev)
            _lib.lib().pf_debug_gemm_profile(buf.data_ptr(), cap)
            ops.conv_gemm(x, wt, cout, **kw)
            torch.cuda.synchronize()
            _lib.lib().pf_debug_gemm_profile(None, 0)
            full = buf.view(cap, 32).cpu()
            full = full[(full[:, 12] != 0) & (full[:, 23] != 0)].double()
            if full.shape[0]:
                d = lambda a, b: float((full[:, b] - full[:, a]).mean())
                nst = int((full[0, 15:23] != 0).sum())
                steps = [d(14, 15)] + [d(15 + i - 1, 15 + i) for i in range(1, nst)]
                print("   timeline of tile #2 (%d blocks): wait+barrier %.0f | stage-2 DMA + first fragments %.0f | K steps %s | epilogue %.0f | tile total %.0f"
                      % (full.shape[0], d(12, 13), d(13, 14), " ".join("%.0f" % v for v in steps), d(15 + nst - 1, 23), d(12, 23)))
            else:
                print("   timeline: no block ran a second tile of the 8-wave kernel (or the library is not the timeline build)")
        if args.phases:
            from panfusion_amd import _lib
            cap = 1 << 16
            buf = torch.zeros(cap * 32, dtype=torch.int64, device=dev)
            _lib.lib().pf_debug_gemm_profile(buf.data_ptr(), cap)
            ops.conv_gemm(x, wt, cout, **kw)
            torch.cuda.synchronize()
            _lib.lib().pf_debug_gemm_profile(None, 0)
            full = buf.view(cap, 32).cpu()
            full = full[full[:, 3] != 0].double()
            st = full[:, :4]
            if float(full[:, 12:28].abs().sum()) > 0:
                w = full[:, 4:28].view(-1, 8, 3).mean(0)
                print("   per-wave K-loop clocks [wait+barrier, DMA issue, ds_read+MFMA]:", " ".join("w%d[%.0f %.0f %.0f]" % (i, *w[i]) for i in range(8)))
            if float(full[:, 4].abs().sum()) > 0 and float(full[:, 12:28].abs().sum()) == 0:     # epilogue sub-stamps
                e = full[:, 2:12]
                seq = [2, 4, 5, 6, 7, 8, 9, 10, 11]
                names = ["ring barrier", "next-tile setup+DMA issue", "fragments->LDS (slice 0)", "barrier", "copy-out issue",
                         "fragments->LDS (slice 1)", "barrier", "copy-out issue"]
                print("   epilogue: " + "  ".join("%s %.0f" % (n, (full[:, b] - full[:, a]).mean()) for n, a, b in zip(names, seq[:-1], seq[1:])))
            if float(full[:, 6].abs().sum()) > 0 and float(full[:, 8:28].abs().sum()) == 0:       # the 32x32x16 kernel's accumulated clocks
                tiles, nit = full[:, 6], full[:, 7]
                print("   g32: %d blocks, %.1f tiles per block, %d K stages per tile: K loop %.0f clocks per tile = %.0f per stage; between K loops (ring barrier, "
                      "next tile's DMA issue, epilogue, landing wait) %.0f per tile" % (full.shape[0], tiles.mean(), nit.mean(),
                      (full[:, 4] / tiles).mean(), (full[:, 4] / tiles / nit).mean(), (full[:, 5] / tiles).mean()))
            d = (st[:, 1:] - st[:, :-1])
            span = float(st[:, 3].max() - st[:, 0].min())
            print("   phases (shader clocks, %d blocks): first tile %.0f  K loop %.0f  epilogue %.0f  | block total %.0f  kernel span %.0f (100 MHz memtime ticks?)"
                  % (st.shape[0], d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), (st[:, 3] - st[:, 0]).mean(), span))
        print("%-12s M%-7d N%-6d K%-6d  %8.1f us  %7.1f TF/s   algorithmic HBM %6.2f TB/s" % (name, M, cout, ks * ks * cin, us, fl / us / 1e6, byt / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
