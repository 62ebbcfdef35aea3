"""Microbenchmark of the weight-stationary linear kernel (pf_linear_ws) against the tile kernel (pf_conv_gemm) on the
C = 320 layer shapes of the step.   python tools/lws_bench.py [--reps 20] [--rows 163840]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panfusion_amd import ops  # noqa: E402


def timeit(fn, reps):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=163840)
    ap.add_argument("--batch", type=int, default=40)
    args = ap.parse_args()
    dev, T = "cuda", torch.float16
    M, K = args.rows, 320
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, K, device=dev, generator=g).to(T)
    res = torch.randn(M, 320, device=dev, generator=g)
    for name, N in (("q (16-bit out)", 320), ("q|k", 640), ("q|k|v", 960), ("out-proj fp32+res", 320), ("FF1 GEGLU", 2560)):
        w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(T)
        b = torch.randn(N, device=dev, generator=g)
        fl = 2.0 * M * N * K
        if name == "q|k|v":
            nk = M // args.batch
            new = lambda: ops.linear_ws(x, w, ops.LWS_QKV, rows_per_batch=nk)
            old = lambda: (ops.conv_gemm(x, w[:640], 640, w_in=M), ops.linear_t(x.view(args.batch, nk, K), w[640:]))
        elif name.startswith("out-proj"):
            new = lambda: ops.linear_ws(x, w, ops.LWS_F32, bias=b, residual=res)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b, residual=res)
        elif name.startswith("FF1"):
            new = lambda: ops.linear_ws(x, w, ops.LWS_GEGLU, bias=b)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b, geglu=True)
        else:
            new = lambda: ops.linear_ws(x, w, ops.LWS_16, bias=b)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b)
        t_old, t_new = timeit(old, args.reps), timeit(new, args.reps)
        print("%-20s M%-7d N%-5d  tile kernel %7.1f us %6.1f TF/s | weight-stationary %7.1f us %6.1f TF/s  (x%.2f)"
              % (name, M, N, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, t_old / t_new), flush=True)
    # the K = 640 shape (32^2 level: 40 views x 1024 tokens)
    M, K = args.rows // 4, 640
    x = torch.randn(M, K, device=dev, generator=g).to(T)
    res = torch.randn(M, 640, device=dev, generator=g)
    for name, N in (("q|k K640", 1280), ("FF1 GEGLU K640", 5120), ("q K640", 640)):
        w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(T)
        b = torch.randn(N, device=dev, generator=g)
        fl = 2.0 * M * N * K
        if name.startswith("FF1"):
            new = lambda: ops.linear_ws(x, w, ops.LWS_GEGLU, bias=b)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b, geglu=True)
        elif name.startswith("out-proj"):
            new = lambda: ops.linear_ws(x, w, ops.LWS_F32, bias=b, residual=res)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b, residual=res)
        else:
            new = lambda: ops.linear_ws(x, w, ops.LWS_16, bias=b)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b)
        t_old, t_new = timeit(old, args.reps), timeit(new, args.reps)
        print("%-20s M%-7d N%-5d  tile kernel %7.1f us %6.1f TF/s | weight-stationary %7.1f us %6.1f TF/s  (x%.2f)"
              % (name, M, N, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, t_old / t_new), flush=True)

    # the K = 1280 shape (16^2 level: 40 views x 256 tokens)
    M, K = args.rows // 16, 1280
    x = torch.randn(M, K, device=dev, generator=g).to(T)
    for name, N in (("to_q K1280", 1280), ("q|k K1280", 2560), ("FF1 GEGLU K1280", 10240)):
        w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(T)
        b = torch.randn(N, device=dev, generator=g)
        fl = 2.0 * M * N * K
        if name.startswith("FF1"):
            new = lambda: ops.linear_ws(x, w, ops.LWS_GEGLU, bias=b)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b, geglu=True)
        else:
            new = lambda: ops.linear_ws(x, w, ops.LWS_16, bias=b)
            old = lambda: ops.conv_gemm(x, w, N, w_in=M, bias=b)
        t_old, t_new = timeit(old, args.reps), timeit(new, args.reps)
        print("%-20s M%-7d N%-5d  tile kernel %7.1f us %6.1f TF/s | weight-stationary %7.1f us %6.1f TF/s  (x%.2f)"
              % (name, M, N, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, t_old / t_new), flush=True)


if __name__ == "__main__":
    main()
