"""Round 6: board power, shader clock and ENERGY PER LAUNCH of the step's kernel classes, one at a time (one MI355X).

    python tools/power_by_kernel.py [--seconds 5] [--cases conv64,self64,...]

The benchmark step runs at the board's power limit (profiles/r6_power_probe.txt), so what a launch costs the step is its
ENERGY, not its issue slots.  Every case loops one launch on N(0,1) operands for --seconds; a thread samples the busiest card's sysfs hwmon
sensors (power1_input, freq1_input) every 100 ms and the middle 60 % of the loop is kept.  Printed per case: sustained us per launch, median
board power and clock, joules per launch, and pJ per algorithmic FLOP (matrix kernels) or per algorithmic byte (stream kernels)."""
import argparse
import glob
import os
import statistics
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panfusion_amd import ops  # noqa: E402

DEV = "cuda"


def sensors():
    best = (0, 0)
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        try:
            try:
                p = int(open(h + "/power1_input").read())
            except OSError:
                p = int(open(h + "/power1_average").read())
            f = int(open(h + "/freq1_input").read())
        except (OSError, ValueError):
            continue
        if p > best[0]:
            best = (p, f)
    return best[0] / 1e6, best[1] / 1e6


def conv_case(n, h, w, cin, cout, ks, res32=False, geglu=False):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g).half()
    wt = (torch.randn(cout, ks * ks * cin, device=DEV, generator=g) / (ks * ks * cin) ** 0.5).half()
    b = torch.randn(cout, device=DEV, generator=g)
    M = n * h * w
    ns = cout // 2 if geglu else cout
    res = torch.randn(M, ns, device=DEV, generator=g) if res32 else None
    out = torch.empty(M, ns, device=DEV, dtype=torch.float32 if res32 else torch.float16)
    kw = dict(n_img=n, h_in=h, w_in=w, ksize=ks, pad=ks // 2, bias=b, residual=res, out=out, geglu=geglu)
    fl = 2.0 * M * cout * ks * ks * cin
    byt = 2.0 * (M * cin + cout * ks * ks * cin) + M * ns * (8 if res32 else 2)
    return (lambda: ops.conv_gemm(x, wt, cout, **kw)), fl, byt


def linear_case(rows, K, N, res32=False, geglu=False):
    """ops.linear: the weight-stationary kernel where it serves the shape (as in the step), the tile kernel otherwise."""
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(rows, K, device=DEV, generator=g).half()
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).half()
    b = torch.randn(N, device=DEV, generator=g)
    ns = N // 2 if geglu else N
    res = torch.randn(rows, ns, device=DEV, generator=g) if res32 else None
    out = torch.empty(rows, ns, device=DEV, dtype=torch.float32 if res32 else torch.float16)
    return (lambda: ops.linear(x, w, bias=b, residual=res, out=out, geglu=geglu)), 2.0 * rows * N * K, 2.0 * (rows * K + N * K) + rows * ns * (8 if res32 else 2)


def attn_case(B, H, D, nq, nk):
    g = torch.Generator(device=DEV).manual_seed(1)
    C = H * D
    q = torch.randn(B * nq, C, device=DEV, generator=g).half()
    k = torch.randn(B * nk, C, device=DEV, generator=g).half()
    ld = (nk + 31) // 32 * 32
    vt = torch.randn(B, C, ld, device=DEV, generator=g).half()
    out = torch.empty(B, nq, C, device=DEV, dtype=torch.float16)
    kw = dict(q_ld=C, k_ld=C, vt_ld=ld, q_bs=nq * C, k_bs=nk * C, vt_bs=C * ld, out=out)
    return (lambda: ops.attention(q, k, vt, B, H, D, nq, nk, **kw)), 4.0 * B * H * nq * nk * D, 2.0 * (2 * B * nq * C + B * nk * C + B * C * ld)


def elem_case(kind, n, hw, C):
    x32 = torch.randn(n, hw, C, device=DEV)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    E = n * hw * C
    if kind == "apply":
        sc, sh = ops.groupnorm_scale_shift(x32, None, n, hw, 32, 1e-5, g, b)
        return (lambda: ops.scale_shift_act(x32, None, n, hw, sc, sh, 1, out_dtype=torch.float16)), 0.0, 6.0 * E
    if kind == "layernorm":
        return (lambda: ops.layernorm(x32.view(-1, C), g, b, 1e-5, out_dtype=torch.float16)), 0.0, 6.0 * E
    if kind == "gnstats":
        return (lambda: ops.groupnorm_scale_shift(x32, None, n, hw, 32, 1e-5, g, b)), 0.0, 4.0 * E
    raise KeyError(kind)


CASES = {
    # matrix kernels (class of DESIGN.md 5's table in brackets)
    "conv64": lambda: conv_case(40, 64, 64, 320, 320, 3),                      # [conv 3x3, M >= 40960]
    "conv32": lambda: conv_case(40, 32, 32, 640, 640, 3),
    "conv16": lambda: conv_case(40, 16, 16, 1280, 1280, 3),                    # [conv 3x3, small M]
    "conv64res32": lambda: conv_case(40, 64, 64, 320, 320, 3, res32=True),     # resnet conv2: fp32 residual in, fp32 stream out
    "ff1_320": lambda: linear_case(163840, 320, 2560, geglu=True),             # [linear K <= 640] weight-stationary kernel, GEGLU epilogue
    "qkv320": lambda: linear_case(163840, 320, 960),                           # q | k | v, weight-stationary kernel
    "ff2_320": lambda: linear_case(163840, 1280, 320, res32=True),             # [linear K 641-2000] tile kernel, fp32 token stream
    "lin320": lambda: linear_case(163840, 320, 320, res32=True),               # [linear K <= 640] fp32 token stream: HBM-bound
    "ff1_640": lambda: linear_case(40960, 640, 5120, geglu=True),
    "ff1_1280": lambda: linear_case(10240, 1280, 10240, geglu=True),
    "small1024": lambda: linear_case(1024, 1280, 1280),                        # [linear small M]
    "self64": lambda: attn_case(40, 5, 64, 4096, 4096),                        # [attention self]
    "epa_e": lambda: attn_case(2, 20, 32, 2048, 20480),                        # [attention EPA] (no bias table here)
    "text64": lambda: attn_case(40, 5, 64, 4096, 77),                          # [attention text]
    # stream kernels
    "apply64": lambda: elem_case("apply", 40, 4096, 320),                      # GroupNorm-apply + SiLU, fp32 stream -> fp16
    "layernorm64": lambda: elem_case("layernorm", 40, 4096, 320),
    "gnstats64": lambda: elem_case("gnstats", 40, 4096, 320),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--cases", default=",".join(CASES))
    args = ap.parse_args()
    idle = [sensors() for _ in range(5) if not time.sleep(0.1)]
    print("idle: %.0f W, %.0f MHz" % (statistics.median(p for p, _ in idle), statistics.median(f for _, f in idle)))
    print("%-12s %10s %8s %8s %10s %12s %12s" % ("case", "us/launch", "W", "MHz", "J/launch", "pJ/FLOP", "pJ/byte"))
    for name in args.cases.split(","):
        fn, fl, byt = CASES[name]()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us0 = e0.elapsed_time(e1) * 1e3 / 20
        reps = max(50, int(args.seconds * 1e6 / us0))
        samples, stop = [], threading.Event()

        def watch():
            t0 = time.time()
            while not stop.is_set():
                samples.append((time.time() - t0,) + sensors())
                time.sleep(0.1)
        th = threading.Thread(target=watch)
        th.start()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1) / 1e3
        stop.set()
        th.join()
        mid = [s for s in samples if 0.2 * total <= s[0] <= 0.8 * total] or samples
        W = statistics.median(s[1] for s in mid)
        F = statistics.median(s[2] for s in mid)
        us = total * 1e6 / reps
        J = W * us * 1e-6
        print("%-12s %10.1f %8.0f %8.0f %10.4f %12s %12s   (%d launches, %.1f s, %d samples; short burst %.1f us)"
              % (name, us, W, F, J, "%.2f" % (J / fl * 1e12) if fl else "-", "%.1f" % (J / byt * 1e12), reps, total, len(mid), us0), flush=True)
        time.sleep(1.0)


if __name__ == "__main__":
    main()
