"""Time of the VAE decode that follows the sampling loop (PanFusion.py:166-172): 20 view latents 64x64 -> 512^2 images
and the 64x(128+16) padded panorama latent -> 512x1152 -> crop, SD-2 VAE widths, synthetic weights.

    python tools/vae_bench.py [--dtype fp16|bf16] [--precision mixed|fast] [--reps 3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--precision", default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--views", type=int, default=20)
    ap.add_argument("--shapes", action="store_true", help="per-shape table of the decode's GEMM launches (device events, one extra pass)")
    args = ap.parse_args()
    from panfusion_amd import vae as PV
    from panfusion_amd.models.sd2_unet_params import fill_synthetic
    from panfusion_amd.models.vae_params import SD2_VAE, VAEDecoderParams
    dev = torch.device("cuda")
    with torch.device(dev):
        params = VAEDecoderParams(**SD2_VAE)
    fill_synthetic(params, 9)
    dec = PV.VAEDecoder(params, compute_dtype={"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype], precision=args.precision)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, args.views, 4, 64, 64, generator=g).to(dev)
    pano = torch.randn(1, 1, 4, 64, 128, generator=g).to(dev)
    from panfusion_amd import ops
    ops.TRACE = []                                           # algorithmic FLOPs of the decode: summed over its MFMA launches
    PV.decode_views_and_pano(lat, pano, dec)
    torch.cuda.synchronize()
    trace, ops.TRACE = ops.TRACE, None
    flop = sum(t[1] for t in trace)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        images, pano_img = PV.decode_views_and_pano(lat, pano, dec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    print("VAE decode of %d views + padded panorama, %s %s: %.1f ms  (%.1f TFLOP in its MFMA launches: %.0f TF/s)  peak memory %.1f GB"
          % (args.views, args.dtype, dec.precision, dt * 1e3, flop / 1e12, flop / dt / 1e12, torch.cuda.max_memory_allocated() / 2 ** 30))
    if args.shapes:
        shapes, tot = {}, 0.0
        for name, fl, e0, e1, tag in trace:
            a = shapes.setdefault("%s %s" % (name, tag), [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:24]:
            tot += v[1]
            print("    %-64s launches %3d  %8.3f ms  %7.1f TF/s" % (k, v[2], v[1] * 1e3, v[0] / v[1] / 1e12))
        print("    MFMA launches listed: %.1f ms of the decode" % (tot * 1e3))
    # the encoder side of a training step (PanFusion.py:66-71): 20 views of 256^2 + the padded 512 x 1152 panorama
    from panfusion_amd.models.vae_params import VAEEncoderParams
    from panfusion_amd.utils.pano import pad_pano
    with torch.device(dev):
        eparams = VAEEncoderParams(**SD2_VAE)
    fill_synthetic(eparams, 10)
    enc = PV.VAEEncoder(eparams, compute_dtype={"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype], precision=args.precision)
    imgs = (torch.rand(1, args.views, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    pano_img = (torch.rand(1, 1, 3, 512, 1024, generator=g) * 2 - 1).to(dev)

    def encode():
        z = PV.encode_image(imgs, enc)
        zp = PV.encode_image(pad_pano(pano_img, 64), enc)
        return z, zp

    encode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        z, zp = encode()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    eflop = (args.views * 0.25 + 1152 / 512) * 0.566e12     # ~0.566 TFLOP per 512^2 image through the encoder
    print("VAE encode of %d views of 256^2 + padded 512x1152 panorama, %s %s: %.1f ms  (~%.0f TF/s algorithmic)  latents %s %s"
          % (args.views, args.dtype, enc.precision, dt * 1e3, eflop / dt / 1e12, tuple(z.shape), tuple(zp.shape)))


if __name__ == "__main__":
    main()
