"""Time of ONE TRAINING STEP of the dual-branch denoiser through the reference's boundary (PanFusion.training_step,
PanFusion.py:64-98: one mv_base_model call without CFG, MSE on both outputs; bs 1 per GPU, 20 views of 256^2 and the
512 x 1024 panorama -- README.md:199, PanoDataset.py:224,227): forward on the inference kernels, backward on
train_engine's tape, gradients for the 91 EPA tensors and the 512 LoRA matrices.  SD-2-base widths, synthetic weights.

    python tools/train_bench.py [--dtype fp16|bf16] [--views-latent 32|64] [--steps 3] [--small]
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
    ap.add_argument("--precision", default=None)
    ap.add_argument("--views-latent", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args()
    import bench
    from panfusion_amd import ops
    from panfusion_amd.models.sd2_unet_params import SD2_BASE
    from panfusion_amd.utils.pano import icosahedron_sample_camera
    dev = torch.device("cuda")
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    cfg = dict(SD2_BASE)
    lat, pano_hw = args.views_latent, (64, 128)
    if args.small:
        cfg.update(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=128)
        lat, pano_hw = 16, (16, 32)
    model = bench.build_model(dev, dtype, cfg, precision=args.precision)
    model.differentiable = True
    th, ph = icosahedron_sample_camera()
    m = len(th)
    g = lambda s: torch.Generator().manual_seed(s)
    latents = torch.randn(1, m, 4, lat, lat, generator=g(0)).to(dev)
    pano_latent = torch.randn(1, 1, 4, *pano_hw, generator=g(1)).to(dev)
    noise, pano_noise = torch.randn(latents.shape, generator=g(2)).to(dev), torch.randn(pano_latent.shape, generator=g(3)).to(dev)
    prompt = torch.randn(1, m, 77, cfg["cross_attention_dim"], generator=g(4)).to(dev)
    pano_prompt = torch.randn(1, 1, 77, cfg["cross_attention_dim"], generator=g(5)).to(dev)
    cams = {"FoV": torch.full((1, m), 90), "theta": torch.tensor(np.degrees(th), dtype=torch.float64)[None],
            "phi": torch.tensor(np.degrees(ph), dtype=torch.float64)[None]}
    t = torch.full((1, m), 500, device=dev)
    params = model.trainable_tensors()
    opt = torch.optim.AdamW(params, lr=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        pred, pano_pred = model(latents, pano_latent, t, prompt, pano_prompt, cams)
        loss = torch.nn.functional.mse_loss(pred, noise) + torch.nn.functional.mse_loss(pano_pred, pano_noise)
        loss.backward()
        opt.step()
        return loss

    step()                                              # tables, packs, kernel attributes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    with torch.no_grad():
        model(latents, pano_latent, t, prompt, pano_prompt, cams)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model(latents, pano_latent, t, prompt, pano_prompt, cams)
        torch.cuda.synchronize()
        dt_f = (time.perf_counter() - t1) / args.steps
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
    n_grad = sum(p.grad is not None for p in params)
    print("training step (%s %s): %d views of %d^2 + %dx%d panorama, one sample: %.1f ms / step (forward alone %.1f ms), loss %.4f, "
          "%d / %d trainable tensors with gradients, peak memory %.1f GB"
          % (args.dtype, model.precision, m, lat * 8, pano_hw[0] * 8, pano_hw[1] * 8, dt * 1e3, dt_f * 1e3, float(loss), n_grad, len(params),
             torch.cuda.max_memory_allocated() / 2 ** 30))
    for name, (fl, sec, n) in sorted(fam.items()):
        print("    %-18s launches %5d  %8.2f ms  %7.1f TF/s" % (name, n, sec * 1e3, fl / sec / 1e12))


if __name__ == "__main__":
    main()
