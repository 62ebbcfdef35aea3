"""Time of ONE TRAINING STEP of the dual-branch denoiser through the reference's boundary (PanFusion.training_step,
PanFusion.py:64-98: one mv_base_model call without CFG, MSE on both outputs; bs 1 per GPU, 20 views of 256^2 and the
512 x 1024 panorama -- README.md:199, PanoDataset.py:224,227): forward on the inference kernels, backward on
train_engine's tape, gradients for the 91 EPA tensors and the 512 LoRA matrices.  SD-2-base widths, synthetic weights.

    python tools/train_bench.py [--dtype fp16|bf16] [--views-latent 32|64] [--steps 3] [--small] [--layout-cond]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--precision", default=None)
    ap.add_argument("--views-latent", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="skip the instrumented extra step (profiling runs)")
    ap.add_argument("--shapes", action="store_true", help="per-shape table of the GEMM / attention launches of one step")
    ap.add_argument("--layout-cond", action="store_true", help="layout-conditioned training: the panorama ControlNet trains (all parameters)")
    args = ap.parse_args()
    import bench
    from panfusion_amd.models.sd2_unet_params import SD2_BASE
    dev = torch.device("cuda")
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    cfg = dict(SD2_BASE)
    lat, pano_hw = args.views_latent, (64, 128)
    if args.small:
        cfg.update(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=128)
        lat, pano_hw = 16, (16, 32)
    r = bench.training_step_leg(dev, dtype, cfg, args.precision, lat, pano_hw, args.steps, want_trace=not args.no_trace, layout_cond=args.layout_cond)
    print("training step (%s %s): %s: %.1f ms / step (forward alone %.1f ms), loss %.4f, %d / %d trainable tensors with gradients, "
          "peak memory %.1f GB" % (args.dtype, r["precision"], r["workload"], r["ms_per_step"], r["forward_only_ms"], r["loss"],
                                   r["with_gradient"], r["trainable_tensors"], r["peak_memory_gb"]))
    for name, k in sorted(r.get("kernels", {}).items()):
        print("    %-18s launches %5d  %8.2f ms  %7.1f TF/s" % (name, k["launches"], k["ms"], k["tflops"]))
    if args.shapes:
        for name, k in sorted(r["shapes"].items(), key=lambda kv: -kv[1]["ms"]):
            print("    %-70s launches %4d  %8.3f ms  %7.1f TF/s" % (name, k["launches"], k["ms"], k["tflops"]))


if __name__ == "__main__":
    main()
