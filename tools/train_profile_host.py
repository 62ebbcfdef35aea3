"""Host-side profile (cProfile) of one training step of tools/train_bench.py's configuration: where the Python time of
the launch-bound step goes.   python tools/train_profile_host.py [--steps 2]"""
import argparse
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import bench
    from panfusion_amd.models.sd2_unet_params import SD2_BASE
    from panfusion_amd.utils.pano import icosahedron_sample_camera
    dev = torch.device("cuda")
    cfg = dict(SD2_BASE)
    model = bench.build_model(dev, torch.float16, cfg)
    model.differentiable = True
    th, ph = icosahedron_sample_camera()
    m = len(th)
    g = lambda s: torch.Generator().manual_seed(s)
    latents, pano = torch.randn(1, m, 4, 32, 32, generator=g(0)).to(dev), torch.randn(1, 1, 4, 64, 128, generator=g(1)).to(dev)
    prompt, pprompt = torch.randn(1, m, 77, 1024, generator=g(4)).to(dev), torch.randn(1, 1, 77, 1024, generator=g(5)).to(dev)
    cams = {"FoV": torch.full((1, m), 90), "theta": torch.tensor(np.degrees(th), dtype=torch.float64)[None],
            "phi": torch.tensor(np.degrees(ph), dtype=torch.float64)[None]}
    t = torch.full((1, m), 500, device=dev)
    params = model.trainable_tensors()
    opt = torch.optim.AdamW(params, lr=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        a, b = model(latents, pano, t, prompt, pprompt, cams)
        (a.square().mean() + b.square().mean()).backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    print("== whole steps (the backward runs on the autograd engine's thread: one opaque run_backward line here)")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    # the backward alone, driven from this thread
    from panfusion_amd import train_engine as TE
    args_ = (latents, pano, t, prompt, pprompt, cams, None, None)
    pr = cProfile.Profile()
    for _ in range(args.steps):
        tape = []
        with torch.no_grad():
            a, b, pers, pano_br, side = model._forward(*args_, tape=tape)
            d = {pers: torch.randn_like(a).flatten(0, 1).contiguous(), pano_br: torch.randn_like(b).flatten(0, 1).contiguous()}
            torch.cuda.synchronize()
            pr.enable()
            TE.backward(tape, d, TE.ParamGrads())
            torch.cuda.synchronize()
            pr.disable()
    print("== train_engine.backward alone")
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
