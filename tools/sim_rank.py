"""Per-rank compute time of the sharded step, measured in ONE process on one MI355X.

    python tools/sim_rank.py --world 8 --ranks 0,1 [--steps 6]

torch.distributed is replaced by local stand-ins (an all-gather fills every slot with this rank's
block, a broadcast is a no-op), so the timing is the rank's kernels + the graph-segment structure
without any wire time: an upper bound on what N GPUs can deliver, and the workload on which the tile
plan can be tuned for the smaller per-rank GEMMs of 4 and 8 ranks.  Results are NOT parity-checked
(the gathered tensors are fake); functional coverage of the sharded path is tests/test_sharding_gloo.py
and tools/gpu_dist_dry.sh."""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fake_dist(world, rank):
    dist.get_world_size = lambda group=None: world
    dist.get_rank = lambda group=None: rank
    dist.new_group = lambda ranks=None, **k: tuple(ranks)

    class Done:                                   # what async_op=True returns: nothing to wait for
        def wait(self):
            return True

    def all_gather_into_tensor(out, x, group=None, async_op=False):
        n = out.numel() // max(x.numel(), 1) if x.numel() else 0
        if n:
            out.view(n, -1).copy_(x.reshape(1, -1).expand(n, -1))
        return Done() if async_op else None

    dist.all_gather_into_tensor = all_gather_into_tensor
    dist.broadcast = lambda t, src=0, group=None, async_op=False: Done() if async_op else None
    dist.all_reduce = lambda t, op=None, group=None: None
    dist.barrier = lambda *a, **k: None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--ranks", default="0")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--precision", default=None, choices=["mixed", "fast"])
    ap.add_argument("--cfg4", action="store_true", help="BASELINE.json configs[3]: 128 x 256 panorama latent")
    ap.add_argument("--cfg5", action="store_true", help="BASELINE.json configs[4]: + panorama ControlNet on a layout image")
    ap.add_argument("--split", default=None, help="views per group, e.g. 0,7,7,6 (default: sharding.plan's choice for the configuration)")
    ap.add_argument("--no-graphs", action="store_true", help="eager (for a rocprofv3 kernel trace: 4 + warmup + steps passes per rank)")
    ap.add_argument("--shapes", type=int, default=0, help="also print the N most expensive GEMM / attention shapes of one eager step of the rank")
    args = ap.parse_args()
    import bench
    from panfusion_amd import sharding
    from panfusion_amd.models.sd2_unet_params import SD2_BASE
    from panfusion_amd.utils.pano import icosahedron_sample_camera
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    th, ph = icosahedron_sample_camera()
    cams_deg = (np.degrees(th), np.degrees(ph))
    cfg = dict(SD2_BASE)
    pano_hw = (128, 256) if args.cfg4 else (64, 128)
    layout = (torch.rand(1, 1, 3, pano_hw[0] * 8, pano_hw[1] * 8, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(dev) if args.cfg5 else None
    if args.split:
        os.environ["PF_SHARD_SPLIT"] = args.split
    build = lambda steps, graphs: sharding.build_sharded(bench.build_model, bench.build_inputs, dev, dtype, cfg, 20, (64, 64), pano_hw, cams_deg,
                                                         steps, graphs, precision=args.precision, layout_cond=args.cfg5, layout=layout)
    for rank in [int(r) for r in args.ranks.split(",")]:
        fake_dist(args.world, rank)
        dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
        model, loop = build(args.steps + args.warmup + 1, not args.no_graphs)
        loop.prepare()
        for _ in range(args.warmup):
            loop.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        print("world %d rank %d  %-8s %-60s %7.2f ms/step  (graphs %s)" % (args.world, rank, "cfg4" if args.cfg4 else "cfg5" if args.cfg5 else "cfg2",
                                                                      "%s %s %s" % (loop.layout_desc, args.dtype, model.precision), ms, loop.use_graphs), flush=True)
        del model, loop
        torch.cuda.empty_cache()
        if args.shapes:
            from panfusion_amd import ops
            model, loop = build(4, False)
            loop.prepare()
            loop.step()
            torch.cuda.synchronize()
            ops.TRACE = []
            loop.step()
            torch.cuda.synchronize()
            trace, ops.TRACE = ops.TRACE, None
            shapes, fam = {}, {}
            for name, fl, e0, e1, tag in trace:
                dt = e0.elapsed_time(e1) * 1e-3
                for d, k in ((shapes, "%s %s" % (name, tag)), (fam, name)):
                    a = d.setdefault(k, [0.0, 0.0, 0])
                    a[0] += fl
                    a[1] += dt
                    a[2] += 1
            for k, v in fam.items():
                print("    %-64s launches %4d  %8.3f ms  %7.1f TF/s" % (k + " (all)", v[2], v[1] * 1e3, v[0] / v[1] / 1e12))
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:args.shapes]:
                print("    %-64s launches %4d  %8.3f ms  %7.1f TF/s" % (k, v[2], v[1] * 1e3, v[0] / v[1] / 1e12))
            del model, loop
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
