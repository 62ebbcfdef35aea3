"""DDIM denoise steps/s of the PanFusion hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the loop body of PanFusion.inference (reference
models/pano/PanFusion.py:146-162): 90-degree latent roll, CFG-paired dual-branch SD-2 UNet
forward with 7 EPA fusions, CFG merge, two DDIM updates.  Workload = BASELINE.json configs[1]:
512x1024 panorama (64x128 latent) + 20 perspective views of 512^2 (64x64 latents), batch 1 prompt
(CFG pair => 40 view samples + 2 pano samples per step), SD-2-base UNet shapes, synthetic seeded
weights and inputs (no checkpoint / dataset is reachable offline).

Prints ONE JSON line (rank 0).  `roofline` is measured live with device events around every launch
of the dominant kernel family in an extra instrumented step after the timed region; `cpu_baseline`
times the CPU oracle (oracle/, test infrastructure) on a bounded sample on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_STEP_CFG2 = 38.66e12        # SURVEY.md §8d / BASELINE.md §2 (2 FLOPs per MAC)
PEAK_BF16_DENSE_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak


def build_model(dev, dtype, cfg, layout_cond=False, precision=None):
    import torch
    from panfusion_amd.models.pano import MultiViewBaseModel
    from panfusion_amd.models.sd2_unet_params import ControlNetParams, UNetParams, fill_synthetic
    with torch.device(dev):
        unet, pano_unet = UNetParams(**cfg), UNetParams(**cfg)
        unet.add_lora(4)
        pano_unet.add_lora(4)
        pano_cn = ControlNetParams(**cfg) if layout_cond else None      # reference default: panorama branch only
    fill_synthetic(unet, 1)
    fill_synthetic(pano_unet, 2)
    if pano_cn is not None:
        fill_synthetic(pano_cn, 3)          # incl. the zero-convs (zero-initialised in diffusers: would skip no work, but be a no-op)
    model = MultiViewBaseModel(unet, pano_unet, None, pano_cn, True, compute_dtype=dtype, precision=precision).to(dev)
    for i, blk in enumerate([*model.cp_blocks_encoder, model.cp_blocks_mid, *model.cp_blocks_decoder]):
        fill_synthetic(blk, 10 + i)     # EPA output projections are zero-initialised in the reference
    model.repack()
    return model


def build_inputs(dev, m, lat_hw, pano_hw, ctx_dim, cams_deg):
    import torch
    from panfusion_amd.pipeline import init_noise
    g = lambda s: torch.Generator().manual_seed(s)
    pano_noise = torch.randn(1, 1, 4, *pano_hw, generator=g(0)).to(dev)
    theta, phi = cams_deg
    cameras = {"FoV": torch.full((1, m), 90), "theta": torch.tensor(theta, dtype=torch.float64)[None],
               "phi": torch.tensor(phi, dtype=torch.float64)[None]}
    _, latents = init_noise(pano_noise, cameras, *lat_hw)
    prompt = torch.randn(1, m, 77, ctx_dim, generator=g(1))
    pano_prompt = torch.randn(1, 1, 77, ctx_dim, generator=g(2))
    null = torch.randn(1, 1, 77, ctx_dim, generator=g(3))
    prompt_embd = torch.cat([null.expand(-1, m, -1, -1), prompt]).to(dev)
    pano_prompt_embd = torch.cat([null, pano_prompt]).to(dev)
    return latents, pano_noise, prompt_embd, pano_prompt_embd, cameras


def cpu_baseline(cfg, ctx_dim, lat_hw, pano_hw, m, cams_deg, flop_per_step):
    """The reference loop body on the host cores, timed through the CPU oracle (oracle/, fp32, all usable cores) on a
    BOUNDED sample of the same workload: ONE REAL CFG SAMPLE of the step -- all m = 20 views at the full
    64 x 64 latents + the 64 x 128 panorama latent through oracle.mvgen.DualBranchDenoiser (= the reference's
    MultiViewBaseModel + WarpAttn on the restated SD-2-base UNets), including the per-call get_masks / get_coords the
    reference rebuilds inside EVERY WarpAttn.forward (models/pano/modules.py:24-27).  A step is two such samples (the CFG
    pair is a batch of two: same arithmetic twice), so value = 1 / (2 t_sample); `masks_cached` is the same with the time
    spent inside the geometry calls taken out.  (Round 2 timed a quarter-size 2-view sample and scaled it by counted FLOPs.)
    The oracle's N x N-materialising attentions run in (sample, head) chunks of <= 2 GiB -- same arithmetic, bounded memory."""
    import torch
    import numpy as np
    from oracle import fixtures as FX
    from oracle import geometry as G
    from oracle import mvgen as MV
    from oracle import sd2_unet as U
    cores = usable_cores()
    torch.set_num_threads(cores)
    with torch.device("meta"):
        unet, pano_unet = U.UNet2DConditionModel(**cfg), U.UNet2DConditionModel(**cfg)
        model = MV.DualBranchDenoiser(unet, pano_unet, None, None, True)
    model = model.to_empty(device="cpu")
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.normal_(0.0, 0.02, generator=gen)
        for mod in model.modules():
            if isinstance(mod, MV._PE):
                mod.freq_bands.copy_(G.spherical_freq_bands(mod.freq_bands.numel()))
        lat, pl = torch.randn(1, m, 4, *lat_hw, generator=gen), torch.randn(1, 1, 4, *pano_hw, generator=gen)
        pe, ppe = torch.randn(1, m, 77, ctx_dim, generator=gen), torch.randn(1, 1, 77, ctx_dim, generator=gen)
        t = torch.full((1, m), 981, dtype=torch.long)
        cams = {"FoV": torch.full((1, m), 90), "theta": torch.tensor(np.asarray(cams_deg[0])[None], dtype=torch.float64),
                "phi": torch.tensor(np.asarray(cams_deg[1])[None], dtype=torch.float64)}
        geo = [0.0]
        real_masks, real_coords = G.get_masks, G.get_coords

        def timed(fn):
            def f(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    geo[0] += time.perf_counter() - t0
            return f
        G.get_masks, G.get_coords = timed(real_masks), timed(real_coords)
        try:
            with FX.chunked_attention():
                model(lat[:, :1, :, :8, :8], pl[..., :8, :16], t[:, :1], pe[:, :1], ppe, {k: v[:, :1] for k, v in cams.items()})  # warm-up (threads, pages)
                geo[0] = 0.0
                t0 = time.perf_counter()
                model(lat, pl, t, pe, ppe, cams)
                t_sample = time.perf_counter() - t0
        finally:
            G.get_masks, G.get_coords = real_masks, real_coords
    t_geo = geo[0]
    return {"value": 1.0 / (2.0 * t_sample), "unit": "steps/s", "cores": cores, "kind": "port",
            "masks_cached": {"value": 1.0 / (2.0 * (t_sample - t_geo)), "unit": "steps/s"},
            "sample": "oracle port of the reference loop body (MultiViewBaseModel + WarpAttn + SD-2-base UNets, fp32, %d threads): "
                      "ONE real CFG sample of the step -- %d views at %dx%d + panorama %dx%d latents, %.2f TFLOP -- timed in full: "
                      "%.1f s, of which %.1f s inside the reference's per-call get_masks / get_coords (7 EPA blocks, 20 cameras); "
                      "a step is two such samples (CFG pair): %.1f s per step"
                      % (cores, m, lat_hw[0], lat_hw[1], pano_hw[0], pano_hw[1], flop_per_step / 2e12, t_sample, t_geo, 2 * t_sample)}


def training_step_leg(dev, dtype, cfg, precision=None, views_latent=32, pano_hw=(64, 128), steps=3, want_trace=False,
                      layout_cond=False):
    """The reference's TRAINING step through the same boundary (PanFusion.training_step, PanFusion.py:64-98; one sample per
    GPU, 20 views of 256^2 + the 512 x 1024 panorama, README.md:199, PanoDataset.py:224,227): forward on the inference
    kernels, backward on train_engine's tape into the 91 EPA tensors and the 512 LoRA matrices, AdamW step.  Not the
    metric of this file: a reported extra (`training_step`), timed after everything else.
    layout_cond: the reference's layout-conditioned training (PanoGenerator.py:153-157, 165-174): the panorama ControlNet is
    added and ALL of its parameters train, next to the EPA blocks; the LoRA matrices do not (`train_lora and not add_cn`)."""
    import numpy as np
    import torch
    from panfusion_amd import ops
    from panfusion_amd.utils.pano import icosahedron_sample_camera
    model = build_model(dev, dtype, cfg, layout_cond=layout_cond, precision=precision)
    model.differentiable = True
    th, ph = icosahedron_sample_camera()
    m = len(th)
    g = lambda s: torch.Generator().manual_seed(s)
    lat = views_latent
    latents = torch.randn(1, m, 4, lat, lat, generator=g(0)).to(dev)
    pano_latent = torch.randn(1, 1, 4, *pano_hw, generator=g(1)).to(dev)
    noise, pano_noise = torch.randn(latents.shape, generator=g(2)).to(dev), torch.randn(pano_latent.shape, generator=g(3)).to(dev)
    prompt = torch.randn(1, m, 77, cfg["cross_attention_dim"], generator=g(4)).to(dev)
    pano_prompt = torch.randn(1, 1, 77, cfg["cross_attention_dim"], generator=g(5)).to(dev)
    cams = {"FoV": torch.full((1, m), 90), "theta": torch.tensor(np.degrees(th), dtype=torch.float64)[None],
            "phi": torch.tensor(np.degrees(ph), dtype=torch.float64)[None]}
    t = torch.full((1, m), 500, device=dev)
    params = model.trainable_tensors()
    kw = {}
    if layout_cond:
        cn_ids = {id(p_) for p_ in model.pano_cn.parameters()}
        epa_ids = {id(p_) for blk in [*model.cp_blocks_encoder, model.cp_blocks_mid, *model.cp_blocks_decoder] for p_ in blk.parameters()}
        for p_ in params:
            p_.requires_grad_(id(p_) in cn_ids or id(p_) in epa_ids)
        params = [p_ for p_ in params if p_.requires_grad]
        kw["pano_layout_cond"] = (torch.rand(1, 1, 3, pano_hw[0] * 8, pano_hw[1] * 8, generator=g(8)) * 2 - 1).to(dev)
    opt = torch.optim.AdamW(params, lr=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        pred, pano_pred = model(latents, pano_latent, t, prompt, pano_prompt, cams, **kw)
        loss = torch.nn.functional.mse_loss(pred, noise) + torch.nn.functional.mse_loss(pano_pred, pano_noise)
        loss.backward()
        opt.step()
        return loss

    step()                                              # tables, packs, kernel attributes
    step()                                              # (the caching allocator settles: the layout-conditioned step peaks at 43 GB)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    with torch.no_grad():
        model(latents, pano_latent, t, prompt, pano_prompt, cams, **kw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            model(latents, pano_latent, t, prompt, pano_prompt, cams, **kw)
        torch.cuda.synchronize()
        dt_f = (time.perf_counter() - t1) / steps
    # the caller side of the step: VAE encode of the 20 views + the padded panorama (PanFusion.py:66-71)
    enc_ms = None
    try:
        from panfusion_amd import vae as PV
        from panfusion_amd.models.sd2_unet_params import fill_synthetic
        from panfusion_amd.models.vae_params import SD2_VAE, VAEEncoderParams
        from panfusion_amd.utils.pano import pad_pano
        with torch.device(dev):
            eparams = VAEEncoderParams(**SD2_VAE)
        fill_synthetic(eparams, 10)
        enc = PV.VAEEncoder(eparams, compute_dtype=dtype, precision=precision)
        imgs = (torch.rand(1, m, 3, lat * 8, lat * 8, generator=g(6)) * 2 - 1).to(dev)
        pimg = (torch.rand(1, 1, 3, pano_hw[0] * 8, pano_hw[1] * 8, generator=g(7)) * 2 - 1).to(dev)
        run = lambda: (PV.encode_image(imgs, enc), PV.encode_image(pad_pano(pimg, 64), enc))
        run()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - t2) / steps * 1e3
    except Exception as exc:                              # (an extra of an extra)
        enc_ms = "%s: %s" % (type(exc).__name__, exc)
    out = {"ms_per_step": dt * 1e3, "forward_only_ms": dt_f * 1e3, "vae_encode_ms": enc_ms, "steps": steps, "loss": float(loss.detach()),
           "trainable_tensors": len(params), "with_gradient": sum(p_.grad is not None for p_ in params),
           "peak_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "precision": model.precision,
           "workload": "%d views of %d^2 + %dx%d panorama, one sample, SD-2-base widths, %s, AdamW"
                       % (m, lat * 8, pano_hw[0] * 8, pano_hw[1] * 8,
                          "panorama ControlNet (all parameters) + EPA trainable" if layout_cond else "rank-4 LoRA")}
    if want_trace:
        ops.TRACE = []
        step()
        torch.cuda.synchronize()
        fam = {}
        trace = ops.TRACE
        for name, fl, e0, e1, tag in trace:
            a = fam.setdefault(name, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        ops.TRACE = None
        out["kernels"] = {k: {"launches": v[2], "ms": v[1] * 1e3, "tflops": v[0] / v[1] / 1e12} for k, v in fam.items()}
        shapes = {}
        for name, fl, e0, e1, tag in trace:
            a = shapes.setdefault("%s %s" % (name, tag), [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        out["shapes"] = {k: {"launches": v[2], "ms": v[1] * 1e3, "tflops": v[0] / v[1] / 1e12} for k, v in shapes.items()}
    return out


def launch_class(name, tag):
    """The class of one MFMA launch of the instrumented step, from its trace tag (ops._traced): the table the roofline line breaks
    the step into (`roofline.by_class`)."""
    import re
    f = {k: int(v) for k, v in re.findall(r"([A-Za-z]+)(\d+)", tag)}
    if name == "k_attention":
        if f.get("D") == 32:
            return "attention EPA (D 32, sparse bias)"
        return "attention text (<= 128 keys)" if f.get("nk", 0) <= 128 else "attention self (D 64)"
    if name not in ("k_conv_gemm", "k_linear_ws"):
        return name
    M, K = f.get("M", 0) * max(f.get("b", 1), 1) if name == "k_conv_gemm" else f.get("M", 0), f.get("K", 0)
    if name == "k_conv_gemm" and f.get("k") == 3:
        return "conv 3x3, M >= 40960" if M >= 40960 else "conv 3x3, small M"
    if M <= 4096:
        return "linear / 1x1, small M (<= 4096)"
    return "linear / 1x1, K <= 640" if K <= 640 else "linear / 1x1, K 641-2000" if K <= 2000 else "linear / 1x1, K > 2000"


def usable_cores():
    """Host cores this process may actually use (affinity mask and cgroup CPU quota), capped at 64:
    the oracle's small-batch fp32 kernels do not scale past that."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"],
                    help="16-bit MFMA operand type.  fp16 (default) runs the MIXED scheme -- fp32 residual streams + "
                         "split-precision stream-path GEMMs -- the configuration that meets north_star's 1e-3 parity bar "
                         "(tests/test_gpu_mixed.py); bf16 runs the all-16-bit scheme of round 1 (1.2e-2), a labelled secondary line")
    ap.add_argument("--precision", default=None, choices=["mixed", "fast"], help="override the scheme of --dtype")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-training-leg", action="store_true",
                    help="skip the extra `training_step` measurement (N = 1 only; --no-cpu-baseline skips it too)")
    ap.add_argument("--trace-out", default=None, help="write the per-shape table of the instrumented step here")
    ap.add_argument("--small", action="store_true", help="reduced widths/sizes (debug only, not the metric)")
    ap.add_argument("--cfg5", action="store_true",
                    help="BASELINE.json configs[4]: layout-conditioned (panorama ControlNet, 512x1024 condition image) -- not the metric line")
    ap.add_argument("--cfg4", action="store_true",
                    help="BASELINE.json configs[3]: 1024x2048 panorama (128x256 latent) + 20x512^2 views -- not the metric line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from panfusion_amd import ops
    from panfusion_amd.models.sd2_unet_params import SD2_BASE
    from panfusion_amd.pipeline import DenoiseLoop
    from panfusion_amd.utils.pano import icosahedron_sample_camera
    import numpy as np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    # PF_DIST_BACKEND=gloo + fewer GPUs than ranks: functional dry run of the sharded path on one GPU
    backend = os.environ.get("PF_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    cfg = dict(SD2_BASE)
    m, lat_hw, pano_hw, flop = 20, (64, 64), (64, 128), FLOP_PER_STEP_CFG2
    workload = "cfg2: 512x1024 pano + 20x512^2 views, CFG pair, SD-2-base UNet shapes"
    if args.cfg5:
        flop = FLOP_PER_STEP_CFG2 + 2 * 0.66e12          # + ControlNet encoder + embedding per CFG sample
        workload = "cfg5: cfg2 + panorama ControlNet on a 512x1024 layout condition (NOT the headline config)"
    if args.cfg4:
        pano_hw, flop = (128, 256), 64.36e12
        workload = "cfg4: 1024x2048 pano + 20x512^2 views, CFG pair, SD-2-base UNet shapes (NOT the headline config)"
    if args.small:
        cfg.update(block_out_channels=(64, 128, 256, 256), num_heads=(1, 2, 4, 4), cross_attention_dim=128)
        lat_hw, pano_hw, flop, workload = (16, 16), (16, 32), float("nan"), "debug-small"
    th, ph = icosahedron_sample_camera()
    cams_deg = (np.degrees(th), np.degrees(ph))

    layout = None
    if args.cfg5:       # the layout condition image of the panorama ControlNet (rolled with the panorama every step, PanFusion.py:150-153)
        layout = (torch.rand(1, 1, 3, pano_hw[0] * 8, pano_hw[1] * 8, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(dev)
    if world > 1:
        from panfusion_amd import sharding
        model, loop = sharding.build_sharded(build_model, build_inputs, dev, dtype, cfg, m, lat_hw, pano_hw,
                                             cams_deg, args.steps + args.warmup + 1, not args.no_graphs,
                                             precision=args.precision, layout_cond=args.cfg5, layout=layout)
    else:
        model = build_model(dev, dtype, cfg, layout_cond=args.cfg5, precision=args.precision)
        inputs = build_inputs(dev, m, lat_hw, pano_hw, cfg["cross_attention_dim"], cams_deg)
        loop = DenoiseLoop(model, *inputs, steps=args.steps + args.warmup + 1, use_graphs=not args.no_graphs,
                           pano_layout_cond=layout)

    # PF_VIEW_PRIORITY=1 (experiment): the loop runs on a HIGH-priority stream; the panorama branch's side stream keeps the
    # default priority, so its kernels fill the chip only where the view branch (the critical path) leaves it idle
    import contextlib
    hp = torch.cuda.stream(torch.cuda.Stream(dev, priority=-1)) if os.environ.get("PF_VIEW_PRIORITY") else contextlib.nullcontext()
    with hp:
        loop.prepare()                                    # tables + one graph per rotation offset, untimed
        for _ in range(args.warmup):
            loop.step()
        if world > 1:
            dist.barrier()
            sharding.reset_comm()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.step()
        torch.cuda.synchronize()
    t_steps_done = time.perf_counter()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    distributed = None
    if world > 1:
        # what a SCALE record needs to check itself: the world this process group REALLY has, every rank's own time for the
        # timed steps (before / without the closing barrier), what each collective moved
        t_rank = torch.tensor([t_steps_done - t0], device=dev, dtype=torch.float64)
        all_t = [torch.zeros_like(t_rank) for _ in range(world)]
        dist.all_gather(all_t, t_rank)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in all_t]
        tt = torch.tensor([elapsed], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        distributed = {"backend": dist.get_backend(), "world_size_initialised": dist.get_world_size(),
                       "devices_visible": torch.cuda.device_count(),
                       "rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms), "per_rank": [round(v, 3) for v in per_rank_ms]},
                       "layout": getattr(loop, "layout_desc", None),
                       # hipGraph segments still in use after the run on EVERY rank (the fallback to eager launches is collective: a capture
                       # that fails anywhere drops the graphs everywhere), and whether the EPA collectives were posted asynchronously
                       "graphs_in_use": bool(getattr(loop, "use_graphs", False)), "async_collectives": bool(sharding.ASYNC),
                       "collectives_rank0": sharding.comm_stats(args.steps),
                       "note": "rank_ms = each rank's own wall time of the timed steps up to its final synchronize (no closing barrier); "
                               "collectives: issued per step on rank 0, bytes = what that rank contributes per call"}

    # ---- instrumented step: device events around every GEMM / attention launch ------------------
    trace = []
    serial = getattr(model, "two_streams", None)
    if serial is not None:
        model.two_streams = False                     # kernels one at a time: per-launch durations, not overlap
    ops.TRACE = trace
    torch.cuda._sleep(int(0.25 * 2.4e9))              # GPU busy ~0.25 s: the host enqueues the whole eager step
    loop.step_eager()                                 # ahead of it, so event pairs bracket kernels, not launch gaps
    torch.cuda.synchronize()
    ops.TRACE = None
    if serial is not None:
        model.two_streams = serial
    fam = {}
    shapes = {}
    classes = {}
    GEMM_FAMILY = ("k_conv_gemm", "k_linear_ws")      # the GEMM family: the tile kernels + the weight-stationary linear kernel
    for name, flops, e0, e1, tag in trace:
        sec = e0.elapsed_time(e1) * 1e-3
        f = fam.setdefault("k_conv_gemm" if name in GEMM_FAMILY else name, [0.0, 0.0, 0])
        f[0] += flops
        f[1] += sec
        f[2] += 1
        g = shapes.setdefault(name + " " + tag, [0.0, 0.0, 0])
        g[0] += flops
        g[1] += sec
        g[2] += 1
        c = classes.setdefault(launch_class(name, tag), [0.0, 0.0, 0, 0.0])
        c[0] += flops
        c[1] += sec
        c[2] += 1
        c[3] += sec if name == "k_linear_ws" else 0.0
    if args.trace_out and rank == 0:
        with open(args.trace_out, "w") as fh:
            fh.write("classes of the instrumented (eager, single-stream) step: launches, ms, TF/s, fraction of the 2.5 PF dense peak\n")
            for k, (fl, sec, n, _) in sorted(classes.items(), key=lambda kv: -kv[1][1]):
                fh.write("  %-44s launches %3d  ms %8.3f  TF/s %7.1f  frac %.3f\n" % (k, n, sec * 1e3, fl / sec / 1e12, fl / sec / 1e12 / PEAK_BF16_DENSE_TFLOPS))
            fh.write("\n")
            for k, (fl, sec, n) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                fh.write("%-60s launches %3d  ms %8.3f  TF/s %7.1f\n" % (k, n, sec * 1e3, fl / sec / 1e12))
    dom = max(fam, key=lambda k: fam[k][1]) if fam else None
    roofline = None
    # Committed rocprofv3 summaries this line quotes (profiles/, produced by tools/gpu_run.sh <tag> pmc / serial / mfma on the same command): each
    # carries the sha256 of the kernel sources it was measured on; one measured on OTHER sources is not quoted -- the field is null and the
    # note says why.
    from panfusion_amd import _lib as _pf_lib
    src_hash = _pf_lib.source_hash()
    stale = []

    def committed(names, parse):
        for name in names:
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    doc, h = parse(fh.read())
            except (OSError, ValueError, KeyError, IndexError):
                continue
            if h != src_hash:
                stale.append("profiles/%s was measured on kernel sources %s, this build is %s" % (name, h or "(no hash recorded)", src_hash))
                return None, name
            return doc, name
        return None, None

    def parse_json(text):
        doc = json.loads(text)
        return doc, doc.get("csrc_sha256")

    def parse_hashed_text(text):
        first = text.split("\n", 1)[0]
        return text, first.split(":", 1)[1].strip() if first.startswith("csrc_sha256:") else None

    traffic, traffic_file = committed(("r6_traffic.json", "r5_traffic.json"), parse_json)
    instep, instep_file = committed(("r6_final_mfma_instep.txt", "r5_final_mfma_instep.txt"), parse_hashed_text)
    steady, steady_file = committed(("r6_final_kernels_steady.txt", "r5_final_kernels_steady.txt"), parse_hashed_text)
    import re as _re
    mfma_busy = None
    if instep:
        ma = _re.search(r"attention, all launches, cycle-weighted: MFMA busy ([0-9.]+) %", instep)
        mg = _re.search(r"GEMM family .* cycle-weighted: MFMA busy ([0-9.]+) %", instep)
        mfma_busy = {"attention": float(ma.group(1)) / 100 if ma else None, "gemm_family": float(mg.group(1)) / 100 if mg else None,
                     "source": "SQ_VALU_MFMA_BUSY_CYCLES over every dispatch of one eager step, profiles/%s" % instep_file}
    tail = None
    if steady:
        mt = _re.search(r"non-MFMA tail .*: ([0-9.]+) ms and ([0-9.]+) launches per step", steady)
        if mt:
            tail = (float(mt.group(1)) * 1e-3, float(mt.group(2)))
    if dom:
        fl, sec, n = fam[dom]
        roofline = {"bound": "mfma", "kernel": "k_conv_gemm+k_linear_ws" if dom == "k_conv_gemm" else dom, "achieved": fl / sec / 1e12, "peak": PEAK_BF16_DENSE_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / sec / 1e12 / PEAK_BF16_DENSE_TFLOPS,
                    "traffic": traffic["bytes_per_launch"] if traffic and dom in traffic.get("kernel", "") else None,
                    "traffic_note": ("bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, L2<->fabric incl. Infinity-Cache hits: an upper bound on HBM bytes) from the "
                                     "rocprofv3 PMC passes committed in profiles/%s (not re-measured in this run)" % traffic_file) if traffic
                                    else "no committed PMC summary matches this build: " + ("; ".join(stale) or "none found"),
                    # achieved L2<->fabric bandwidth against the 8 TB/s HBM spec (north_star: rocprof HBM GB/s beside MFMA utilisation):
                    # the family's PMC bytes per launch over ITS time in this run; the non-MFMA tail's PMC bytes over its time in the
                    # committed steady-state kernel table
                    "hbm": {"peak_tbps": 8.0,
                            "family_tbps": traffic["bytes_per_launch"] * n / sec / 1e12 if traffic else None,
                            "family_frac": traffic["bytes_per_launch"] * n / sec / 8e12 if traffic else None,
                            "tail_tbps": traffic["tail_bytes_per_launch"] * tail[1] / tail[0] / 1e12 if traffic and tail and traffic.get("tail_bytes_per_launch") else None,
                            "tail_ms_per_step": tail[0] * 1e3 if tail else None, "tail_launches_per_step": tail[1] if tail else None,
                            "sources": [f for f in (traffic_file and "profiles/" + traffic_file, steady_file and "profiles/" + steady_file) if f]},
                    "mfma_busy": mfma_busy,
                    "attention_mfma_busy": mfma_busy["attention"] if mfma_busy else None,
                    # what the power budget leaves of the 2.5 PF dense peak on N(0,1) fp16 operands when a kernel issues NOTHING but MFMAs
                    # (tools/ubench/mfma_power.hip, profiles/r6_gemm32_power_cap.txt): the ceiling the family is really priced against
                    "peak_power_capped": {"v_mfma_f32_16x16x32_f16": 1867.0, "v_mfma_f32_32x32x16_f16": 1645.0, "zero_operands": 2483.0, "unit": "TFLOP/s",
                                          "frac_of_16x16x32_cap": fl / sec / 1e12 / 1867.0, "source": "profiles/r6_gemm32_power_cap.txt section 1"},
                    "stale_profiles": stale or None,
                    "launches_per_step": n, "avg_launch_us": sec / n * 1e6, "algorithmic_flop_per_launch": fl / n,
                    "share_of_step_time": sec / (elapsed / args.steps),
                    "family": "k_conv_gemm (implicit-GEMM tile kernels) + k_linear_ws (weight-stationary linear, C = 320 layers)",
                    "by_class": {k: {"launches": v[2], "ms": v[1] * 1e3, "tflops": v[0] / v[1] / 1e12, "frac": v[0] / v[1] / 1e12 / PEAK_BF16_DENSE_TFLOPS,
                                     **({"ms_in_k_linear_ws": v[3] * 1e3} if v[3] else {})}
                                 for k, v in sorted(classes.items(), key=lambda kv: -kv[1][1])},
                    "other": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] * 1e3, "launches": v[2]}
                              for k, v in fam.items() if k != dom}}

    if rank == 0:
        metric = ("DDIM denoise steps/sec, 1024x2048 pano + 20x512^2 views (configs[3])" if args.cfg4
                  else "DDIM denoise steps/sec, layout-conditioned 512x1024 pano + 20x512^2 views (configs[4])" if args.cfg5
                  else "DDIM denoise steps/sec, 512x1024 pano + 20x512^2 views")
        res = {"metric": metric, "value": args.steps / elapsed,
               "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (seeded inputs, random-init SD-2-base-shaped weights)",
               "config": {"workload": workload, "views": m, "view_latent": list(lat_hw), "pano_latent": list(pano_hw),
                          "cfg_pair": True, "hip_graphs": not args.no_graphs,
                          "precision": "%s operands, %s" % (args.dtype, {
                              "mixed": "mixed scheme: fp32 residual streams + split-precision stream-path GEMMs (<= 1e-3 rel-L2 vs the fp32 oracle)",
                              "fast": "all activations 16-bit (parity 1.7e-3 fp16 / 1.2e-2 bf16: does NOT meet the 1e-3 bar)"}[model.precision]),
                          "parallelism": "single" if world == 1 else getattr(loop, "layout_desc", "sharded")},
               "tflops_per_step": flop / 1e12, "frac_of_mfma_ceiling": (flop * args.steps / elapsed) / (PEAK_BF16_DENSE_TFLOPS * 1e12),
               "roofline": roofline}
        if distributed is not None:
            res["distributed"] = distributed
        if not args.no_cpu_baseline and world == 1 and not args.small:
            res["cpu_baseline"] = cpu_baseline(cfg, cfg["cross_attention_dim"], lat_hw, pano_hw, m, cams_deg, flop)
        # (skipped with --no-cpu-baseline as well: that flag marks the quick / profiled runs of tools/*.sh -- under
        # rocprofv3 --pmc the ~6000 launches of a training step take tens of minutes)
        if not args.no_training_leg and not args.no_cpu_baseline and world == 1 and not (args.small or args.cfg4 or args.cfg5):
            # extra, after the metric and its baseline: the training step through the same boundary
            try:
                del loop, model
                torch.cuda.empty_cache()
                res["training_step"] = training_step_leg(dev, dtype, cfg, args.precision)
            except Exception as exc:                     # never lose the metric line to the extra
                res["training_step"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
