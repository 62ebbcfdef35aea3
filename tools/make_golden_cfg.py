"""Golden outputs of the CPU oracle at the EXACT BASELINE.json configurations (VERDICT r2 "Next round" item 1).

Runs in the build container (CPU, minutes per case); the ``-m gpu`` tests of tests/test_gpu_configs.py rebuild the
same seeded weights / inputs on the GPU box (oracle/fixtures.py) and compare the HIP path with what is stored here.

    python tools/make_golden_cfg.py cfg2 cfg1 cfg4 cfg5        # any subset

  cfg2  -> tests/golden/cfg2_eps.npz       m = 20 icosahedron views x CFG pair, first-step denoiser call at SD-2-base
                                           widths: both epsilon outputs + 8-channel slices of both streams after each of
                                           the 7 EPA blocks (the 20-view EPA softmax has K = 20 480 keys)
  cfg1  -> tests/golden/cfg1_ddim10.npz    configs[0]: m = 4, 256^2 views, 10 DDIM steps at full widths: latents after
                                           every step (drift per step).  Round 5: the denoiser of this fixture is the REFERENCE's
                                           own class (models/pano/MVGenModel.py imported from /root/reference, third-party modules
                                           shimmed: oracle/fixtures.reference_denoiser), not the port -- PF_GOLDEN_PORT=1 restores the port
  cfg2b -> tests/golden/cfg2b_eps.npz      the same configuration at a LATE step of the loop: timestep t = 21, accumulated rotation 180
                                           degrees (m = 20 x CFG pair; both epsilon outputs)
  cfg1s -> tests/golden/cfg1_stress_eps.npz  RANGE STRESS: cfg1's first denoiser call (CFG pair) with the stream-writing layers' output
                                           channels scaled log-uniformly over 3 decades (oracle/fixtures.apply_range_stress): both epsilon
                                           outputs + the largest |value| of every residual-stream tensor (reference class as denoiser)
  cfg4  -> tests/golden/cfg4_eps.npz       128x256 panorama latent + 20 views, the CFG pair (b = 2) as the loop calls it
  cfg5  -> tests/golden/cfg5_eps.npz       cfg2's geometry + panorama ControlNet on a 512x1024 layout image, the CFG pair (b = 2)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ddim as oddim  # noqa: E402
from oracle import fixtures as FX  # noqa: E402
from oracle import mvgen as MV  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
NSLICE = 8


def epa_slices(model):
    """Forward hooks on the 7 EPA blocks: channels ::C/8 of sample -1 / view 0 of both outputs."""
    got, hooks = {}, []
    blocks = [*model.cp_blocks_encoder, model.cp_blocks_mid, *model.cp_blocks_decoder]

    def make(i):
        def hook(mod, args, out):
            p, e = out
            st = p.shape[1] // NSLICE
            m = p.shape[0] // e.shape[0]
            got["epa%d_pers" % i] = p[-m, ::st].numpy().copy()          # view 0 of the last (conditional) sample
            got["epa%d_pano" % i] = e[-1, ::st].numpy().copy()
        return hook
    for i, blk in enumerate(blocks):
        hooks.append(blk.register_forward_hook(make(i)))
    return got, hooks


def call(model, args, **extra):
    with torch.no_grad(), FX.chunked_attention():
        return model(args["latents"], args["pano_latent"], args["timestep"], args["prompt_embd"],
                     args["pano_prompt_embd"], args["cameras"], **extra)


def cfg2():
    model = FX.build_full_width()
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=True)
    got, hooks = epa_slices(model)
    t0 = time.time()
    s, ps = call(model, args)
    print("cfg2 oracle forward %.0f s" % (time.time() - t0), flush=True)
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, "cfg2_eps.npz"), sample=s.numpy(), pano_sample=ps.numpy(), **got)


def cfg2ref():
    """cfg 2's first denoiser call, CONDITIONAL half of the CFG pair (the samples of a batch are independent), through the REFERENCE's own
    class -- models/pano/MVGenModel.py:38-297 with its WarpAttn / get_masks / CrossAttention (dense per-head bias: 20 heads x 2048 x 20480
    fp32 = 3.4 GB per direction at C = 640) -- around the restated UNet objects (the 20-view / 20 480-key EPA path of
    the reference code had never produced a golden file).  Stored next to the port-generated cfg2_eps.npz together with the measured
    port-vs-reference difference on this half; tests/test_oracle_vs_reference.py asserts the two files agree to 1e-5, the GPU test compares
    the HIP path with this one too."""
    om = FX.build_full_width()
    rm = FX.reference_denoiser(om)
    print("cfg2ref: denoiser =", type(rm).__module__, type(rm).__name__, flush=True)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=False)
    t0 = time.time()
    s, ps = call(rm, args)
    print("cfg2ref reference forward (one CFG half) %.0f s" % (time.time() - t0), flush=True)
    old = np.load(os.path.join(OUT, "cfg2_eps.npz"))
    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
    d_v, d_p = rel(old["sample"][1:], s.numpy()), rel(old["pano_sample"][1:], ps.numpy())
    print("cfg2ref: port-generated cfg2_eps.npz (conditional half) vs the reference class: %.2e views / %.2e panorama" % (d_v, d_p), flush=True)
    np.savez_compressed(os.path.join(OUT, "cfg2_ref_cond.npz"), sample=s.numpy(), pano_sample=ps.numpy(),
                        port_vs_reference=np.array([d_v, d_p]))


def _ref_cond(name, fixture, out, pano_hw, controlnet=False):
    """Conditional half of a configuration's first denoiser call through the REFERENCE's own class (see cfg2ref), compared with the
    conditional half of the port-generated fixture and stored next to it."""
    om = FX.build_full_width(controlnet=controlnet)
    rm = FX.reference_denoiser(om)
    print("%s: denoiser =" % name, type(rm).__module__, type(rm).__name__, flush=True)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), pano_hw, cfg_pair=False)
    extra = {}
    if controlnet:
        extra["pano_layout_cond"] = torch.roll(FX.layout_image(pano_hw), pano_hw[1] * 8 // 4, dims=-1)
    t0 = time.time()
    s, ps = call(rm, args, **extra)
    print("%s reference forward (one CFG half) %.0f s" % (name, time.time() - t0), flush=True)
    old = np.load(os.path.join(OUT, fixture))
    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
    d_v, d_p = rel(old["sample"][1:], s.numpy()), rel(old["pano_sample"][1:], ps.numpy())
    print("%s: port-generated %s (conditional half) vs the reference class: %.2e views / %.2e panorama" % (name, fixture, d_v, d_p), flush=True)
    np.savez_compressed(os.path.join(OUT, out), sample=s.numpy(), pano_sample=ps.numpy(), port_vs_reference=np.array([d_v, d_p]))


def cfg2bref():
    """cfg 2's LATE call (t = 21, accumulated rotation 180 degrees) through the reference class, conditional half."""
    om = FX.build_full_width()
    rm = FX.reference_denoiser(om)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=False, t=21, rot=180.0)
    t0 = time.time()
    s, ps = call(rm, args)
    print("cfg2bref reference forward (one CFG half) %.0f s" % (time.time() - t0), flush=True)
    old = np.load(os.path.join(OUT, "cfg2b_eps.npz"))
    rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
    d_v, d_p = rel(old["sample"][1:], s.numpy()), rel(old["pano_sample"][1:], ps.numpy())
    print("cfg2bref: port-generated cfg2b_eps.npz (conditional half) vs the reference class: %.2e views / %.2e panorama" % (d_v, d_p), flush=True)
    np.savez_compressed(os.path.join(OUT, "cfg2b_ref_cond.npz"), sample=s.numpy(), pano_sample=ps.numpy(), port_vs_reference=np.array([d_v, d_p]))


def cfg4ref():
    """configs[3] (128 x 256 panorama latent): the reference class's dense bias is 10 heads x 8192 x 20480 fp32 = 6.7 GB per direction at C = 320."""
    _ref_cond("cfg4ref", "cfg4_eps.npz", "cfg4_ref_cond.npz", (128, 256))


def cfg5ref():
    """configs[4] (panorama ControlNet): MVGenModel.py:68-83,154-170,200-203 of the reference class around the restated ControlNet."""
    _ref_cond("cfg5ref", "cfg5_eps.npz", "cfg5_ref_cond.npz", (64, 128), controlnet=True)


def cfg2b():
    model = FX.build_full_width()
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=True, t=21, rot=180.0)
    t0 = time.time()
    s, ps = call(model, args)
    print("cfg2b oracle forward %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "cfg2b_eps.npz"), sample=s.numpy(), pano_sample=ps.numpy())


def cfg1():
    model = FX.build_full_width()
    if os.environ.get("PF_GOLDEN_PORT", "0") != "1":
        model = FX.reference_denoiser(model)          # the reference class itself drives the fixture
        print("cfg1: denoiser =", type(model).__module__, type(model).__name__, flush=True)
    cams = FX.horizon4_cameras()
    latents, pano_latent, pe, ppe = FX.loop_inputs(cams, (32, 32), (64, 128))
    sched = oddim.DDIM()
    traj_v, traj_p = [], []
    total = 0.0
    t0 = time.time()
    with FX.chunked_attention():
        for t in sched.set_timesteps(10):
            latents, pano_latent, cams = oddim.denoise_step(model, sched, t, latents, pano_latent, pe, ppe, cams)
            total += 90.0
            traj_v.append(latents.numpy().copy())
            traj_p.append(oddim.rotate_latent(pano_latent, cams, -total)[0].numpy().copy())     # un-rotated frame
            print("cfg1 step t=%d  %.0f s" % (int(t), time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "cfg1_ddim10.npz"), latents=np.stack(traj_v), pano_latent=np.stack(traj_p))


def cfg1s():
    om = FX.build_full_width()
    FX.apply_range_stress(om)
    model = FX.reference_denoiser(om) if os.environ.get("PF_GOLDEN_PORT", "0") != "1" else om
    args = FX.first_step_call(FX.horizon4_cameras(), (32, 32), (64, 128), cfg_pair=True)
    peaks = {}

    def hook(name):
        def fn(mod, a, out):
            out = getattr(out, "sample", out)
            out = out[0] if isinstance(out, (tuple, list)) else out
            peaks[name] = max(peaks.get(name, 0.0), float(out.abs().max()))
        return fn
    hooks = []
    for tag, unet in (("views", om.unet), ("pano", om.pano_unet)):
        hooks.append(unet.conv_in.register_forward_hook(hook(tag + ".conv_in")))
        for name, mod in unet.named_modules():
            if type(mod).__name__ in ("ResnetBlock2D", "Transformer2DModel"):
                hooks.append(mod.register_forward_hook(hook(tag + "." + name)))
    t0 = time.time()
    s, ps = call(model, args)
    print("cfg1s oracle forward %.0f s" % (time.time() - t0), flush=True)
    for h in hooks:
        h.remove()
    names = sorted(peaks)
    top = sorted(peaks.items(), key=lambda kv: -kv[1])[:5]
    print("cfg1s: largest |stream| values:", ", ".join("%s %.3g" % kv for kv in top), flush=True)
    print("cfg1s: eps finite:", bool(torch.isfinite(s).all() and torch.isfinite(ps).all()), " |eps| rms %.3g / %.3g" % (float(s.pow(2).mean().sqrt()), float(ps.pow(2).mean().sqrt())))
    np.savez_compressed(os.path.join(OUT, "cfg1_stress_eps.npz"), sample=s.numpy(), pano_sample=ps.numpy(),
                        stream_names=np.array(names), stream_peaks=np.array([peaks[n] for n in names], dtype=np.float32))


def cfg4():
    model = FX.build_full_width()
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (128, 256), cfg_pair=True)
    t0 = time.time()
    s, ps = call(model, args)
    print("cfg4 oracle forward %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "cfg4_eps.npz"), sample=s.numpy(), pano_sample=ps.numpy())


def cfg5():
    model = FX.build_full_width(controlnet=True)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=True)
    cond = torch.roll(FX.layout_image((64, 128)), 1024 // 4, dims=-1)       # rolled with the panorama (PanFusion.py:152-153)
    cond = torch.cat([cond] * 2)                                            # gen_cls_free_guide_pair duplicates it (PanoGenerator.py:240-251)
    t0 = time.time()
    s, ps = call(model, args, pano_layout_cond=cond)
    print("cfg5 oracle forward %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "cfg5_eps.npz"), sample=s.numpy(), pano_sample=ps.numpy())


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("PF_THREADS", os.cpu_count() or 8)))
    for name in (sys.argv[1:] or ["cfg2", "cfg2b", "cfg1", "cfg4", "cfg5"]):
        globals()[name]()
        print(name, "done", flush=True)
