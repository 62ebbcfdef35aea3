"""Seeded builders for the EXACT BASELINE.json configurations (SURVEY.md §8d "Synthetic inputs").

TEST INFRASTRUCTURE (see oracle/__init__.py) -- shared by tools/make_golden_cfg.py (which runs the oracle
at full widths in the build container and commits the outputs under tests/golden/) and by the ``-m gpu`` parity
tests (which rebuild the same weights / inputs from the same seeds on the GPU box and compare the HIP path with
the committed outputs).  Everything is drawn from CPU generators: identical on every machine.

What a configuration is (reference call sites):
  * cfg 2 -- the headline: one prompt, m = 20 icosahedron views of 64x64 latents + a 64x128 panorama latent, the
    CFG pair (b = 2), first DDIM timestep, cameras rotated by the loop's first 90 degrees
    (models/pano/PanFusion.py:146-162; MVGenModel.py:38-297);
  * cfg 1 -- configs[0]: m = 4 views of 256^2 (32x32 latents), 10 DDIM steps, SD-2 widths, CPU-runnable;
  * cfg 4 -- configs[3]: 128x256 panorama latent + 20 views (one CFG sample here: the call is what is pinned);
  * cfg 5 -- configs[4]: cfg 2's geometry + the panorama ControlNet on a 512x1024 layout image
    (PanoGenerator.py:153-157, MVGenModel.py:68-83) (one CFG sample).
"""
import numpy as np
import torch

from . import ddim as oddim
from . import geometry as G
from . import mvgen as MV
from . import sd2_unet as U

SEEDS = dict(unet=101, pano_unet=102, enc=103, mid=104, dec=105, epa=106, cn=111)


def build_full_width(controlnet=False, cfg=None):
    """SD-2-base widths, LoRA rank 4 attached, EPA output projections re-randomised -- the construction of
    tests/test_gpu_mixed.py::full_width (same seeds), optionally with the panorama ControlNet."""
    torch.manual_seed(0)
    cfg = dict(cfg or U.SD2_BASE)
    unet, pano_unet = U.UNet2DConditionModel(**cfg), U.UNet2DConditionModel(**cfg)
    unet.add_lora(4)
    pano_unet.add_lora(4)
    U.init_synthetic(unet, SEEDS["unet"])
    U.init_synthetic(pano_unet, SEEDS["pano_unet"])
    pano_cn = None
    if controlnet:
        pano_cn = U.ControlNetModel.from_unet(pano_unet)
        U.init_synthetic(pano_cn.controlnet_cond_embedding, SEEDS["cn"])
        U.init_synthetic(pano_cn.controlnet_down_blocks, SEEDS["cn"] + 1)      # zero-initialised in diffusers: would be a no-op
        U.init_synthetic(pano_cn.controlnet_mid_block, SEEDS["cn"] + 2)
    om = MV.DualBranchDenoiser(unet, pano_unet, None, pano_cn, True)
    U.init_synthetic(om.cp_blocks_encoder, SEEDS["enc"])
    U.init_synthetic(om.cp_blocks_mid, SEEDS["mid"])
    U.init_synthetic(om.cp_blocks_decoder, SEEDS["dec"])
    MV.randomize_epa(om, SEEDS["epa"])
    return om.eval()


def reference_denoiser(om):
    """The REFERENCE's own ``models.pano.MVGenModel.MultiViewBaseModel`` (imported from /root/reference by oracle/ref_import.py,
    third-party modules shimmed) around the same UNet objects and EPA weights as the port ``om`` -- what
    tools/make_golden_cfg.py drives for the cfg 1 fixture and tests/test_oracle_vs_reference.py pins the port against at
    SD-2-base widths.  Build container only (the GPU box has no /root/reference)."""
    from . import ref_import
    ref = ref_import.load()
    rm = ref.MultiViewBaseModel(om.unet, om.pano_unet, None, om.pano_cn, True).eval()
    res = rm.load_state_dict({k: v for k, v in om.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if k.startswith("cp_blocks")], res
    return rm


STRESS_LAYERS = ("conv2", "proj_out")


def apply_range_stress(om, seed=7, decades=3.0):
    """Range-stress variant of the synthetic weights: every layer that writes a NORMALISED activation INTO a
    residual stream -- conv_in, ResnetBlock2D.conv2, Transformer2DModel.proj_out of both UNets and the EPA blocks' attention /
    feed-forward output projections (the stream -> stream maps conv_shortcut / Downsample2D / Upsample2D keep their gain: scaled too,
    the range would compound by 10^decades per layer -- 7e22 after the decoder, fp32 itself overflows) -- gets its output channels (weight rows and bias) multiplied by
    a per-channel scale drawn log-uniformly over ``decades`` decades, 1 ... 10^decades, ONE vector per stream width (the same
    channel is the outlier in every layer of that width, as in trained SD-2 weights).  The fan-in-scaled Gaussians of
    init_synthetic keep every stream at O(1); with this the stream channels reach 1e3 ... 1e4 and a GroupNorm group is dominated
    by its largest channel -- what fp16 operands (5 exponent bits) have to survive.  In place; returns the scale vectors."""
    scales = {}

    def vec(c):
        if c not in scales:
            g = torch.Generator().manual_seed(seed * 100003 + c)
            scales[c] = 10.0 ** (torch.rand(c, generator=g) * decades)
        return scales[c]

    def scale_out(mod):
        s = vec(mod.weight.shape[0])
        with torch.no_grad():
            mod.weight.mul_(s.reshape(-1, *([1] * (mod.weight.dim() - 1))))
            if mod.bias is not None:
                mod.bias.mul_(s)
    for unet in (om.unet, om.pano_unet):
        for name, mod in unet.named_modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)) and (name == "conv_in" or name.endswith(STRESS_LAYERS)):
                scale_out(mod)
    for blk in [*om.cp_blocks_encoder, om.cp_blocks_mid, *om.cp_blocks_decoder]:
        scale_out(blk.transformer.attn1.to_out)
        scale_out(blk.transformer.ff.net[2])
    return scales


def ico_cameras(b=1):
    th, ph = G.icosahedron_cameras()
    th, ph = np.degrees(th), np.degrees(ph)
    m = len(th)
    return {"FoV": torch.full((b, m), 90), "theta": torch.tensor(th, dtype=torch.float64)[None].repeat(b, 1),
            "phi": torch.tensor(ph, dtype=torch.float64)[None].repeat(b, 1)}


def horizon4_cameras(b=1):
    """cfg 1: theta = 0, 90, 180, 270, phi = 0 (supplied directly, the dataset cannot emit 4 views)."""
    return {"FoV": torch.full((b, 4), 90), "theta": torch.tensor([[0.0, 90.0, 180.0, 270.0]], dtype=torch.float64).repeat(b, 1),
            "phi": torch.zeros(b, 4, dtype=torch.float64)}


def loop_inputs(cameras, lat_hw, pano_hw, ctx_dim=1024):
    """panorama noise seed 0, view noise = e2p(nearest) of it, prompts seeds 1 / 2, null prompt seed 3.
    Returns (latents (1,m,4,h,w), pano_latent (1,1,4,H,W), prompt_embd (2,m,77,D), pano_prompt_embd (2,1,77,D)) --
    the prompt tensors already hold [null ; prompt] along the batch as PanFusion.inference builds them (PanFusion.py:134-138)."""
    g = lambda s: torch.Generator().manual_seed(s)
    m = cameras["FoV"].shape[1]
    pano_noise = torch.randn(1, 1, 4, *pano_hw, generator=g(0))
    _, latents = oddim.init_noise(pano_noise, cameras, *lat_hw)
    prompt = torch.randn(1, m, 77, ctx_dim, generator=g(1))
    pano_prompt = torch.randn(1, 1, 77, ctx_dim, generator=g(2))
    null = torch.randn(1, 1, 77, ctx_dim, generator=g(3))
    return latents, pano_noise, torch.cat([null.expand(-1, m, -1, -1), prompt]), torch.cat([null, pano_prompt])


def layout_image(pano_hw):
    return torch.rand(1, 1, 3, pano_hw[0] * 8, pano_hw[1] * 8, generator=torch.Generator().manual_seed(4)) * 2 - 1


def first_step_call(cameras, lat_hw, pano_hw, cfg_pair=True, t=981, rot=90.0):
    """Arguments of the denoiser call of the FIRST loop iteration (PanFusion.py:149-158): panorama latent rolled by
    ``rot``, cameras rotated with it, everything CFG-paired (or the conditional sample alone)."""
    latents, pano_latent, pe, ppe = loop_inputs(cameras, lat_hw, pano_hw)
    pano_latent, cams = oddim.rotate_latent(pano_latent, cameras, rot)
    m = latents.shape[1]
    if cfg_pair:
        args = dict(latents=oddim.cfg_pair(latents), pano_latent=oddim.cfg_pair(pano_latent),
                    timestep=torch.full((2, m), t, dtype=torch.long), prompt_embd=pe, pano_prompt_embd=ppe,
                    cameras=oddim.cfg_pair(cams))
    else:
        args = dict(latents=latents, pano_latent=pano_latent, timestep=torch.full((1, m), t, dtype=torch.long),
                    prompt_embd=pe[1:], pano_prompt_embd=ppe[1:], cameras=cams)
    return args


def chunked_attention():
    """Context manager: the oracle's N x N-materialising attentions evaluated in chunks of <= 2 GiB of scores -- the
    same arithmetic per (sample, head) matrix, just not 27 GB at once (40 views x 5 heads x 4096^2 at cfg 2;
    5 heads x 32768^2 and 20 heads x 8192 x 20480 at cfg 4)."""
    import contextlib
    LIMIT = 2 << 30

    @contextlib.contextmanager
    def ctx():
        saved, saved_epa = U.Attention.forward, MV._BiasedCrossAttention.forward

        def forward(self, x, context=None):
            ctxt = x if context is None else context
            n, nk = x.shape[1], ctxt.shape[1]
            per = self.heads * n * nk * 4                      # bytes of scores per sample
            if per * x.shape[0] <= LIMIT:
                return saved(self, x, context)
            if per <= LIMIT:
                step = max(1, LIMIT // per)
                return torch.cat([saved(self, x[i:i + step], None if context is None else context[i:i + step])
                                  for i in range(0, x.shape[0], step)])
            b, h = x.shape[0], self.heads                      # one sample is too large already: per head, 8192 query rows
            q, k, v = self.to_q(x), self.to_k(ctxt), self.to_v(ctxt)
            d = q.shape[-1] // h
            out = torch.empty_like(q)
            for i in range(b):
                for j in range(h):
                    sl = slice(j * d, (j + 1) * d)
                    for r in range(0, n, 8192):
                        p = (q[i, r:r + 8192, sl] @ k[i, :, sl].T * self.scale).softmax(dim=-1)
                        out[i, r:r + 8192, sl] = p @ v[i, :, sl]
            return self.to_out[1](self.to_out[0](out))

        def forward_epa(self, x, context, bias):
            b, n, _ = x.shape
            nk, h = context.shape[1], self.heads
            if b * h * n * nk * 4 <= LIMIT:
                return saved_epa(self, x, context, bias)
            q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
            d = q.shape[-1] // h
            out = torch.empty_like(q)
            for i in range(b):
                for j in range(h):
                    sl = slice(j * d, (j + 1) * d)
                    p = (q[i, :, sl] @ k[i, :, sl].T * d ** -0.5 + bias[i]).softmax(dim=-1)
                    out[i, :, sl] = p @ v[i, :, sl]
            return self.to_out(out)
        U.Attention.forward, MV._BiasedCrossAttention.forward = forward, forward_epa
        try:
            yield
        finally:
            U.Attention.forward, MV._BiasedCrossAttention.forward = saved, saved_epa
    return ctx()
