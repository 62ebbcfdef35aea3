"""Host logic of the product (weight packing, layer sequencing, pad/crop/skip bookkeeping, sampling
loop) run end to end on the CPU with the test double tests/fake_ops.py standing in for the HIP
front end, and compared with the oracle.  fp32 throughout, so agreement is at fp32 round-off."""
import pytest
import torch

import fake_ops
from conftest import build_tiny_oracle, cam4, golden, rel_l2
from oracle import mvgen as MV

MODS = ["panfusion_amd.engine", "panfusion_amd.pipeline", "panfusion_amd.vae", "panfusion_amd.text_encoder", "panfusion_amd.models.pano.modules",
        "panfusion_amd.models.pano.utils", "panfusion_amd.utils.pano",
        "panfusion_amd.external.Perspective_and_Equirectangular.e2p",
        "panfusion_amd.external.Perspective_and_Equirectangular.p2e"]


@pytest.fixture
def fake_backend(monkeypatch):
    import importlib
    for name in MODS:
        monkeypatch.setattr(importlib.import_module(name), "ops", fake_ops)


def hip_model(oracle_model, precision="fast", dtype=torch.float32):
    from panfusion_amd.models.pano import MultiViewBaseModel
    m = MultiViewBaseModel(oracle_model.unet, oracle_model.pano_unet, None, None, oracle_model.pano_pad,
                           compute_dtype=dtype, precision=precision)
    if oracle_model.unet is not None:
        m.load_state_dict({k: v for k, v in oracle_model.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    return m


@pytest.fixture(scope="module")
def oracle_model():
    return build_tiny_oracle()


def test_denoiser_sequencing_matches_reference_golden(fake_backend, oracle_model):
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    cams = {k: torch.stack([v, v]) for k, v in cam4().items()}
    s, ps = hip_model(oracle_model)(t("latents"), t("pano_latent"), torch.full((2, 4), 981), t("prompt_embd"),
                                    t("pano_prompt_embd"), cams)
    assert rel_l2(s, t("sample")) < 2e-5 and rel_l2(ps, t("pano_sample")) < 2e-5


def test_mixed_scheme_host_logic_and_emulated_error(fake_backend, oracle_model):
    """The mixed scheme's HOST logic (fp32 stream plumbing, [hi | lo] operands, [W_hi | W_hi | W_lo] weights, the
    kernel's two-source K walk) on the CPU test double: with fp32 "16-bit" types it is the oracle to round-off; with
    fp16 it emulates what the GPU path stores where and must sit under north_star's 1e-3 -- and well under the
    all-16-bit scheme (profiles/archive/r2_precision_budget.txt: 1.7e-3 -> 6.8e-4 at SD-2-base widths)."""
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    cams = {k: torch.stack([v, v]) for k, v in cam4().items()}
    args = (t("latents"), t("pano_latent"), torch.full((2, 4), 981), t("prompt_embd"), t("pano_prompt_embd"), cams)
    s, ps = hip_model(oracle_model, "mixed")(*args)
    assert rel_l2(s, t("sample")) < 2e-5 and rel_l2(ps, t("pano_sample")) < 2e-5
    err = {}
    for prec in ("fast", "mixed"):
        s, ps = hip_model(oracle_model, prec, torch.float16)(*args)
        err[prec] = max(rel_l2(s, t("sample")), rel_l2(ps, t("pano_sample")))
    print("emulated fp16 rel-L2: fast %.3e, mixed %.3e" % (err["fast"], err["mixed"]))
    assert err["mixed"] <= 1e-3 and err["mixed"] < 0.6 * err["fast"]


def test_state_dict_loaded_through_a_parent_module_repacks(fake_backend, oracle_model):
    """nn.Module.load_state_dict recurses through _load_from_state_dict, so an overridden
    load_state_dict of a child never runs when a checkpoint is loaded through a parent (the reference's
    LightningModule); the packed weights must be dropped anyway."""
    import copy
    m = hip_model(oracle_model)
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    cams = {k: torch.stack([v, v]) for k, v in cam4().items()}
    args = (t("latents"), t("pano_latent"), torch.full((2, 4), 981), t("prompt_embd"), t("pano_prompt_embd"), cams)
    before = m(*args)[0]

    class Parent(torch.nn.Module):
        def __init__(self, child):
            super().__init__()
            self.mv_base_model = child
    parent = Parent(m)
    original = copy.deepcopy(parent.state_dict())
    sd = copy.deepcopy(original)
    for k in sd:
        if "cp_blocks_mid.transformer.ff.net.2.weight" in k or k.endswith("unet.conv_in.weight"):
            sd[k] = sd[k] * 1.5
    try:
        parent.load_state_dict(sd)
        after = m(*args)[0]
        assert rel_l2(after, before) > 1e-3    # stale packed weights would give the identical output
    finally:
        parent.load_state_dict(original)       # the UNet modules are shared with the module-scoped oracle fixture


def test_pano_only_and_unpadded(fake_backend, oracle_model):
    g = golden("mvgen_tiny.npz")
    pl, ppe = torch.from_numpy(g["pano_latent"]), torch.from_numpy(g["pano_prompt_embd"])
    for pad in (True, False):
        o = MV.DualBranchDenoiser(None, oracle_model.pano_unet, pano_pad=pad)
        with torch.no_grad():
            _, want = o(None, pl, torch.tensor([981, 981]), None, ppe, None)
        s, got = hip_model(o)(None, pl, torch.tensor([981, 981]), None, ppe, None)
        assert s is None and rel_l2(got, want) < 2e-5


def test_controlnet_residuals(fake_backend, oracle_model):
    """Layout-conditioned call (reference MVGenModel.py:62-83,154-170,200-203):
    ControlNet on the panorama branch (the reference's default, PanoGenerator.py:186-191) and on both."""
    from oracle import sd2_unet as U
    from panfusion_amd.models.pano import MultiViewBaseModel
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    cams = {k: torch.stack([v, v]) for k, v in cam4().items()}
    gen = torch.Generator().manual_seed(7)

    def make_cn(unet, seed):
        cn = U.ControlNetModel.from_unet(unet)
        U.init_synthetic(cn.controlnet_cond_embedding, seed)
        U.init_synthetic(cn.controlnet_down_blocks, seed + 1)       # zero-initialised in diffusers: would test nothing
        U.init_synthetic(cn.controlnet_mid_block, seed + 2)
        return cn

    pano_cn, pers_cn = make_cn(oracle_model.pano_unet, 71), make_cn(oracle_model.unet, 75)
    pano_cond = torch.rand(2, 1, 3, 128, 256, generator=gen) * 2 - 1          # 8x the 16x32 latent
    pers_cond = torch.rand(2, 4, 3, 128, 128, generator=gen) * 2 - 1
    for use_pers in (False, True):
        o = MV.DualBranchDenoiser(oracle_model.unet, oracle_model.pano_unet, pers_cn if use_pers else None, pano_cn)
        o.load_state_dict({k: v for k, v in oracle_model.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
        args = (t("latents"), t("pano_latent"), torch.full((2, 4), 981), t("prompt_embd"), t("pano_prompt_embd"), cams)
        with torch.no_grad():
            ws, wp = o(*args, pers_layout_cond=pers_cond, pano_layout_cond=pano_cond)
            ws0, wp0 = o(*args)
        assert rel_l2(wp, wp0) > 1e-3, "the ControlNet must change the output for the test to mean anything"
        m = MultiViewBaseModel(o.unet, o.pano_unet, o.pers_cn, o.pano_cn, True, compute_dtype=torch.float32)
        m.load_state_dict({k: v for k, v in o.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
        s, ps = m(*args, pers_layout_cond=pers_cond, pano_layout_cond=pano_cond)
        assert rel_l2(s, ws) < 2e-5 and rel_l2(ps, wp) < 2e-5, (use_pers, rel_l2(s, ws), rel_l2(ps, wp))
        s0, ps0 = m(*args)                                                      # no cond -> ControlNet skipped
        assert rel_l2(s0, ws0) < 2e-5 and rel_l2(ps0, wp0) < 2e-5


def test_per_sample_cameras(fake_backend, oracle_model):
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    c = cam4()
    cams = {"FoV": torch.stack([c["FoV"], c["FoV"]]), "theta": torch.stack([c["theta"], c["theta"] + 33.0]),
            "phi": torch.stack([c["phi"], c["phi"] - 5.0])}
    args = (t("latents"), t("pano_latent"), torch.full((2, 4), 500), t("prompt_embd"), t("pano_prompt_embd"), cams)
    with torch.no_grad():
        ws, wp = oracle_model(*args)
    s, ps = hip_model(oracle_model)(*args)
    assert rel_l2(s, ws) < 2e-5 and rel_l2(ps, wp) < 2e-5


def test_sampling_loop_matches_golden(fake_backend, oracle_model):
    from panfusion_amd.pipeline import DenoiseLoop
    g, gd = golden("mvgen_tiny.npz"), golden("ddim3_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    cam1 = {k: v[None] for k, v in cam4().items()}
    loop = DenoiseLoop(hip_model(oracle_model), t("latents")[:1], t("pano_latent")[:1], t("prompt_embd"),
                       t("pano_prompt_embd"), cam1, steps=3)
    l3, p3 = loop.run()
    assert rel_l2(l3, torch.from_numpy(gd["latents"])) < 1e-4
    assert rel_l2(p3, torch.from_numpy(gd["pano_latent"])) < 1e-4


def test_loop_rolls_layout_condition_with_the_panorama(fake_backend):
    """PanFusion.py:150-153: the layout condition image is rolled by rot_diff every step, like the panorama."""
    from panfusion_amd.pipeline import DenoiseLoop
    seen = []

    class Probe:
        def __call__(self, lat, pano, t, pe, ppe, cams, pers_cond=None, pano_cond=None):
            seen.append(pano_cond.clone())
            return torch.zeros_like(lat), torch.zeros_like(pano)

    gen = torch.Generator().manual_seed(3)
    cond = torch.rand(1, 1, 3, 16, 64, generator=gen)
    cam1 = {k: v[None] for k, v in cam4().items()}
    loop = DenoiseLoop(Probe(), torch.zeros(1, 4, 4, 4, 4), torch.zeros(1, 1, 4, 4, 8), torch.zeros(2, 4, 3, 8),
                       torch.zeros(2, 1, 3, 8), cam1, steps=6, pano_layout_cond=cond)
    loop.run()
    assert len(seen) == 6
    for i, c in enumerate(seen):
        want = torch.roll(cond, int(90 / 360 * 64) * (i + 1), -1)
        assert c.shape[0] == 2 and torch.equal(c[0], want[0]) and torch.equal(c[1], want[0])


def test_reference_api_wrappers(fake_backend):
    from panfusion_amd.external.Perspective_and_Equirectangular import e2p, p2e
    from panfusion_amd.models.pano import get_coords, get_masks
    from panfusion_amd.utils.pano import pad_pano, unpad_pano
    from oracle import geometry as G
    c = cam4()
    x = torch.randn(4, 3, 8, 16)
    assert torch.equal(e2p(x, c["FoV"], c["theta"], c["phi"], (8, 8), mode="nearest"),
                       G.e2p(x, c["FoV"], c["theta"], c["phi"], (8, 8), mode="nearest"))
    y = torch.randn(4, 3, 8, 8)
    a, b = p2e(y, c["FoV"], c["theta"], c["phi"], (8, 16)), G.p2e(y, c["FoV"], c["theta"], c["phi"], (8, 16))
    assert torch.allclose(a[0], b[0], atol=1e-6) and torch.equal(a[1], b[1])
    pm, em = get_masks(8, 8, 8, 16, c, "cpu")
    wm, we = G.get_masks(8, 8, 8, 16, c)
    assert pm.shape == wm.shape and torch.allclose(pm, wm, atol=1e-6) and torch.allclose(em, we, atol=1e-6)
    pc, ec = get_coords(8, 8, 8, 16, c, "cpu")
    wc, wec = G.get_coords(8, 8, 8, 16, c)
    assert torch.equal(pc, wc) and torch.equal(ec, wec)
    z = torch.randn(2, 2, 3, 4, 8)
    assert torch.equal(unpad_pano(pad_pano(z, 2), 2), z) and torch.equal(pad_pano(z, 2), G.pad_pano(z, 2))


@pytest.mark.parametrize("precision,dtype,tol", [("fast", torch.float32, 2e-5), ("mixed", torch.float32, 2e-5),
                                                 ("mixed", torch.float16, 2e-3)])
def test_vae_decode_host_logic(fake_backend, precision, dtype, tol):
    """SURVEY.md §8f row 1 (PanFusion.py:166-172, PanoGenerator.py:213-238): weight packing and layer sequencing of
    the VAE decoder (post_quant_conv as a centre tap, value bias folded behind the output projection, attention as
    two batched GEMMs + row softmax, nearest-x2 upsampling convs, padded-panorama decode + crop, tensor_to_image)
    on the CPU test double against the oracle restatement of diffusers' AutoencoderKL decoder."""
    from oracle import sd2_unet as U
    from oracle import vae as OV
    from panfusion_amd import vae as PV
    from panfusion_amd.models.vae_params import VAEDecoderParams
    cfg = OV.tiny_vae_config(width=64, groups=8)
    ov = OV.AutoencoderKLDecoder(**cfg)
    U.init_synthetic(ov, 51)
    params = VAEDecoderParams(**cfg)
    params.load_state_dict({k: v for k, v in ov.state_dict().items() if k.startswith(("decoder.", "post_quant_conv."))},
                           strict=True)                                    # identical key sets per half (diffusers names)
    dec = PV.VAEDecoder(params, compute_dtype=dtype, precision=precision)
    g = torch.Generator().manual_seed(5)
    lat, pano = torch.randn(1, 2, 4, 8, 8, generator=g), torch.randn(1, 1, 4, 8, 16, generator=g)
    with torch.no_grad():
        want = OV.decode_latent(lat, ov)
        wi, wp = OV.decode_views_and_pano(lat, pano, ov, latent_pad=4)
    got = PV.decode_latent(lat, dec)
    assert got.shape == (1, 2, 3, 64, 64) and rel_l2(got, want) < tol, rel_l2(got, want)
    gi, gp = PV.decode_views_and_pano(lat, pano, dec, latent_pad=4)
    assert gi.dtype == torch.uint8 and gi.shape == (1, 2, 64, 64, 3) and gp.shape == (1, 1, 64, 128, 3)
    want_u8 = lambda x: OV.tensor_to_image(x)
    # uint8 images: identical except where fp32 round-off straddles a rounding boundary
    assert float((gi.int() - want_u8(wi).int()).abs().float().mean()) < (0.01 if dtype == torch.float32 else 0.5)
    assert int((gp.int() - want_u8(wp).int()).abs().max()) <= (1 if dtype == torch.float32 else 8)


@pytest.mark.parametrize("precision,dtype,tol", [("fast", torch.float32, 2e-5), ("mixed", torch.float32, 2e-5), ("mixed", torch.float16, 2e-3)])
def test_vae_encode_host_logic(fake_backend, precision, dtype, tol):
    """The encoder half for the training step (PanoGenerator.encode_image, PanoGenerator.py:214-225): weight packing and
    sequencing (down-convs with the trailing zero row / column of Downsample2D(padding=0), mid attention, moments through
    quant_conv as a centre tap, posterior sample with a given normal draw) on the CPU test double against the oracle."""
    from oracle import sd2_unet as U
    from oracle import vae as OV
    from panfusion_amd import vae as PV
    from panfusion_amd.models.vae_params import VAEEncoderParams
    cfg = OV.tiny_vae_config(width=64, groups=8)
    ov = OV.AutoencoderKL(**cfg)
    U.init_synthetic(ov, 52)
    params = VAEEncoderParams(**cfg)
    params.load_state_dict({k: v for k, v in ov.state_dict().items() if k.startswith(("encoder.", "quant_conv."))}, strict=True)
    enc = PV.VAEEncoder(params, compute_dtype=dtype, precision=precision)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(1, 2, 3, 64, 64, generator=g) * 2 - 1
    eps = torch.randn(1, 2, 4, 8, 8, generator=g)
    with torch.no_grad():
        dist = ov.encode(x.flatten(0, 1)).latent_dist
        want = OV.encode_image(x, ov, eps=eps.flatten(0, 1))
    mean, logvar = enc.encode(x.flatten(0, 1))
    assert rel_l2(mean, dist.mean) < tol and rel_l2(logvar, dist.logvar) < tol
    got = PV.encode_image(x, enc, eps=eps)
    assert got.shape == (1, 2, 4, 8, 8) and rel_l2(got, want) < tol, rel_l2(got, want)


def tiny_clip(seed=0, layers=3):
    """transformers' own CLIPTextModel (the class the reference loads, PanoGenerator.py:116-118) at a small width with
    heads of 64 like OpenCLIP-H, random weights: the oracle for the text-encoding row (third-party code that IS
    installed in this image, unlike diffusers)."""
    from transformers import CLIPTextConfig, CLIPTextModel
    torch.manual_seed(seed)
    cfg = CLIPTextConfig(hidden_size=128, intermediate_size=512, num_attention_heads=2, num_hidden_layers=layers,
                         vocab_size=1000, max_position_embeddings=77, hidden_act="gelu", bos_token_id=998, eos_token_id=999,
                         pad_token_id=0)
    m = CLIPTextModel(cfg).eval()
    with torch.no_grad():                     # HF initialises biases / LN to trivial values: randomise them
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return m


@pytest.mark.parametrize("dtype,precision,tol", [(torch.float32, "fast", 2e-5), (torch.float16, "mixed", 1e-3)])
def test_text_encoder_host_logic(fake_backend, dtype, precision, tol):
    """SURVEY.md §8f row 2 (PanoGenerator.py:197-211): weight packing and layer sequencing of the CLIP text encoder
    (value bias folded behind out_proj, GELU through the GEGLU epilogue with constant-one value rows, causal +
    padding mask as the attention kernel's bias table, sequence padded 77 -> 80) on the CPU test double against
    transformers' CLIPTextModel."""
    from panfusion_amd.text_encoder import TextEncoder
    m = tiny_clip()
    ids = torch.randint(1, 990, (3, 77), generator=torch.Generator().manual_seed(1))
    ids[:, 0], ids[0, 20:], ids[1, 50:] = 998, 999, 999
    with torch.no_grad():
        want = m(ids)[0]
    got = TextEncoder(m, compute_dtype=dtype, precision=precision).encode_ids(ids)
    assert got.shape == want.shape == (3, 77, 128) and got.dtype == torch.float32
    assert rel_l2(got, want) < tol, rel_l2(got, want)
