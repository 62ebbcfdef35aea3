"""Host logic of the EPA training path (panfusion_amd/training.py: recompute + backward sequencing, which gradient lands
in which slot, the power-of-two gradient normalisation, the autograd wiring of WarpAttn.forward) on the CPU with
tests/fake_ops.py standing in for the HIP front end, against torch autograd through the oracle's EPA block (reference
models/pano/modules.py:15-59 + models/modules/transformer.py:40-161).  fp32 throughout: agreement at round-off."""
import importlib

import pytest
import torch

import fake_ops
from conftest import cam4, rel_l2
from oracle import mvgen as MV

MODS = ["panfusion_amd.engine", "panfusion_amd.training", "panfusion_amd.models.pano.modules"]


@pytest.fixture
def fake_backend(monkeypatch):
    for name in MODS:
        monkeypatch.setattr(importlib.import_module(name), "ops", fake_ops)


def make_blocks(dim, seed=0):
    from panfusion_amd.models.pano import WarpAttn
    torch.manual_seed(seed)
    ref = MV.EPABlock(dim)
    with torch.no_grad():                      # zero-initialised output layers would make most gradients vanish
        for p in ref.parameters():
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() == 1 else p.shape[-1] ** -0.5))
        ref.transformer.norm1.weight.add_(1.0)
        ref.transformer.norm2.weight.add_(1.0)
    hip = WarpAttn(dim, compute_dtype=torch.float32, precision="fast")
    hip.load_state_dict(ref.state_dict())
    return ref, hip


def run_case(ref, hip, b, dim, ph, eh, grad_scale=1.0, seed=1):
    g = torch.Generator().manual_seed(seed)
    m = 4
    cams = {k: torch.cat([v] * b) for k, v in cam4().items()}
    xp = torch.randn(b * m, dim, ph, ph, generator=g)
    xe = torch.randn(b, dim, eh, 2 * eh, generator=g)
    wp, we = torch.randn(xp.shape, generator=g) * grad_scale, torch.randn(xe.shape, generator=g) * grad_scale
    out = {}
    for name, mod in (("ref", ref), ("hip", hip)):
        a, c = xp.clone().requires_grad_(True), xe.clone().requires_grad_(True)
        for p in mod.parameters():
            p.grad = None
        op, oe = mod(a, c, cams)
        ((op * wp).sum() + (oe * we).sum()).backward()
        out[name] = dict(op=op.detach(), oe=oe.detach(), dxp=a.grad, dxe=c.grad,
                         **{k: p.grad for k, p in mod.named_parameters()})
    return out


def test_epa_backward_matches_autograd(fake_backend):
    ref, hip = make_blocks(64)
    out = run_case(ref, hip, 1, 64, 8, 8)
    assert set(out["ref"]) == set(out["hip"]) and len(out["ref"]) == 4 + 13
    for k, want in out["ref"].items():
        assert want is not None and out["hip"][k] is not None, k
        assert rel_l2(out["hip"][k], want) < 2e-5, (k, rel_l2(out["hip"][k], want))


def test_epa_backward_batch_of_two_and_tiny_gradients(fake_backend):
    """b = 2 (one attention launch over both samples: same cameras) and upstream gradients of 1e-7 (what an MSE over a
    latent batch hands down): the device-side normalisation keeps the 16-bit operands in range and is undone exactly."""
    ref, hip = make_blocks(64, seed=3)
    out = run_case(ref, hip, 2, 64, 8, 8, grad_scale=1e-7)
    for k, want in out["ref"].items():
        assert rel_l2(out["hip"][k], want) < 2e-5, (k, rel_l2(out["hip"][k], want))
    assert float(out["hip"]["dxp"].abs().max()) < 1e-4            # (the gradients really are tiny)


def test_optimizer_step_repacks_forward_and_backward_weights(fake_backend):
    ref, hip = make_blocks(64, seed=5)
    opt_r, opt_h = torch.optim.SGD(ref.parameters(), lr=0.05), torch.optim.SGD(hip.parameters(), lr=0.05)
    first = run_case(ref, hip, 1, 64, 8, 8)
    opt_r.step()
    opt_h.step()
    second = run_case(ref, hip, 1, 64, 8, 8)
    assert rel_l2(second["ref"]["op"], first["ref"]["op"]) > 1e-3       # the step changed the block
    for k, want in second["ref"].items():
        assert rel_l2(second["hip"][k], want) < 5e-5, (k, rel_l2(second["hip"][k], want))


def test_inference_path_unchanged_without_grad(fake_backend):
    ref, hip = make_blocks(64, seed=7)
    cams = cam4()
    xp, xe = torch.randn(4, 64, 8, 8), torch.randn(1, 64, 8, 16)
    with torch.no_grad():
        op, oe = hip(xp, xe, cams)
        rp, re_ = ref(xp, xe, cams)
    assert not op.requires_grad and rel_l2(op, rp) < 2e-5 and rel_l2(oe, re_) < 2e-5
    for p in hip.parameters():
        p.requires_grad_(False)
    op, oe = hip(xp, xe, cams)                   # nothing requires grad: no autograd node either
    assert op.grad_fn is None


# ------------------------------------------------------------------------------------ data-parallel training (gloo)
def _ddp_worker(rank, world, port, out):
    import os
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    for name in MODS:
        importlib.import_module(name).ops = fake_ops
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, hip = make_blocks(64, seed=9)
        ddp = torch.nn.parallel.DistributedDataParallel(hip)
        g = torch.Generator().manual_seed(100 + rank)                      # every rank its own sample
        xp, xe = torch.randn(4, 64, 8, 8, generator=g), torch.randn(1, 64, 8, 16, generator=g)
        op, oe = ddp(xp, xe, cam4())
        (op.square().mean() + oe.square().mean()).backward()
        torch.save({k: p.grad for k, p in hip.named_parameters()}, os.path.join(out, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_ddp_all_reduces_the_block_gradients(tmp_path):
    """The reference trains with DDP (main.py:68, one sample per GPU): the block's parameter gradients come out of an
    autograd.Function, so DDP's reducer hooks see them like any other -- two gloo ranks with different samples end with
    the mean of the two single-process gradients (the RCCL all-reduce of the GPU run is the same code path)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(tmp_path / ("r%d.pt" % r)) for r in range(2)]
    singles = []
    for rank in range(2):                                # autograd through the oracle block, one process
        ref, _ = make_blocks(64, seed=9)
        g = torch.Generator().manual_seed(100 + rank)
        xp, xe = torch.randn(4, 64, 8, 8, generator=g), torch.randn(1, 64, 8, 16, generator=g)
        op, oe = ref(xp, xe, cam4())
        (op.square().mean() + oe.square().mean()).backward()
        singles.append({k: p.grad for k, p in ref.named_parameters()})
    for k in singles[0]:
        want = (singles[0][k] + singles[1][k]) / 2
        assert torch.equal(got[0][k], got[1][k])
        assert rel_l2(got[0][k], want) < 2e-5, (k, rel_l2(got[0][k], want))


# ------------------------------------------------------------------------------------ the whole denoiser (host logic)
DEN_MODS = MODS + ["panfusion_amd.train_engine"]


@pytest.fixture
def fake_denoiser_backend(monkeypatch):
    for name in DEN_MODS:
        monkeypatch.setattr(importlib.import_module(name), "ops", fake_ops)


def _denoiser_case(seed=0, lora_rank=4):
    from conftest import build_tiny_oracle, golden
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    oracle = build_tiny_oracle(lora_rank=lora_rank)
    cams = {k: v[None] for k, v in cam4().items()}
    args = (t("latents")[:1], t("pano_latent")[:1], torch.full((1, 4), 981), t("prompt_embd")[:1], t("pano_prompt_embd")[:1], cams)
    gen = torch.Generator().manual_seed(seed)
    w_s, w_p = torch.randn(args[0].shape, generator=gen), torch.randn(args[1].shape, generator=gen)
    return oracle, args, w_s, w_p


@pytest.mark.parametrize("precision,lora_rank", [("fast", 4), ("mixed", 4), ("mixed", 8)])
def test_denoiser_training_step_matches_autograd(fake_denoiser_backend, precision, lora_rank):
    """One training step of the dual-branch denoiser (forward on the engine, backward on train_engine's tape) against torch
    autograd through the oracle denoiser (reference MultiViewBaseModel + EPA on the restated UNets with UNFUSED LoRA):
    gradients of every EPA parameter and every LoRA matrix of both UNets.  fp32 test double: agreement at round-off.  lora_rank 8:
    the stacked rank of a q / k / v group is 24 (more than one 16-row chunk of the column-sum kernel)."""
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, args, w_s, w_p = _denoiser_case(lora_rank=lora_rank)
    s, ps = oracle(*args)
    ((s * w_s).sum() + (ps * w_p).sum()).backward()
    want = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None}
    for p in oracle.parameters():
        p.grad = None
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                             precision=precision, differentiable=True)     # mixed: fp32 streams, split-precision operands / weights
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    s2, ps2 = hip(*args)
    assert s2.requires_grad and rel_l2(s2, s) < 2e-5 and rel_l2(ps2, ps) < 2e-5
    ((s2 * w_s).sum() + (ps2 * w_p).sum()).backward()
    got = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    lora = [k for k in want if "lora" in k]
    epa = [k for k in want if k.startswith("cp_blocks")]
    assert len(lora) == 2 * 16 * 2 * 4 * 2 and len(epa) == 7 * 13        # 2 UNets x 16 blocks x 2 attentions x 4 projections x (down, up)
    frozen = [k for k in got if k not in lora and k not in epa]
    assert not frozen, frozen[:4]
    worst = max((rel_l2(got[k], want[k]), k) for k in lora + epa)
    print("worst gradient: %.2e at %s" % worst)
    for k in lora + epa:
        assert k in got, k
        assert rel_l2(got[k], want[k]) < 1e-4, (k, rel_l2(got[k], want[k]))


def test_recompute_mode_gives_the_same_gradients(fake_denoiser_backend, monkeypatch):
    """PF_TRAIN_KEEP=0 (the backward recomputes every layer from its input: the low-memory mode, and round 2's behaviour) against
    the default (the training forward keeps its activations): same outputs up to the forward's fusion differences, same
    gradients up to round-off -- with the panorama ControlNet trainable, so that every kind of tape entry is walked both ways."""
    from panfusion_amd import train_engine
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, cn, cns, args, kw, w_s, w_p = _with_controlnet("pano")
    grads = {}
    for keep in (True, False):
        monkeypatch.setattr(train_engine, "KEEP", keep)
        hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, *cns, oracle.pano_pad, compute_dtype=torch.float32,
                                 precision="mixed", differentiable=True)
        hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
        for p in hip.parameters():
            p.grad = None
        s2, ps2 = hip(*args, **kw)
        ((s2 * w_s).sum() + (ps2 * w_p).sum()).backward()
        grads[keep] = {k: p.grad.clone() for k, p in hip.named_parameters() if p.grad is not None}
    assert sorted(grads[True]) == sorted(grads[False]) and len(grads[True]) > 900
    worst = max((rel_l2(grads[True][k], grads[False][k]), k) for k in grads[True])
    assert worst[0] < 1e-4, worst


def test_denoiser_inference_is_untouched_by_the_training_switch(fake_denoiser_backend):
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, args, _, _ = _denoiser_case()
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="fast", differentiable=True)
    with torch.no_grad():
        s, ps = hip(*args)
    assert not s.requires_grad and s.grad_fn is None
    hip.differentiable = False
    s2, ps2 = hip(*args)
    assert s2.grad_fn is None and torch.equal(s, s2) and torch.equal(ps, ps2)


def test_optimizer_step_refolds_lora_into_the_forward_weights(fake_denoiser_backend):
    """After an optimizer step on the LoRA matrices the next forward must see W + up' @ down' (re-folded attention
    projections; the cached text K / V^T recomputed INTO their buffers, which graphs captured earlier read by address) -- and nothing
    else is re-packed."""
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, args, w_s, w_p = _denoiser_case()
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="fast", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    opt = torch.optim.SGD(hip.trainable_tensors(), lr=0.5)
    s0, ps0 = hip(*args)
    ((s0 * w_s).sum() + (ps0 * w_p).sum()).backward()
    pack_before = hip.packed("unet", args[1].device)
    res_before = pack_before.down[0].resnets[1].train       # (resnets[0] sits before the first LoRA-carrying layer: never walked)
    with torch.no_grad():
        hip(*args)                                        # an inference forward (validation, DenoiseLoop): caches the prompts' K / V^T
    # (a captured DenoiseLoop graph pins the entries it reads by address -- pipeline.DenoiseLoop._graph_keepalive; the panorama
    # UNet's entry stays unpinned here: the re-fold must DROP it instead of recomputing K / V^T nobody will read)
    import weakref

    class _GraphEntry:                                    # stands for the graph entry whose keep-alive list holds the pin token
        pass
    entry = _GraphEntry()
    for hit in pack_before.text_kv_cache.values():
        hit.setdefault("pins", weakref.WeakSet()).add(entry)
    pano_pack = hip.packed("pano_unet", args[1].device)
    assert getattr(pano_pack, "text_kv_cache", {})
    kv_before = {k: (hit, {i: (v[0].data_ptr(), v[0].clone()) for i, v in hit.items() if isinstance(i, int)})
                 for k, hit in getattr(pack_before, "text_kv_cache", {}).items()}
    opt.step()                                            # oracle shares the UNet modules: it steps with it
    for name in ("cp_blocks_encoder", "cp_blocks_mid", "cp_blocks_decoder"):
        getattr(oracle, name).load_state_dict(getattr(hip, name).state_dict())
    with torch.no_grad():
        want_s, want_ps = oracle(*args)
    s1, ps1 = hip(*args)
    assert rel_l2(want_s, s0.detach()) > 1e-3                      # the step moved the outputs
    assert rel_l2(s1.detach(), want_s) < 5e-5 and rel_l2(ps1.detach(), want_ps) < 5e-5
    assert hip.packed("unet", args[1].device) is pack_before and pack_before.down[0].resnets[1].train is res_before
    with torch.no_grad():
        s2, ps2 = hip(*args)                              # served from the cache
    assert rel_l2(s2, want_s) < 5e-5 and rel_l2(ps2, want_ps) < 5e-5
    assert kv_before
    assert all(not h.get("pins") for h in pano_pack.text_kv_cache.values())            # dropped at the re-fold, rebuilt on use
    for k, (hit, olds) in kv_before.items():                 # same entries, same storage, new contents
        assert pack_before.text_kv_cache[k] is hit
        for i, (ptr, old) in olds.items():
            assert hit[i][0].data_ptr() == ptr and not torch.equal(hit[i][0], old)
    # once the graph entry is gone the pin is gone -- the next re-fold drops the entry instead of refreshing it for ever
    del entry
    import gc
    gc.collect()
    assert all(not h.get("pins") for h in pack_before.text_kv_cache.values())
    ((s1 * w_s).sum() + (ps1 * w_p).sum()).backward()
    opt.step()
    hip.refold_lora()
    assert not pack_before.text_kv_cache, "un-pinned text K / V^T entries are dropped at the re-fold"


def test_training_step_with_layout_condition_frozen_controlnet(fake_denoiser_backend):
    """PanFusion.training_step with ``pano_layout_cond`` (PanFusion.py:85-89): the panorama ControlNet's residuals enter the
    tape as constants; EPA + LoRA gradients == torch autograd through the oracle with the ControlNet's parameters frozen."""
    from panfusion_amd.models.pano import MultiViewBaseModel
    from oracle import mvgen as MV
    from oracle import sd2_unet as U
    oracle0, args, w_s, w_p = _denoiser_case()
    cn = U.ControlNetModel.from_unet(oracle0.pano_unet)
    U.init_synthetic(cn.controlnet_cond_embedding, 71)
    U.init_synthetic(cn.controlnet_down_blocks, 72)
    U.init_synthetic(cn.controlnet_mid_block, 73)
    cn.requires_grad_(False)
    oracle = MV.DualBranchDenoiser(oracle0.unet, oracle0.pano_unet, None, cn, oracle0.pano_pad)
    oracle.load_state_dict({k: v for k, v in oracle0.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    pl = args[1]
    cond = torch.rand(1, 1, 3, pl.shape[-2] * 8, pl.shape[-1] * 8, generator=torch.Generator().manual_seed(9)) * 2 - 1
    s, ps = oracle(*args, pano_layout_cond=cond)
    s0, ps0 = oracle(*args)
    assert rel_l2(ps, ps0) > 1e-2                                  # the condition does something
    ((s * w_s).sum() + (ps * w_p).sum()).backward()
    want = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None}
    for p in oracle.parameters():
        p.grad = None
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, cn, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="mixed", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    s2, ps2 = hip(*args, pano_layout_cond=cond)
    assert s2.requires_grad and rel_l2(s2, s) < 2e-5 and rel_l2(ps2, ps) < 2e-5
    ((s2 * w_s).sum() + (ps2 * w_p).sum()).backward()
    got = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    keys = [k for k in want if "lora" in k or k.startswith("cp_blocks")]
    assert len(keys) == 2 * 16 * 2 * 4 * 2 + 7 * 13
    assert not [k for k in got if k.startswith("pano_cn")]         # frozen: no ControlNet gradients
    for k in keys:
        assert k in got and rel_l2(got[k], want[k]) < 1e-4, (k, rel_l2(got[k], want[k]) if k in got else None)


def _with_controlnet(which="pano", seed=71):
    """The tiny oracle denoiser with a ControlNet on the panorama branch (the reference's default, PanoGenerator.py:186-191) or
    on the view branch (pers_cn, MVGenModel.py:66-74), and the matching layout images."""
    from oracle import mvgen as MV
    from oracle import sd2_unet as U
    oracle0, args, w_s, w_p = _denoiser_case()
    cn = U.ControlNetModel.from_unet(oracle0.pano_unet if which == "pano" else oracle0.unet)
    U.init_synthetic(cn.controlnet_cond_embedding, seed)
    U.init_synthetic(cn.controlnet_down_blocks, seed + 1)
    U.init_synthetic(cn.controlnet_mid_block, seed + 2)
    cns = (None, cn) if which == "pano" else (cn, None)
    oracle = MV.DualBranchDenoiser(oracle0.unet, oracle0.pano_unet, *cns, oracle0.pano_pad)
    oracle.load_state_dict({k: v for k, v in oracle0.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    lat = args[1] if which == "pano" else args[0]
    cond = torch.rand(*lat.shape[:2], 3, lat.shape[-2] * 8, lat.shape[-1] * 8, generator=torch.Generator().manual_seed(9)) * 2 - 1
    kw = {"pano_layout_cond": cond} if which == "pano" else {"pers_layout_cond": cond}
    return oracle, cn, cns, args, kw, w_s, w_p


@pytest.mark.parametrize("which", ["pano", "pers"])
def test_training_step_with_trainable_controlnet(fake_denoiser_backend, which):
    """The reference's layout_cond=True training (PanoGenerator.py:153-157, 165-168: every parameter of the ControlNet trains;
    PanFusion.py:85-89 passes the layout images): gradients of ALL ControlNet parameters -- conditioning embedding, conv_in,
    time embedding, every resnet / transformer / down-sampler, the 13 zero-convs -- and still of the EPA blocks and the LoRA
    matrices, against torch autograd through the oracle.  fp32 test double: agreement at round-off.  Both attachment points:
    the panorama branch (one image per sample) and the view branch (m images per sample)."""
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, cn, cns, args, kw, w_s, w_p = _with_controlnet(which)
    s, ps = oracle(*args, **kw)
    ((s * w_s).sum() + (ps * w_p).sum()).backward()
    want = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None}
    for p in oracle.parameters():
        p.grad = None
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, *cns, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="mixed", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    s2, ps2 = hip(*args, **kw)
    assert rel_l2(s2, s) < 2e-5 and rel_l2(ps2, ps) < 2e-5
    ((s2 * w_s).sum() + (ps2 * w_p).sum()).backward()
    got = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    cn_keys = [k for k in want if k.startswith(which + "_cn.")]
    assert len(cn_keys) == len(list(cn.parameters())) and len(cn_keys) > 300
    missing = [k for k in cn_keys if k not in got]
    assert not missing, missing[:6]
    errs = sorted(((rel_l2(got[k], want[k]), k) for k in cn_keys), reverse=True)
    print("trainable ControlNet (%s): %d tensors, worst %s" % (which, len(cn_keys), "  ".join("%.1e %s" % e for e in errs[:4])))
    assert errs[0][0] < 2e-4, errs[:6]
    for k in [k for k in want if "lora" in k or k.startswith("cp_blocks")]:
        assert k in got and rel_l2(got[k], want[k]) < 1e-4, k


def test_optimizer_step_on_the_controlnet_reaches_the_next_forward(fake_denoiser_backend):
    """A training ControlNet's pack is rebuilt when its parameters moved (packed(): version counters), the UNets' packs stay."""
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, cn, cns, args, kw, w_s, w_p = _with_controlnet("pano")
    cond = kw["pano_layout_cond"]
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, cn, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="fast", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    opt = torch.optim.SGD(list(cn.parameters()), lr=0.05)
    s0, ps0 = hip(*args, pano_layout_cond=cond)
    unet_pack = hip.packed("pano_unet", args[1].device)
    ((s0 * w_s).sum() + (ps0 * w_p).sum()).backward()
    opt.step()
    with torch.no_grad():
        want_s, want_ps = oracle(*args, pano_layout_cond=cond)
        s1, ps1 = hip(*args, pano_layout_cond=cond)
    assert rel_l2(want_ps, ps0.detach()) > 1e-4                    # the step moved the panorama output
    assert rel_l2(s1, want_s) < 5e-5 and rel_l2(ps1, want_ps) < 5e-5
    assert hip.packed("pano_unet", args[1].device) is unet_pack


def test_backward_stops_at_the_earliest_trainable_entry(fake_denoiser_backend, monkeypatch):
    """Frozen LoRA matrices (the reference's layout-conditioned runs, PanoGenerator.py:173): only the EPA blocks train -- their
    gradients are unchanged, no LoRA gradient is produced, and the backward does not walk the first encoder level of either
    branch (nothing trainable lies before the first EPA block; torch autograd prunes the same way)."""
    from panfusion_amd import train_engine
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, args, w_s, w_p = _denoiser_case()
    for k, p in oracle.named_parameters():
        if "lora" in k:
            p.requires_grad_(False)
    s, ps = oracle(*args)
    ((s * w_s).sum() + (ps * w_p).sum()).backward()
    want = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None and k.startswith("cp_blocks")}
    assert len(want) == 7 * 13
    for p in oracle.parameters():
        p.grad = None
    calls = []
    real = train_engine.resnet_backward
    monkeypatch.setattr(train_engine, "resnet_backward", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="mixed", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    s2, ps2 = hip(*args)
    ((s2 * w_s).sum() + (ps2 * w_p).sum()).backward()
    got = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(want)
    for k in want:
        assert rel_l2(got[k], want[k]) < 1e-4, k
    n_resnets = sum(len(b.resnets) for u in (oracle.unet, oracle.pano_unet) for b in [*u.down_blocks, u.mid_block, *u.up_blocks])
    assert len(calls) == n_resnets - 2 * 2                          # the two resnets of encoder level 0, both branches
    for p in oracle.parameters():
        p.requires_grad_(True)


def test_no_grad_forward_after_optimizer_step_sees_the_new_lora(fake_denoiser_backend):
    """validation / predict after fit (torch.no_grad, DenoiseLoop) goes straight to _forward -- it must re-fold
    the LoRA-carrying projections too, not mix the current EPA weights with the LoRA from before the step.  Also the
    in-place case: packs built by an inference forward, parameters changed with p.data.copy_ (no load_state_dict)."""
    import pickle
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, args, w_s, w_p = _denoiser_case()
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="fast", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    opt = torch.optim.SGD(hip.trainable_tensors(), lr=0.5)
    s0, ps0 = hip(*args)
    ((s0 * w_s).sum() + (ps0 * w_p).sum()).backward()
    opt.step()
    for name in ("cp_blocks_encoder", "cp_blocks_mid", "cp_blocks_decoder"):
        getattr(oracle, name).load_state_dict(getattr(hip, name).state_dict())
    with torch.no_grad():
        want_s, want_ps = oracle(*args)
        s1, ps1 = hip(*args)                               # inference forward right after the step
    assert rel_l2(want_s, s0.detach()) > 1e-3
    assert rel_l2(s1, want_s) < 5e-5 and rel_l2(ps1, want_ps) < 5e-5
    # in-place change of a LoRA matrix between two inference forwards
    with torch.no_grad():
        lora = [t for t in hip.trainable_tensors() if t.dim() == 2 and min(t.shape) == 4][0]
        lora.data.copy_(lora.data * 3.0)
        lora._version  # noqa: B018  (data.copy_ does not bump the counter: bump it the way an in-place op on the parameter does)
        lora.mul_(1.0)
        want_s2, want_ps2 = oracle(*args)
        s2, ps2 = hip(*args)
    assert rel_l2(want_s2, want_s) > 1e-5
    assert rel_l2(s2, want_s2) < 5e-5 and rel_l2(ps2, want_ps2) < 5e-5
    # the module (and an EPA block alone) survive pickling: no lambdas among the load_state_dict hooks
    pickle.dumps(hip.cp_blocks_mid)


def test_training_reaches_lora_in_the_attention_processor_layout(fake_denoiser_backend):
    """The LoRA matrices of a reference checkpoint sit in ``attn.processor.to_{q,k,v,out}_lora`` (set_attn_processor,
    PanoGenerator.py:132-151; no diffusers forward ever migrates them here): the training path must find them there --
    same gradients as with the same matrices in ``<linear>.lora_layer``."""
    from panfusion_amd.models.pano import MultiViewBaseModel
    from panfusion_amd.models.sd2_unet_params import UNetParams, fill_synthetic
    from oracle import sd2_unet as U
    from conftest import TINY, golden
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    cams = {k: v[None] for k, v in cam4().items()}
    args = (t("latents")[:1], t("pano_latent")[:1], torch.full((1, 4), 981), t("prompt_embd")[:1], t("pano_prompt_embd")[:1], cams)
    grads = {}
    for layout in ("lora_layer", "processor"):
        unets = []
        for seed in (5, 6):
            u = UNetParams(**U.tiny_config(**TINY))
            u.add_lora(4, layout="lora_layer")
            fill_synthetic(u, seed)                       # same seeded values for both layouts ...
            if layout == "processor":                     # ... moved into processors
                v = UNetParams(**U.tiny_config(**TINY))
                v.add_lora(4, layout="processor")
                sd = {}
                for k, p in u.state_dict().items():
                    for n in ("to_q", "to_k", "to_v"):
                        k = k.replace(".%s.lora_layer." % n, ".processor.%s_lora." % n)
                    k = k.replace(".to_out.0.lora_layer.", ".processor.to_out_lora.")
                    sd[k] = p
                v.load_state_dict(sd)
                u = v
            unets.append(u)
        model = MultiViewBaseModel(unets[0], unets[1], None, None, True, compute_dtype=torch.float32, precision="fast",
                                   differentiable=True)
        for i, blk in enumerate([*model.cp_blocks_encoder, model.cp_blocks_mid, *model.cp_blocks_decoder]):
            fill_synthetic(blk, 10 + i)
        s, ps = model(*args)
        (s.square().mean() + ps.square().mean()).backward()
        names = {id(p): k for k, p in model.named_parameters()}
        tens = model.trainable_tensors()
        assert len(tens) == 603 and all(p.grad is not None for p in tens)
        grads[layout] = {names[id(p)].replace(".processor.to_out_lora.", ".to_out.0.lora_layer.")
                         .replace(".processor.to_q_lora.", ".to_q.lora_layer.").replace(".processor.to_k_lora.", ".to_k.lora_layer.")
                         .replace(".processor.to_v_lora.", ".to_v.lora_layer."): p.grad for p in tens}
    assert set(grads["lora_layer"]) == set(grads["processor"])
    for k, want in grads["lora_layer"].items():
        assert rel_l2(grads["processor"][k], want) < 1e-6, k


def _training_batch(seed=3):
    g = torch.Generator().manual_seed(seed)
    cams = {k: v[None] for k, v in cam4().items()}
    cams["theta"], cams["phi"] = cams["theta"] + 7.3, cams["phi"] + 3.1      # (no nearest-neighbour ties in the noise projection)
    images, pano = torch.rand(1, 4, 3, 128, 128, generator=g) * 2 - 1, torch.rand(1, 1, 3, 128, 256, generator=g) * 2 - 1
    pe, ppe = torch.randn(1, 4, 7, 128, generator=g), torch.randn(1, 1, 7, 128, generator=g)
    draws = dict(eps_views=torch.randn(1, 4, 4, 16, 16, generator=g), eps_pano=torch.randn(1, 1, 4, 16, 32 + 2 * 8, generator=g),
                 t=torch.tensor([481]), pano_noise=torch.randn(1, 1, 4, 16, 32, generator=g))
    return images, pano, cams, pe, ppe, draws


def test_whole_training_step_matches_the_oracle(fake_denoiser_backend, monkeypatch):
    """pipeline.training_step = the body of PanFusion.training_step (PanFusion.py:64-98): VAE encode of the views and the padded
    panorama, init_noise, add_noise, denoiser, the two MSE losses -- against the oracle restatement with the same draws; then
    loss.backward() against autograd through the oracle."""
    from oracle import ddim as OD
    from oracle import sd2_unet as U
    from oracle import vae as OV
    from conftest import build_tiny_oracle
    from panfusion_amd import pipeline, vae as PV
    from panfusion_amd.models.pano import MultiViewBaseModel
    from panfusion_amd.models.vae_params import VAEEncoderParams
    for name in ("panfusion_amd.pipeline", "panfusion_amd.vae", "panfusion_amd.utils.pano",
                 "panfusion_amd.external.Perspective_and_Equirectangular.e2p"):
        monkeypatch.setattr(importlib.import_module(name), "ops", fake_ops)
    om = build_tiny_oracle()
    cfg = OV.tiny_vae_config(width=64, groups=8)
    ov = OV.AutoencoderKL(**cfg)
    U.init_synthetic(ov, 53)
    images, pano, cams, pe, ppe, draws = _training_batch()
    want = OD.training_step(om, ov, images, pano, cams, pe, ppe, draws)
    want[0].backward()
    wg = {k: p.grad.clone() for k, p in om.named_parameters() if p.grad is not None and ("lora" in k or k.startswith("cp_blocks"))}
    for p in om.parameters():
        p.grad = None
    params = VAEEncoderParams(**cfg)
    params.load_state_dict({k: v for k, v in ov.state_dict().items() if k.startswith(("encoder.", "quant_conv."))}, strict=True)
    enc = PV.VAEEncoder(params, compute_dtype=torch.float32, precision="fast")
    hip = MultiViewBaseModel(om.unet, om.pano_unet, None, None, om.pano_pad, compute_dtype=torch.float32, precision="fast", differentiable=True)
    hip.load_state_dict({k: v for k, v in om.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    got = pipeline.training_step(hip, enc, images, pano, cams, pe, ppe, draws=draws)
    for a, b in zip(got, want):
        a, b = float(a.detach()), float(b.detach())
        assert abs(a - b) <= 2e-5 * abs(b), (a, b)
    got[0].backward()
    gg = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    assert len(wg) == 603
    for k, w in wg.items():
        assert rel_l2(gg[k], w) < 1e-4, (k, rel_l2(gg[k], w))


def _ddp_denoiser_worker(rank, world, port, out):
    import os
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    for name in DEN_MODS:
        importlib.import_module(name).ops = fake_ops
    from panfusion_amd.models.pano import MultiViewBaseModel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle, args, _, _ = _denoiser_case()
        hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                                 precision="fast", differentiable=True)
        hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
        train = {id(t) for t in hip.trainable_tensors()}
        for p in hip.parameters():                       # the reference freezes the UNets and trains LoRA + EPA only
            p.requires_grad_(id(p) in train)
        ddp = torch.nn.parallel.DistributedDataParallel(hip)
        g = torch.Generator().manual_seed(200 + rank)    # every rank its own sample (one per GPU in the reference)
        lat, pl = torch.randn(args[0].shape, generator=g), torch.randn(args[1].shape, generator=g)
        s, ps = ddp(lat, pl, *args[2:])
        (s.square().mean() + ps.square().mean()).backward()
        torch.save({k: p.grad for k, p in hip.named_parameters() if p.grad is not None}, os.path.join(out, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_ddp_over_the_whole_denoiser(tmp_path):
    """DDP (the reference's strategy, main.py:68) around the differentiable denoiser: two gloo ranks with different samples end
    with the mean of the two single-process gradients on all 603 trainable tensors."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_ddp_denoiser_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(tmp_path / ("r%d.pt" % r)) for r in range(2)]
    singles = []
    for rank in range(2):                                # autograd through the oracle, one process
        oracle, args, _, _ = _denoiser_case()
        g = torch.Generator().manual_seed(200 + rank)
        lat, pl = torch.randn(args[0].shape, generator=g), torch.randn(args[1].shape, generator=g)
        s, ps = oracle(lat, pl, *args[2:])
        (s.square().mean() + ps.square().mean()).backward()
        singles.append({k: p.grad for k, p in oracle.named_parameters() if "lora" in k or k.startswith("cp_blocks")})
    assert len(got[0]) == 603 and set(got[0]) == set(singles[0])
    for k in singles[0]:
        want = (singles[0][k] + singles[1][k]) / 2
        assert torch.equal(got[0][k], got[1][k])
        assert rel_l2(got[0][k], want) < 1e-4, (k, rel_l2(got[0][k], want))


def test_lora_on_a_subset_of_the_projections(fake_denoiser_backend):
    """LoRA groups with holes: only to_q / to_v of the self-attentions and only to_k of the text attentions carry matrices
    (the stacked down / block-diagonal up operands of train_engine.LoRAGroup must keep every pair in its own rows and columns)."""
    from panfusion_amd.models.pano import MultiViewBaseModel
    oracle, args, w_s, w_p = _denoiser_case()
    kept = 0
    for unet in (oracle.unet, oracle.pano_unet):
        for mod in unet.modules():
            if all(hasattr(mod, a) for a in ("to_q", "to_k", "to_v", "to_out")):
                is_self = mod.to_k.in_features == mod.to_q.in_features
                keep = {"to_q", "to_v"} if is_self else {"to_k"}
                for name, lin in (("to_q", mod.to_q), ("to_k", mod.to_k), ("to_v", mod.to_v), ("to_out", mod.to_out[0])):
                    if name not in keep:
                        lin.lora_layer = None
                    else:
                        kept += 1
    s, ps = oracle(*args)
    ((s * w_s).sum() + (ps * w_p).sum()).backward()
    want = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None and ("lora" in k or k.startswith("cp_blocks"))}
    for p in oracle.parameters():
        p.grad = None
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="fast", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    assert len(hip.trainable_tensors()) == 91 + 2 * kept
    s2, ps2 = hip(*args)
    assert rel_l2(s2, s) < 2e-5
    ((s2 * w_s).sum() + (ps2 * w_p).sum()).backward()
    got = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    assert set(k for k in got if "lora" in k) == set(k for k in want if "lora" in k) and len(want) == 91 + 2 * kept
    for k, w in want.items():
        assert rel_l2(got[k], w) < 1e-4, (k, rel_l2(got[k], w))


def test_denoiser_training_batch_of_two_with_different_cameras(fake_denoiser_backend):
    """b = 2 with DIFFERENT camera sets per sample (per-sample EPA tables: one attention launch per sample in recompute and
    backward) and different timesteps: gradients against autograd through the oracle."""
    from conftest import build_tiny_oracle, golden
    from panfusion_amd.models.pano import MultiViewBaseModel
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    c = cam4()
    cams = {"FoV": torch.stack([c["FoV"], c["FoV"]]), "theta": torch.stack([c["theta"], c["theta"] + 33.0]),
            "phi": torch.stack([c["phi"], c["phi"] - 5.0])}
    args = (t("latents"), t("pano_latent"), torch.tensor([[981] * 4, [301] * 4]), t("prompt_embd"), t("pano_prompt_embd"), cams)
    gen = torch.Generator().manual_seed(2)
    w_s, w_p = torch.randn(args[0].shape, generator=gen), torch.randn(args[1].shape, generator=gen)
    oracle = build_tiny_oracle()
    s, ps = oracle(*args)
    ((s * w_s).sum() + (ps * w_p).sum()).backward()
    want = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None and ("lora" in k or k.startswith("cp_blocks"))}
    for p in oracle.parameters():
        p.grad = None
    hip = MultiViewBaseModel(oracle.unet, oracle.pano_unet, None, None, oracle.pano_pad, compute_dtype=torch.float32,
                             precision="fast", differentiable=True)
    hip.load_state_dict({k: v for k, v in oracle.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    s2, ps2 = hip(*args)
    assert rel_l2(s2, s) < 2e-5 and rel_l2(ps2, ps) < 2e-5
    ((s2 * w_s).sum() + (ps2 * w_p).sum()).backward()
    got = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    assert len(want) == 603
    for k, w in want.items():
        assert rel_l2(got[k], w) < 1e-4, (k, rel_l2(got[k], w))


def test_pano_only_model_trains(fake_denoiser_backend):
    """PanoOnly (PanoOnly.py:13,39-41: unet = None, latents = None, cameras = None): the panorama branch alone under training --
    gradients of the panorama UNet's LoRA matrices; there are no EPA blocks to train."""
    from conftest import build_tiny_oracle, golden
    from oracle import mvgen as MV
    from panfusion_amd.models.pano import MultiViewBaseModel
    g = golden("mvgen_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    full = build_tiny_oracle()
    oracle = MV.DualBranchDenoiser(None, full.pano_unet, None, None, True)
    args = (None, t("pano_latent")[:1], torch.tensor([481]), None, t("pano_prompt_embd")[:1], None)
    w_p = torch.randn(args[1].shape, generator=torch.Generator().manual_seed(4))
    _, ps = oracle(*args)
    (ps * w_p).sum().backward()
    want = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None and "lora" in k}
    for p in oracle.parameters():
        p.grad = None
    hip = MultiViewBaseModel(None, full.pano_unet, None, None, True, compute_dtype=torch.float32, precision="fast", differentiable=True)
    assert len(hip.trainable_tensors()) == 256
    none, ps2 = hip(*args)
    assert none is None and rel_l2(ps2, ps) < 2e-5
    (ps2 * w_p).sum().backward()
    got = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    assert len(want) == 256
    for k, w in want.items():
        assert rel_l2(got[k], w) < 1e-4, (k, rel_l2(got[k], w))
