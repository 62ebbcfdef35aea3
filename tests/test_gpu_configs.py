"""Parity at the EXACT BASELINE.json configurations: the HIP path in its default
(fp16 mixed) scheme against outputs of the fp32 CPU oracle generated in the build container by
tools/make_golden_cfg.py and committed under tests/golden/.  Weights and inputs are rebuilt here from the same
seeds (oracle/fixtures.py: CPU generators, machine independent) -- the oracle forward itself (minutes of CPU at
these sizes) does not run on the GPU box.

Tolerances (written here, as the north_star asks): <= 1e-3 rel-L2 on both epsilon outputs of one denoiser call
(cfg 2 headline config with all 20 views and the CFG pair -- at the loop's first call and at a LATE call, t = 21 with
180 degrees of accumulated rotation --, cfg 4 and cfg 5 with the CFG pair, as bench.py runs them); the 10-step DDIM
trajectory of cfg 1 carries the CFG merge u + 9 (c - u), which amplifies the two calls' 8e-4 by ~12 relative to
epsilon before the update scales it back: measured 1.0e-3 / 9.3e-4 after ONE step and flat from there on, gated at
1.2e-3 (step 1) and 1.3e-3 (step 10) (drift per step printed with -s).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _have(name):
    return os.path.exists(os.path.join(GOLDEN, name))


def _hip_model(om, controlnet=False):
    from panfusion_amd.models.pano import MultiViewBaseModel
    model = MultiViewBaseModel(om.unet, om.pano_unet, None, om.pano_cn if controlnet else None, True,
                               compute_dtype=torch.float16)                       # default scheme: fp16 mixed
    model.load_state_dict({k: v for k, v in om.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    assert model.precision == "mixed"
    return model


def _call(model, args, **extra):
    d = lambda x: x.to(DEV)
    return model(d(args["latents"]), d(args["pano_latent"]), d(args["timestep"]), d(args["prompt_embd"]),
                 d(args["pano_prompt_embd"]), args["cameras"], **extra)


@pytest.fixture(scope="module")
def full_oracle():
    from oracle import fixtures as FX
    return FX.build_full_width()


def test_cfg2_headline_config_vs_oracle(full_oracle):
    """BASELINE.json configs[1] exactly as bench.py runs it: m = 20 icosahedron views of 64x64 latents + the 64x128
    panorama latent, the CFG pair (b = 2), SD-2-base widths.  The 20-view EPA (K = 20 480 keys) is compared
    after every one of the 7 blocks on 8-channel slices, then both epsilon outputs at <= 1e-3."""
    from oracle import fixtures as FX
    from panfusion_amd.models.pano.modules import WarpAttn
    gd = np.load(os.path.join(GOLDEN, "cfg2_eps.npz"))
    model = _hip_model(full_oracle)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=True)
    got = {}
    blocks = [*model.cp_blocks_encoder, model.cp_blocks_mid, *model.cp_blocks_decoder]
    index = {id(b): i for i, b in enumerate(blocks)}
    saved = WarpAttn.forward_nhwc

    def rec(self, xp, xe, groups, m, **kw):
        op, oe = saved(self, xp, xe, groups, m, **kw)
        i = index[id(self)]
        st = op.shape[-1] // 8
        got["epa%d_pers" % i] = op[-m, :, :, ::st].float().permute(2, 0, 1).cpu()     # view 0 of the conditional sample
        got["epa%d_pano" % i] = oe[-1, :, :, ::st].float().permute(2, 0, 1).cpu()
        return op, oe
    WarpAttn.forward_nhwc = rec
    try:
        s, ps = _call(model, args)
    finally:
        WarpAttn.forward_nhwc = saved
    es, ep = rel_l2(s.cpu(), torch.from_numpy(gd["sample"])), rel_l2(ps.cpu(), torch.from_numpy(gd["pano_sample"]))
    print("\ncfg2 (m=20, CFG pair) rel-L2 vs oracle: views %.3e  pano %.3e   (tolerance 1e-3)" % (es, ep))
    for i in range(7):
        a, b = rel_l2(got["epa%d_pers" % i], torch.from_numpy(gd["epa%d_pers" % i])), \
            rel_l2(got["epa%d_pano" % i], torch.from_numpy(gd["epa%d_pano" % i]))
        print("  after EPA block %d: views %.2e  pano %.2e" % (i, a, b))
        # per-block error budget: measured 4.5e-4 ... 6.8e-4 on these 8-channel slices of the intermediate streams;
        # a regression is caught at the block where it enters, before it flips the 1e-3 gate on the outputs (round 5: 2e-3)
        assert a <= 8e-4 and b <= 8e-4, (i, a, b)
    # per CFG half too: the null-prompt half and the prompted half separately
    for h in (0, 1):
        assert rel_l2(s[h].cpu(), torch.from_numpy(gd["sample"][h])) <= 1e-3
        assert rel_l2(ps[h].cpu(), torch.from_numpy(gd["pano_sample"][h])) <= 1e-3
    assert es <= 1e-3 and ep <= 1e-3, (es, ep)
    # ... and against the fixture the REFERENCE CLASS itself produced for the conditional half (tools/make_golden_cfg.py cfg2ref:
    # models/pano/MVGenModel.py:38-297 with its own WarpAttn / get_masks / dense per-head bias, 3.0e-6 / 2.5e-6 from the port's file)
    if _have("cfg2_ref_cond.npz"):
        gr = np.load(os.path.join(GOLDEN, "cfg2_ref_cond.npz"))
        rv, rp = rel_l2(s[1:].cpu(), torch.from_numpy(gr["sample"])), rel_l2(ps[1:].cpu(), torch.from_numpy(gr["pano_sample"]))
        print("  conditional half vs the reference class's own output: views %.3e  pano %.3e" % (rv, rp))
        assert rv <= 1e-3 and rp <= 1e-3, (rv, rp)


@pytest.mark.skipif(not _have("cfg2b_eps.npz"), reason="fixture not generated")
def test_cfg2_late_step_second_rotation_vs_oracle(full_oracle):
    """The headline configuration again (m = 20 x CFG pair) at a LATE iteration of the loop: timestep t = 21, cameras and
    panorama at 180 degrees of accumulated rotation (another table set, another region of the timestep embedding)."""
    from oracle import fixtures as FX
    gd = np.load(os.path.join(GOLDEN, "cfg2b_eps.npz"))
    model = _hip_model(full_oracle)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=True, t=21, rot=180.0)
    s, ps = _call(model, args)
    es, ep = rel_l2(s.cpu(), torch.from_numpy(gd["sample"])), rel_l2(ps.cpu(), torch.from_numpy(gd["pano_sample"]))
    print("\ncfg2 late step (t=21, 180 deg; m=20, CFG pair) rel-L2 vs oracle: views %.3e  pano %.3e   (tolerance 1e-3)" % (es, ep))
    for h in (0, 1):
        assert rel_l2(s[h].cpu(), torch.from_numpy(gd["sample"][h])) <= 1e-3
        assert rel_l2(ps[h].cpu(), torch.from_numpy(gd["pano_sample"][h])) <= 1e-3
    assert es <= 1e-3 and ep <= 1e-3, (es, ep)
    _against_reference_class("cfg2b_ref_cond.npz", s, ps)


@pytest.mark.parametrize("graphs", [True])
def test_cfg1_ten_ddim_steps_vs_oracle(full_oracle, graphs):
    """BASELINE.json configs[0]: m = 4 views of 256^2 (32x32 latents) + the 512x1024 panorama, 10 DDIM steps at
    SD-2-base widths through DenoiseLoop; latents after EVERY step against the oracle loop (oracle/ddim.py)."""
    from oracle import fixtures as FX
    from panfusion_amd import ops
    from panfusion_amd.pipeline import DenoiseLoop
    gd = np.load(os.path.join(GOLDEN, "cfg1_ddim10.npz"))
    model = _hip_model(full_oracle)
    cams = FX.horizon4_cameras()
    latents, pano_latent, pe, ppe = FX.loop_inputs(cams, (32, 32), (64, 128))
    loop = DenoiseLoop(model, latents.to(DEV), pano_latent.to(DEV), pe.to(DEV), ppe.to(DEV), cams, steps=10, use_graphs=graphs)
    drift = []
    for i in range(10):
        loop.step()
        # loop.pano is kept in the frame of the accumulated rotation (already rolled for the NEXT step, except after the
        # last one, and total_rot counts exactly what it carries): undo it, as DenoiseLoop.result does
        pano = ops.roll_width(loop.pano, int(-loop.total_rot / 360 * loop.W))
        drift.append((rel_l2(loop.lat.cpu(), torch.from_numpy(gd["latents"][i])),
                      rel_l2(pano.cpu(), torch.from_numpy(gd["pano_latent"][i]))))
    print("\ncfg1 10-step DDIM drift (views / pano rel-L2 per step):")
    print("  " + "  ".join("%d: %.2e/%.2e" % (i + 1, a, b) for i, (a, b) in enumerate(drift)))
    lat, pano = loop.result()
    el, ep = rel_l2(lat.cpu(), torch.from_numpy(gd["latents"][-1])), rel_l2(pano.cpu(), torch.from_numpy(gd["pano_latent"][-1]))
    print("  final: views %.3e  pano %.3e  (tolerance 1.0e-3 at every step)" % (el, ep))
    # One step already carries the CFG merge: eps = u + 9 (c - u) amplifies the two calls' error by ~12 relative to eps, the DDIM update
    # scales it back by the step's eps coefficient -- flat from step 2 on.  Round 6: the gate is north_star's 1e-3 at
    # EVERY step for views AND panorama (rounds 3-5: 1.2e-3 / 1.3e-3 around a measured 1.02e-3): the level-0 upsampling convolution in split
    # precision and the sub-pixel form of all three brought the trajectory to 9.05e-4 / 8.72e-4 at its worst step
    # (profiles/r6_final_configs_parity.log).
    for i, (a, b) in enumerate(drift):
        assert a <= 1.0e-3 and b <= 1.0e-3, (i + 1, a, b)
    assert el <= 1.0e-3 and ep <= 1.0e-3, (el, ep)


STRESS_TOL = 1e-3


def _against_reference_class(name, s, ps):
    """The conditional half of the CFG pair against the fixture the REFERENCE's own MultiViewBaseModel produced for it
    (tools/make_golden_cfg.py cfg4ref / cfg5ref; the port equals it to fp32 round-off: tests/test_oracle_golden.py)."""
    if not _have(name):
        return
    gr = np.load(os.path.join(GOLDEN, name))
    rv, rp = rel_l2(s[1:].cpu(), torch.from_numpy(gr["sample"])), rel_l2(ps[1:].cpu(), torch.from_numpy(gr["pano_sample"]))
    print("  conditional half vs the reference class's own output (%s): views %.3e  pano %.3e" % (name, rv, rp))
    assert rv <= 1e-3 and rp <= 1e-3, (name, rv, rp)


@pytest.mark.skipif(not _have("cfg1_stress_eps.npz"), reason="fixture not generated")
def test_cfg1_range_stress_vs_oracle():
    """RANGE STRESS: the fan-in-scaled synthetic weights keep every residual stream at O(1); here the layers
    that write into the streams carry per-channel output scales drawn log-uniformly over 3 decades (oracle/fixtures.apply_range_stress),
    so the fp32 oracle's stream tensors peak at ~1e4 (recorded in the fixture) and a GroupNorm group is dominated by its outlier
    channel.  fp16 operands have 5 exponent bits: the default path must stay finite and within the same 1e-3 of the fp32 oracle
    (reference class as denoiser, tools/make_golden_cfg.py cfg1s) on cfg 1's first denoiser call, CFG pair, SD-2-base widths."""
    from oracle import fixtures as FX
    gd = np.load(os.path.join(GOLDEN, "cfg1_stress_eps.npz"))
    assert float(gd["stream_peaks"].max()) >= 1e3, "the fixture must actually stress the range"
    om = FX.build_full_width()
    FX.apply_range_stress(om)
    model = _hip_model(om)
    args = FX.first_step_call(FX.horizon4_cameras(), (32, 32), (64, 128), cfg_pair=True)
    s, ps = _call(model, args)
    assert bool(torch.isfinite(s).all()) and bool(torch.isfinite(ps).all()), "non-finite epsilon under range stress"
    es, ep = rel_l2(s.cpu(), torch.from_numpy(gd["sample"])), rel_l2(ps.cpu(), torch.from_numpy(gd["pano_sample"]))
    print("\ncfg1 RANGE STRESS (stream peaks %.2g in the fp32 oracle) rel-L2 vs oracle: views %.3e  pano %.3e  (tolerance %.0e)"
          % (float(gd["stream_peaks"].max()), es, ep, STRESS_TOL))
    assert es <= STRESS_TOL and ep <= STRESS_TOL, (es, ep)


@pytest.mark.skipif(not _have("cfg4_eps.npz"), reason="fixture not generated")
def test_cfg4_large_panorama_vs_oracle(full_oracle):
    """BASELINE.json configs[3]: 1024x2048 panorama (128x256 latent, 32 768 self-attention tokens) + 20 views, the CFG pair."""
    from oracle import fixtures as FX
    gd = np.load(os.path.join(GOLDEN, "cfg4_eps.npz"))
    model = _hip_model(full_oracle)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (128, 256), cfg_pair=True)
    s, ps = _call(model, args)
    es, ep = rel_l2(s.cpu(), torch.from_numpy(gd["sample"])), rel_l2(ps.cpu(), torch.from_numpy(gd["pano_sample"]))
    print("\ncfg4 (128x256 pano latent) rel-L2 vs oracle: views %.3e  pano %.3e" % (es, ep))
    assert es <= 1e-3 and ep <= 1e-3, (es, ep)
    _against_reference_class("cfg4_ref_cond.npz", s, ps)


@pytest.mark.skipif(not _have("cfg5_eps.npz"), reason="fixture not generated")
def test_cfg5_layout_controlnet_vs_oracle():
    """BASELINE.json configs[4]: cfg 2's geometry + the panorama ControlNet at SD-2-base widths on a 512x1024 layout image,
    the CFG pair (the condition image duplicated as gen_cls_free_guide_pair does, PanoGenerator.py:240-251)."""
    from oracle import fixtures as FX
    gd = np.load(os.path.join(GOLDEN, "cfg5_eps.npz"))
    om = FX.build_full_width(controlnet=True)
    model = _hip_model(om, controlnet=True)
    args = FX.first_step_call(FX.ico_cameras(), (64, 64), (64, 128), cfg_pair=True)
    cond = torch.cat([torch.roll(FX.layout_image((64, 128)), 1024 // 4, dims=-1)] * 2)
    s, ps = _call(model, args, pano_layout_cond=cond.to(DEV))
    es, ep = rel_l2(s.cpu(), torch.from_numpy(gd["sample"])), rel_l2(ps.cpu(), torch.from_numpy(gd["pano_sample"]))
    print("\ncfg5 (panorama ControlNet) rel-L2 vs oracle: views %.3e  pano %.3e" % (es, ep))
    assert es <= 1e-3 and ep <= 1e-3, (es, ep)
    _against_reference_class("cfg5_ref_cond.npz", s, ps)
