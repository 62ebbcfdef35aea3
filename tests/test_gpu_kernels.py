"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle / a plain
fp32 torch statement of the same op, on seeded inputs.  Needs an MI355X: `-m gpu`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import cam4, golden, rel_l2, unpair
from oracle import ddim as oddim
from oracle import geometry as G
from oracle import sd2_unet as U

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.bfloat16, torch.float16]
# tolerance of ONE 16-bit rounding of the output (+ fp32 accumulation order): bf16 has 8
# significand bits (rel. rounding error <= 2^-9), fp16 has 11 (<= 2^-12)
TOL = {torch.bfloat16: 4e-3, torch.float16: 6e-4}


def ops():
    from panfusion_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q16(x, dtype):
    """Round to the 16-bit storage type and return (device tensor, fp32 CPU copy of the rounded values)."""
    d = x.to(dtype)
    return d.to(DEV), d.float()


def check(name, got, want, tol):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(want.shape))
    assert torch.isfinite(got).all(), "%s: non-finite output" % name
    err = rel_l2(got, want)
    assert err <= tol, "%s: rel-L2 %.3e > %.1e (max abs %.3e)" % (name, err, tol, float((got - want).abs().max()))


def ico():
    th, ph = G.icosahedron_cameras()
    return np.degrees(th), np.degrees(ph)


# ------------------------------------------------------------------------------------ geometry
@pytest.mark.parametrize("rot", [0, 90, 180, 270])
def test_e2p_nearest_indices_bit_exact(rot):
    """north_star: the e2p gather indices are bit-exact (all 20 benchmark cameras, 4 rotations)."""
    g = golden("grids.npz")
    thd, phd = g["theta"], g["phi"]
    for name, (eh, ew, h, w) in {"64": (64, 128, 64, 64), "8": (8, 16, 8, 8)}.items():
        mx, my = ops().e2p_grid([90] * 20, (thd + rot) % 360, phd, eh, ew, h, w, DEV)
        idx = ops().nearest_indices(mx, my, eh, ew).cpu().numpy()
        want = g["idx_%s_rot%d" % (name, rot)].astype(np.int32)
        assert np.array_equal(idx, want), "%d index mismatches at %s rot %d" % ((idx != want).sum(), name, rot)


def _sha(t):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("rot", [0, 90, 180, 270])
def test_p2e_grid_and_indices_bit_exact(rot):
    """north_star: the p2e grid computation is bit-exact too -- mask, float32 (u, v) maps (SHA-256 of the
    reference's maps, p2e.py:9-49,66-67) and the nearest-neighbour gather indices, 20 benchmark cameras x 4
    rotation offsets at 64x64 -> 64x128 and 8x8 -> 8x16 (fixture: tools/make_golden_grids.py)."""
    g = golden("grids_r2.npz")
    thd, phd = g["theta"], g["phi"]
    for name, (vh, vw, H, W) in {"64": (64, 64, 64, 128), "8": (8, 8, 8, 16)}.items():
        key = "%s_rot%d" % (name, rot)
        mu, mv, mask = ops().p2e_grid([90] * 20, (thd + rot) % 360, phd, vh, vw, H, W, DEV)
        want_mask = np.unpackbits(g["p2e_mask_" + key])[:20 * H * W].reshape(20, H, W).astype(bool)
        assert np.array_equal(mask.cpu().numpy().astype(bool), want_mask), key
        if name == "8":
            assert np.array_equal(mu.cpu().numpy(), g["p2e_u_" + key]) and np.array_equal(mv.cpu().numpy(), g["p2e_v_" + key])
        if not (np.array_equal(_sha(mu), g["p2e_u_sha_" + key]) and np.array_equal(_sha(mv), g["p2e_v_sha_" + key])):
            wu = np.stack([G.p2e_grid(vh, vw, 90, (thd[i] + rot) % 360, phd[i], H, W)[0] for i in range(20)]).astype(np.float32)
            bad = mu.cpu().numpy() != wu
            raise AssertionError("%s: %d of %d u values differ from the oracle's (max abs %.3e)"
                                 % (key, bad.sum(), bad.size, np.abs(mu.cpu().numpy() - wu).max()))
        idx = ops().nearest_indices(mu, mv, vh, vw).cpu().numpy()
        assert np.array_equal(idx, g["p2e_idx_" + key].astype(np.int32)), "%d p2e index mismatches at %s" % (
            (idx != g["p2e_idx_" + key]).sum(), key)


@pytest.mark.parametrize("rot", [0, 90, 180, 270])
def test_e2p_nearest_indices_bit_exact_cfg4(rot):
    """BASELINE.json configs[3]: 128x256 panorama latent -> 64x64 views, all 20 cameras."""
    g = golden("grids_r2.npz")
    mx, my = ops().e2p_grid([90] * 20, (g["theta"] + rot) % 360, g["phi"], 128, 256, 64, 64, DEV)
    idx = ops().nearest_indices(mx, my, 128, 256).cpu().numpy()
    want = g["e2p_idx_cfg4_rot%d" % rot]
    assert np.array_equal(idx, want), "%d index mismatches at rot %d" % ((idx != want).sum(), rot)


def test_e2p_p2e_grids_vs_oracle():
    thd, phd = ico()
    th = (thd + 90) % 360
    mx, my, ll = ops().e2p_grid([90] * 20, th, phd, 16, 32, 16, 16, DEV, want_lonlat=True)
    g = golden("grids.npz")
    want = torch.from_numpy(g["e2p_maps_16"]).float()           # float64 -> float32 like e2p.py:74-75
    assert float((mx.cpu() - want[:, 0]).abs().max()) <= 4e-6 and float((my.cpu() - want[:, 1]).abs().max()) <= 4e-6
    mu, mv, mask = ops().p2e_grid([90] * 20, th, phd, 16, 16, 16, 32, DEV)
    assert np.array_equal(mask.cpu().numpy().astype(bool), g["p2e_mask_16"])
    assert torch.equal(mu.cpu(), torch.from_numpy(g["p2e_u_16"]).float())      # bit-exact (see the test above)
    assert torch.equal(mv.cpu(), torch.from_numpy(g["p2e_v_16"]).float())
    cams = {"FoV": torch.full((20,), 90), "theta": torch.tensor(th), "phi": torch.tensor(phd)}
    pc, ec = G.get_coords(16, 16, 16, 32, cams)
    assert float((ll.cpu() - pc).abs().max()) <= 1e-6
    assert float((ops().equi_coords(16, 32, DEV).cpu() - ec).abs().max()) <= 1e-6


@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
def test_remap_e2p_p2e_vs_oracle(mode):
    from panfusion_amd.external.Perspective_and_Equirectangular import e2p, p2e
    thd, phd = ico()
    cams = (torch.full((20,), 90), torch.tensor(thd), torch.tensor(phd))
    x = rnd(20, 3, 16, 32, seed=1)
    got = e2p(x.to(DEV), *cams, (16, 16), mode=mode)
    want = G.e2p(x, *cams, (16, 16), mode=mode)
    if mode == "nearest":
        assert torch.equal(got.cpu(), want)
    else:
        check("e2p bilinear", got, want, 1e-5)
    y = rnd(20, 3, 16, 16, seed=2)
    ge, gm = p2e(y.to(DEV), *cams, (16, 32), mode=mode)
    we, wm = G.p2e(y, *cams, (16, 32), mode=mode)
    assert torch.equal(gm.cpu(), wm)
    if mode == "nearest":
        assert torch.equal(ge.cpu(), we)                     # pure gather * mask: exact
    else:
        check("p2e " + mode, ge, we, 1e-5)
    # scalar camera broadcast
    assert torch.equal(e2p(x.to(DEV), 90, 30.0, 10.0, (8, 8), mode="nearest").cpu(),
                       G.e2p(x, 90, 30.0, 10.0, (8, 8), mode="nearest"))


def test_init_noise_golden():
    from panfusion_amd.pipeline import init_noise
    g, gn = golden("grids.npz"), golden("init_noise.npz")
    pano_noise = torch.randn(1, 1, 4, 64, 128, generator=torch.Generator().manual_seed(0))
    cams = {"FoV": torch.full((1, 20), 90), "theta": torch.tensor(g["theta"])[None], "phi": torch.tensor(g["phi"])[None]}
    _, views = init_noise(pano_noise.to(DEV), cams, 64, 64)
    assert np.array_equal(views[0, :, 0].cpu().numpy(), gn["view_noise_c0"])      # pure gather: exact


def test_spherical_pe_golden():
    ge = golden("epa_tables.npz")
    for n, ck, pk in ((80, "pers_coords", "pe80_pers"), (80, "equi_coords", "pe80_equi"), (320, "equi_coords", "pe320_equi")):
        coords = torch.from_numpy(ge[ck]).to(DEV)
        freq = torch.from_numpy(ge["freq%d" % n]).to(DEV)
        got = ops().spherical_pe(coords, freq).cpu()
        want = torch.from_numpy(ge[pk])
        # sin/cos of identical fp32 arguments: full-range accurate implementations agree to a few ulp
        assert float((got - want).abs().max()) <= 2e-6, (n, ck, float((got - want).abs().max()))


def test_epa_tables_golden_and_flags():
    g, ge = golden("grids.npz"), golden("epa_tables.npz")
    th = (g["theta"] + 90) % 360
    be, bp, fe, fp = ops().epa_tables([90] * 20, th, g["phi"], 8, 8, 8, 16, DEV)
    E, P, m = 128, 64, 20
    pers = torch.from_numpy(ge["pers_masks"]).reshape(m, E, P).permute(1, 0, 2).reshape(E, m * P) + 1
    equi = torch.from_numpy(ge["equi_masks"]).reshape(m * P, E) + 1
    assert float((be.cpu() - pers).abs().max()) <= 2e-5, float((be.cpu() - pers).abs().max())
    assert float((bp.cpu() - equi).abs().max()) <= 2e-5, float((bp.cpu() - equi).abs().max())
    # (supports may differ by entries of magnitude < 2e-5: a bilinear weight that is exactly 0 on one side)
    for bias, flags in ((be, fe), (bp, fp)):
        nq, nk = bias.shape
        tiles = bias.cpu().reshape(nq // 32, 32, nk // 32, 32).abs().amax((1, 3)) > 0
        assert torch.equal(flags.cpu().bool(), tiles)
    # second camera set (m=4, 16x16 / 16x32), stored sparsely
    c = cam4()
    be, bp, _, _ = ops().epa_tables(c["FoV"], c["theta"], c["phi"], 16, 16, 16, 32, DEV)
    pm = torch.zeros(4 * 512 * 256)
    pm[torch.from_numpy(ge["m4_pers_idx"]).long()] = torch.from_numpy(ge["m4_pers_val"])
    pm = pm.reshape(4, 512, 256).permute(1, 0, 2).reshape(512, 1024)
    assert float((be.cpu() - pm).abs().max()) <= 2e-5
    em = torch.zeros(4 * 256 * 512)
    em[torch.from_numpy(ge["m4_equi_idx"]).long()] = torch.from_numpy(ge["m4_equi_val"])
    assert float((bp.cpu() - em.reshape(1024, 512)).abs().max()) <= 2e-5


def test_epa_tables_properties_at_benchmark_size():
    """cfg-2 scale s=2: 20 views of 32x32 against a 32x64 panorama (oracle would need minutes)."""
    thd, phd = ico()
    be, bp, fe, fp = ops().epa_tables([90] * 20, (thd + 90) % 360, phd, 32, 32, 32, 64, DEV)
    E, P, m = 2048, 1024, 20
    assert float(be.min()) >= 0 and float(be.max()) <= 2.0 + 1e-6
    per_view_max = be.view(E, m, P).amax(2)
    assert bool(((per_view_max - 2).abs().le(1e-6) | (per_view_max == 0)).all())     # max +1 or constant -1
    rowmax = bp.amax(1)
    assert bool((rowmax - 2).abs().le(1e-6).all())          # every view pixel lands somewhere on the sphere
    frac = float((be > 0).float().mean())
    assert 0.005 < frac < 0.02, frac                         # ~1.0 %
    seen = (per_view_max > 0).sum(1)
    assert int(seen.min()) >= 3 and int(seen.max()) <= 7
    assert 0.03 < float(fe.float().mean()) < 0.5


def test_get_masks_api_layout():
    from panfusion_amd.models.pano import get_coords, get_masks
    ge, g = golden("epa_tables.npz"), golden("grids.npz")
    cams = {"FoV": torch.full((20,), 90), "theta": torch.tensor((g["theta"] + 90) % 360), "phi": torch.tensor(g["phi"])}
    pm, em = get_masks(8, 8, 8, 16, cams, DEV)
    assert pm.shape == (20, 8, 16, 8, 8) and em.shape == (20, 8, 8, 8, 16)
    assert float((pm.cpu() - torch.from_numpy(ge["pers_masks"])).abs().max()) <= 2e-5
    assert float((em.cpu() - torch.from_numpy(ge["equi_masks"])).abs().max()) <= 2e-5
    pc, ec = get_coords(8, 8, 8, 16, cams, DEV)
    assert float((pc.cpu() - torch.from_numpy(ge["pers_coords"])).abs().max()) <= 1e-6
    assert float((ec.cpu() - torch.from_numpy(ge["equi_coords"])).abs().max()) <= 1e-6


# ------------------------------------------------------------------------------------ norms / pointwise
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(3, 40, 64, 32, 0), (2, 300, 320, 32, 0), (2, 70, 192, 32, 128), (1, 64, 2560, 32, 1280)])
def test_groupnorm_silu(dtype, shape):
    n, hw, C, groups, c1 = shape
    x, xf = q16(rnd(n, hw, C, seed=3) * 2 + 0.5, dtype)
    gam, bet = rnd(C, seed=4) * 0.2 + 1, rnd(C, seed=5) * 0.1
    x0 = x[..., :C - c1].contiguous()
    x1 = x[..., C - c1:].contiguous() if c1 else None
    sc, sh = ops().groupnorm_scale_shift(x0, x1, n, hw, groups, 1e-5, gam.to(DEV), bet.to(DEV))
    y = ops().scale_shift_act(x0, x1, n, hw, sc, sh, 1)
    want = F.silu(F.group_norm(xf.permute(0, 2, 1), groups, gam, bet, 1e-5)).permute(0, 2, 1)
    check("groupnorm+silu", y, want, TOL[dtype])
    y0 = ops().scale_shift_act(x0, x1, n, hw, sc, sh, 0)
    check("groupnorm", y0, F.group_norm(xf.permute(0, 2, 1), groups, gam, bet, 1e-5).permute(0, 2, 1), TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [64, 320, 1280])
def test_layernorm_with_pe(dtype, C):
    rows, pe_rows = 37 * 3, 37
    x, xf = q16(rnd(rows, C, seed=6) + 0.3, dtype)
    pe = rnd(pe_rows, C, seed=7)
    gam, bet = rnd(C, seed=8) * 0.2 + 1, rnd(C, seed=9) * 0.1
    y = ops().layernorm(x, gam.to(DEV), bet.to(DEV), 1e-5, pe=pe.to(DEV))
    check("layernorm+pe", y, F.layer_norm(xf + pe.repeat(3, 1), (C,), gam, bet, 1e-5), TOL[dtype])
    y = ops().layernorm(x, gam.to(DEV), bet.to(DEV), 1e-5)
    check("layernorm", y, F.layer_norm(xf, (C,), gam, bet, 1e-5), TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
def test_pointwise_ops(dtype):
    o = ops()
    x, xf = q16(rnd(50, 2 * 256, seed=10) * 2, dtype)
    a, g = xf.chunk(2, -1)
    check("geglu", o.geglu(x), a * F.gelu(g), TOL[dtype])
    check("silu", o.silu(x), F.silu(xf), TOL[dtype])
    y, yf = q16(rnd(50, 512, seed=11), dtype)
    check("add", o.add(x, y), xf + yf, TOL[dtype])
    t = torch.tensor([981, 961, 1, 0, 500])
    check("timestep features", o.timestep_features(t.to(DEV), 320, dtype), U.Timesteps(320)(t), TOL[dtype])
    z, zf = q16(rnd(2, 5, 16, 64, seed=12), dtype)                  # NHWC
    want = G.pad_pano(zf.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(o.pad_width(z, 2).float().cpu(), want)
    assert torch.equal(o.crop_width(o.pad_width(z, 2), 2).float().cpu(), zf)
    assert torch.equal(o.crop_width(z, 1).float().cpu(), zf[:, :, 1:-1])
    n = rnd(2, 6, 5, 16, seed=13)
    assert torch.equal(o.nchw_to_nhwc(n.to(DEV), dtype).float().cpu(), n.to(dtype).float().permute(0, 2, 3, 1))
    assert torch.equal(o.nhwc_to_nchw(z, torch.float32).cpu(), zf.permute(0, 3, 1, 2))


def test_pad_pano_and_roll_api():
    from panfusion_amd.utils.pano import pad_pano, unpad_pano
    for x in (rnd(2, 3, 5, 16, seed=14), rnd(2, 2, 3, 5, 16, seed=15)):
        for dt in (torch.float32, torch.float16):
            xx = x.to(dt)
            assert torch.equal(pad_pano(xx.to(DEV), 2).cpu(), G.pad_pano(xx, 2))
            assert torch.equal(unpad_pano(pad_pano(xx.to(DEV), 3), 3).cpu(), xx)
    x = rnd(3, 4, 7, 32, seed=16)
    for s in (8, -8, 0, 40, -1600):
        assert torch.equal(ops().roll_width(x.to(DEV), s).cpu(), torch.roll(x, s, -1))


def test_cfg_ddim_step_vs_oracle():
    sched = oddim.DDIM()
    sched.set_timesteps(50)
    x, eu, ec = rnd(20, 4, 16, 32, seed=17), rnd(20, 4, 16, 32, seed=18), rnd(20, 4, 16, 32, seed=19)
    for t in (981, 501, 1):
        want = sched.step(oddim.cfg_merge(torch.cat([eu, ec]), 9.0), t, x)
        coef = [float(c) for c in sched.coefficients(t)]
        got = ops().cfg_ddim_step(x.to(DEV), eu.to(DEV), ec.to(DEV), 9.0, coef, 0)
        assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
        got = ops().cfg_ddim_step(x.to(DEV), eu.to(DEV), ec.to(DEV), 9.0, coef, 8)
        assert float((got.cpu() - torch.roll(want, 8, -1)).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("shape", [(1, 4, 4, 16, 32), (1, 1, 4, 64, 128), (1, 1, 4, 8, 600)])
def test_cfg_ddim_step_pair_in_place_roll_second_copy_and_timestep(shape):
    """The loop's one-launch state update (pf_cfg_ddim_step_pair): BIT-identical to pf_cfg_ddim_step + roll, in place for any
    roll, second copy for the CFG pair, next timestep written to the int64 tensor (PanFusion.py:147-162)."""
    o = ops()
    sched = oddim.DDIM()
    sched.set_timesteps(50)
    x, eu, ec = rnd(*shape, seed=27), rnd(*shape, seed=28), rnd(*shape, seed=29)
    W = shape[-1]
    for t, roll in ((981, 0), (501, W // 4), (21, -3), (1, 5 * W + 7)):
        coef = [float(c) for c in sched.coefficients(t)]
        want = o.cfg_ddim_step(x.to(DEV), eu.to(DEV), ec.to(DEV), 9.0, coef, roll)
        assert torch.equal(want.cpu(), torch.roll(o.cfg_ddim_step(x.to(DEV), eu.to(DEV), ec.to(DEV), 9.0, coef, 0).cpu(), roll, -1))
        pair = torch.stack([x[0], x[0]]).to(DEV)
        tstep = torch.full((2, 5), 7, dtype=torch.long, device=DEV)
        got = o.cfg_ddim_step_pair(pair[:1], eu.to(DEV), ec.to(DEV), 9.0, coef, roll, out=pair[:1], out2=pair[1:], tstep=tstep, t_next=t - 20)
        assert got.data_ptr() == pair.data_ptr()
        bad = (pair[0] != want[0])
        assert int(bad.sum()) == 0, (t, roll, int(bad.sum()), float((pair[0] - want[0]).abs().max()), bad.nonzero()[:4].tolist())
        assert torch.equal(pair[1].cpu(), want[0].cpu())
        assert torch.equal(tstep.cpu(), torch.full((2, 5), t - 20, dtype=torch.long))
        ref = sched.step(oddim.cfg_merge(torch.cat([eu, ec]), 9.0), t, x)
        assert float((pair[:1].cpu() - torch.roll(ref, roll, -1)).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("wrap", [False, True])
def test_boundary_convs(dtype, wrap):
    o = ops()
    conv = torch.nn.Conv2d(4, 64, 3, padding=1)
    x = rnd(3, 4, 9, 16, seed=20)
    with torch.no_grad():
        want = conv(G.pad_pano(x, 1))[..., 1:-1] if wrap else conv(x)
    y = o.conv_in(x.to(DEV), conv.weight.detach().permute(2, 3, 1, 0).contiguous().to(DEV), conv.bias.detach().to(DEV),
                  64, dtype, wrap=wrap)
    check("conv_in", y.permute(0, 3, 1, 2), want, TOL[dtype])
    conv2 = torch.nn.Conv2d(64, 4, 3, padding=1)
    z, zf = q16(rnd(3, 9, 16, 64, seed=21), dtype)
    with torch.no_grad():
        zi = zf.permute(0, 3, 1, 2)
        want = conv2(G.pad_pano(zi, 1))[..., 1:-1] if wrap else conv2(zi)
    y = o.conv_out(z, conv2.weight.detach().permute(0, 2, 3, 1).contiguous().to(DEV), conv2.bias.detach().to(DEV), 4, wrap=wrap)
    check("conv_out", y, want, 2e-5)


@pytest.mark.parametrize("wrap", [False, True])
@pytest.mark.parametrize("shape", [(3, 9, 16, 64, 4), (2, 8, 32, 96, 4), (2, 19, 70, 32, 3), (1, 64, 64, 320, 4)])
def test_fused_head_conv_out_gn(shape, wrap):
    """pf_conv_out_gn (GroupNorm-apply + SiLU inside conv_out's LDS tile, MVGenModel.py:279-294) against fp64 torch and
    against the two-launch form it replaces; tiles cut by the image edge, circular width, cout 3."""
    o = ops()
    n, h, w, C, cout = shape
    x = (rnd(n, h, w, C, seed=40) * 3 + 0.5).to(DEV)
    conv = torch.nn.Conv2d(C, cout, 3, padding=1)
    sc, sh = (rnd(n, C, seed=41) * 0.2 + 0.4).to(DEV), (rnd(n, C, seed=42) * 0.5).to(DEV)    # (per image and channel, as pf_groupnorm_* writes them)
    wt = o.conv_out_weight_t(conv.weight).to(DEV)
    got = o.conv_out_gn(x, sc, sh, 1, wt, conv.bias.detach().to(DEV), cout, wrap=wrap)
    y = o.scale_shift_act(x, None, n, h * w, sc, sh, 1, out_dtype=torch.float32).view(n, h, w, C)
    two = o.conv_out(y, conv.weight.detach().permute(0, 2, 3, 1).contiguous().to(DEV), conv.bias.detach().to(DEV), cout, wrap=wrap)
    with torch.no_grad():
        yi = F.silu(x.cpu().double() * sc.cpu().double().view(n, 1, 1, C) + sh.cpu().double().view(n, 1, 1, C)).permute(0, 3, 1, 2)
        cd = conv.double()
        want = cd(G.pad_pano(yi, 1))[..., 1:-1] if wrap else cd(yi)
    check("conv_out_gn", got, want.float(), 2e-5)
    check("conv_out_gn vs two launches", got, two.cpu(), 2e-5)
    # no activation: plain scale / shift
    got0 = o.conv_out_gn(x, sc, sh, 0, wt, None, cout, wrap=wrap)
    with torch.no_grad():
        yi0 = (x.cpu().double() * sc.cpu().double().view(n, 1, 1, C) + sh.cpu().double().view(n, 1, 1, C)).permute(0, 3, 1, 2)
        want0 = (cd(G.pad_pano(yi0, 1))[..., 1:-1] if wrap else cd(yi0)) - conv.bias.detach().double().view(1, -1, 1, 1)
    check("conv_out_gn (no act, no bias)", got0, want0.float(), 2e-5)


# ------------------------------------------------------------------------------------ GEMM / conv
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(300, 320, 320), (128, 160, 64), (1000, 640, 1280), (77, 64, 128), (4100, 960, 320), (520, 200, 192)])
def test_linear(dtype, mnk):
    M, N, K = mnk
    x, xf = q16(rnd(M, K, seed=22), dtype)
    w, wf = q16(rnd(N, K, seed=23) / K ** 0.5, dtype)
    b = rnd(N, seed=24)
    r, rf = q16(rnd(M, N, seed=25), dtype)
    check("linear", ops().linear(x, w), xf @ wf.T, TOL[dtype])
    check("linear+bias+res", ops().linear(x, w, bias=b.to(DEV), residual=r), xf @ wf.T + b + rf, TOL[dtype])
    got = ops().linear(x, w, bias=b.to(DEV), out_dtype=torch.float32)
    assert got.dtype == torch.float32
    check("linear fp32 out", got, xf @ wf.T + b, 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(300, 640, 320), (64, 2560, 1280)])
def test_linear_geglu_epilogue(dtype, mnk):
    """FF1 with the gating fused into the epilogue == geglu(linear) of transformer.py:8-21."""
    o = ops()
    M, N, K = mnk
    x, xf = q16(rnd(M, K, seed=40), dtype)
    w, wf = q16(rnd(N, K, seed=41) / K ** 0.5, dtype)
    b = rnd(N, seed=42)
    wi, bi = o.interleave_geglu(w, b.to(DEV))
    y = xf @ wf.T + b
    want = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    got = o.linear(x, wi, bias=bi, geglu=True)
    assert got.shape == (M, N // 2)
    check("linear+geglu", got, want, TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [64 * 37 + 40, 64, 64 * 300])
def test_linear_ws_modes_vs_fp32(dtype, M):
    """pf_linear_ws (weight-stationary kernel of the C = 320 token layers) against fp32 torch: 16-bit out, fp32 out + fp32
    residual, GEGLU pairing -- ragged last tile (M = 64 * 37 + 40), a single tile, more tiles than workgroups."""
    o = ops()
    K = 320
    x, xf = q16(rnd(M, K, seed=60), dtype)
    for N, seed in ((320, 61), (640, 62), (2560, 63)):
        w, wf = q16(rnd(N, K, seed=seed) / K ** 0.5, dtype)
        b = rnd(N, seed=seed + 10)
        want = xf @ wf.T + b
        got = o.linear_ws(x, w, o.LWS_16, bias=b.to(DEV))
        check("linear_ws 16-bit N%d" % N, got, want, TOL[dtype])
        if N == 320:
            r = rnd(M, N, seed=70)
            got = o.linear_ws(x, w, o.LWS_F32, bias=b.to(DEV), residual=r.to(DEV))
            assert got.dtype == torch.float32
            check("linear_ws fp32 + residual", got, want + r, 2e-5)
            check("linear_ws fp32, no bias / residual", o.linear_ws(x, w, o.LWS_F32), xf @ wf.T, 2e-5)
        if N == 2560:
            wi, bi = o.interleave_geglu(w, b.to(DEV))
            got = o.linear_ws(x, wi, o.LWS_GEGLU, bias=bi)
            assert got.shape == (M, N // 2)
            check("linear_ws geglu", got, want[:, :N // 2] * F.gelu(want[:, N // 2:]), TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [32 * 37 + 20, 32, 32 * 1300])
def test_linear_ws_k640_vs_fp32(dtype, M):
    """The K = 640 shape of pf_linear_ws (32-token tiles, 256 channels per workgroup: FF1 and q | k of the 32^2 level): 16-bit out at
    N = 256 / 1280 (5 channel blocks: XCD-grouped grid) and GEGLU at N = 5120 (20 channel blocks: flat grid) -- ragged last tile, a
    single tile, more tiles than workgroups; then through ops.linear as the engine routes it."""
    o = ops()
    K = 640
    x, xf = q16(rnd(M, K, seed=160), dtype)
    for N, seed in ((256, 161), (1280, 162), (5120, 163)):
        w, wf = q16(rnd(N, K, seed=seed) / K ** 0.5, dtype)
        b = rnd(N, seed=seed + 10)
        want = xf @ wf.T + b
        if N < 5120:
            check("linear_ws K640 16-bit N%d" % N, o.linear_ws(x, w, o.LWS_16, bias=b.to(DEV)), want, TOL[dtype])
            check("linear_ws K640 no bias N%d" % N, o.linear_ws(x, w, o.LWS_16), xf @ wf.T, TOL[dtype])
        else:
            wi, bi = o.interleave_geglu(w, b.to(DEV))
            got = o.linear_ws(x, wi, o.LWS_GEGLU, bias=bi)
            assert got.shape == (M, N // 2)
            check("linear_ws K640 geglu", got, want[:, :N // 2] * F.gelu(want[:, N // 2:]), TOL[dtype])
            routed = o.linear(x, wi, bias=bi, geglu=True)
            if o.linear_ws_ok(M, N, K, o.LWS_GEGLU, x):
                assert torch.equal(routed, got)
            else:
                check("routed geglu (tile kernel)", routed, want[:, :N // 2] * F.gelu(want[:, N // 2:]), TOL[dtype])
    # N = 640 (to_q of the 32^2 level): 128-channel workgroups, 16-bit output
    w, wf = q16(rnd(640, K, seed=170) / K ** 0.5, dtype)
    b = rnd(640, seed=171)
    want = xf @ wf.T + b
    got = o.linear_ws(x, w, o.LWS_16, bias=b.to(DEV))
    check("linear_ws K640 N640 16-bit", got, want, TOL[dtype])
    if o.linear_ws_ok(M, 640, K, o.LWS_16, x):
        assert torch.equal(o.linear(x, w, bias=b.to(DEV)), got)
    assert not o.linear_ws_ok(1 << 16, 640, K, o.LWS_F32, x) and not o.linear_ws_ok(1 << 16, 320, K, o.LWS_16, x)   # fp32 + residual at N = 640, N % 128: tile kernel


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [16 * 37 + 9, 16, 10240])
def test_linear_ws_k1280_vs_fp32(dtype, M):
    """The K = 1280 shape of pf_linear_ws (16-token tiles, 16 channels per wavefront, 128 per workgroup: q | k, to_q and FF1 of the
    16^2 level): 16-bit out at N = 128 / 1280 / 2560 and GEGLU at N = 10240 (80 channel blocks: flat grid of 3 token ranges)."""
    o = ops()
    K = 1280
    x, xf = q16(rnd(M, K, seed=180), dtype)
    for N, seed in ((128, 181), (1280, 182), (2560, 183), (10240, 184)):
        w, wf = q16(rnd(N, K, seed=seed) / K ** 0.5, dtype)
        b = rnd(N, seed=seed + 10)
        want = xf @ wf.T + b
        if N < 10240:
            got = o.linear_ws(x, w, o.LWS_16, bias=b.to(DEV))
            check("linear_ws K1280 16-bit N%d" % N, got, want, TOL[dtype])
            if o.linear_ws_ok(M, N, K, o.LWS_16, x):
                assert torch.equal(o.linear(x, w, bias=b.to(DEV)), got)
        else:
            wi, bi = o.interleave_geglu(w, b.to(DEV))
            got = o.linear_ws(x, wi, o.LWS_GEGLU, bias=bi)
            assert got.shape == (M, N // 2)
            check("linear_ws K1280 geglu", got, want[:, :N // 2] * F.gelu(want[:, N // 2:]), TOL[dtype])
    assert not o.linear_ws_ok(1 << 16, 1280, K, o.LWS_F32, x)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_batch,nk,N,K", [(40, 1024, 640, 640), (40, 256, 1280, 1280), (1, 20480, 640, 640), (3, 96, 128, 640), (2, 48, 256, 1280),
                                            (1, 81920, 320, 320), (2, 192, 640, 320)])
def test_linear_ws_transposed_output(dtype, n_batch, nk, N, K):
    """PF_LWS_VT: the V projection written as V^T [batch][channel][key] (the operand layout of pf_attention) at all three shapes of the
    kernel, and through ops.linear_vt as engine._attend / the EPA block route it."""
    o = ops()
    M = n_batch * nk
    x, xf = q16(rnd(M, K, seed=190), dtype)
    w, wf = q16(rnd(N, K, seed=191) / K ** 0.5, dtype)
    want = (xf @ wf.T).reshape(n_batch, nk, N).transpose(1, 2)
    vt = o.linear_ws(x, w, o.LWS_VT, rows_per_batch=nk)
    assert vt.shape == (n_batch, N, nk)
    check("linear_ws V^T", vt, want, TOL[dtype])
    routed = o.linear_vt(x, w, n_batch)
    if o.linear_ws_ok(M, N, K, o.LWS_VT, x):
        assert routed is not None and torch.equal(routed, vt)
    else:
        assert routed is None


@pytest.mark.parametrize("M,N,K,mode", [(10240, 1280, 1280, "vt"), (40960, 640, 640, "vt"), (10240, 1280, 1280, "16"), (10240, 10240, 1280, "geglu"), (40960, 1280, 640, "16"), (40960, 5120, 640, "geglu"),
                                        (163840, 960, 320, "qkv"), (163840, 320, 320, "f32")])
def test_linear_ws_is_bit_reproducible_under_contention(M, N, K, mode):
    """pf_linear_ws reads a token tile out of LDS only after its LDS-DMA pieces have landed: 24 launches on the same operands -- into
    output buffers pre-filled with NaN / a constant, some of them beside a large GEMM on another stream -- give bit-identical results.
    (Regression: a counted s_waitcnt that included the epilogues' STORES let a tile be read early, once in ~40 launches at 16-token
    tiles -- stores and loads retire out of order with respect to each other; the wait now counts loads only.)"""
    o = ops()
    T = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(M, K, device=DEV, generator=g).to(T)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(T)
    b = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g) if mode == "f32" else None
    side, big = torch.cuda.Stream(), torch.randn(6144, 6144, device=DEV)
    ref = None
    for it in range(24):
        fill = float("nan") if it % 2 else 3.0
        if it % 3 == 0:
            with torch.cuda.stream(side):
                big @ big
        if mode == "vt":
            vt = torch.full((40, N, M // 40), fill, device=DEV, dtype=T)
            got = (o.linear_ws(x, w, o.LWS_VT, out_vt=vt, rows_per_batch=M // 40),)
        elif mode == "qkv":
            out = torch.full((M, 640), fill, device=DEV, dtype=T)
            vt = torch.full((40, 320, M // 40), fill, device=DEV, dtype=T)
            got = o.linear_ws(x, w, o.LWS_QKV, out=out, out_vt=vt, rows_per_batch=M // 40)
        elif mode == "f32":
            got = (o.linear_ws(x, w, o.LWS_F32, bias=b, residual=res, out=torch.full((M, N), fill, device=DEV)),)
        else:
            out = torch.full((M, N // 2 if mode == "geglu" else N), fill, device=DEV, dtype=T)
            got = (o.linear_ws(x, w, o.LWS_GEGLU if mode == "geglu" else o.LWS_16, bias=b, out=out),)
        torch.cuda.synchronize()
        if ref is None:
            ref = [t.clone() for t in got]
            assert all(torch.isfinite(t.float()).all() for t in ref)
        else:
            assert all(torch.equal(a, r) for a, r in zip(got, ref)), "launch %d differs from the first" % it


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [64 * 37 + 40, 64 * 600])
def test_linear_ws_with_layernorm_epilogue(dtype, M):
    """PF_LWS_F32_LN: output projection + fp32 residual, and the LayerNorm of the result from the same launch (a row of 320
    channels is spread over 8 wavefronts: per-wave (mean, M2) partials combined through LDS) -- against fp32 torch, with a
    non-trivial affine and a row offset that makes E[x^2] - mean^2 lose digits (mean 30, spread 1)."""
    o = ops()
    K = N = 320
    x, xf = q16(rnd(M, K, seed=90), dtype)
    w, wf = q16(rnd(N, K, seed=91) / K ** 0.5, dtype)
    b, r = rnd(N, seed=92), rnd(M, N, seed=93) + 30.0
    g, bt = 1.0 + 0.3 * rnd(N, seed=94), 0.2 * rnd(N, seed=95)
    want = xf @ wf.T + b + r
    out, ln = o.linear_ws(x, w, o.LWS_F32_LN, bias=b.to(DEV), residual=r.to(DEV), ln=(g.to(DEV), bt.to(DEV), 1e-5))
    assert out.dtype == torch.float32 and ln.dtype == dtype and ln.shape == (M, N)
    check("linear_ws fp32 + residual (LN mode)", out, want, 2e-5)
    check("linear_ws LayerNorm epilogue", ln, F.layer_norm(want, (N,), g, bt, 1e-5), TOL[dtype])
    out2, ln2 = o.linear_ln(x, w, b.to(DEV), r.to(DEV), g.to(DEV), bt.to(DEV), 1e-5)      # the routed entry point of engine._attend
    if o.linear_ws_ok(M, N, K, o.LWS_F32_LN, x):
        assert ln2 is not None and torch.equal(out2, out) and torch.equal(ln2, ln)
    else:                                                         # too few token tiles: tile kernel + the caller's own LayerNorm
        assert ln2 is None
        check("linear_ln fallback", out2, want, 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_batch,nk", [(3, 1024), (2, 64), (10, 4096)])
def test_linear_ws_qkv_with_transposed_values(dtype, n_batch, nk):
    """q | k | v in one launch: (q | k) rows + V^T [batch][C][keys] in the layout pf_attention reads; then through the
    routed entry points (ops.linear_qkv / ops.linear) exactly as engine._attend calls them."""
    o = ops()
    K, M = 320, n_batch * nk
    x, xf = q16(rnd(M, K, seed=80), dtype)
    w, wf = q16(rnd(960, K, seed=81) / K ** 0.5, dtype)
    want = xf @ wf.T
    qk, vt = o.linear_ws(x, w, o.LWS_QKV, rows_per_batch=nk)
    assert qk.shape == (M, 640) and vt.shape == (n_batch, 320, nk)
    check("qkv: q | k", qk, want[:, :640], TOL[dtype])
    check("qkv: V^T", vt, want[:, 640:].reshape(n_batch, nk, 320).transpose(1, 2), TOL[dtype])
    routed = o.linear_qkv(x, w, n_batch)
    if M >= o.LINEAR_WS_MIN_ROWS:
        assert routed is not None and torch.equal(routed[0], qk) and torch.equal(routed[1], vt)
        r = rnd(M, 320, seed=82).to(DEV)
        a = o.linear(x, w[:320], residual=r)                      # fp32 residual stream in -> fp32 out (either kernel, by the row count)
        check("routed linear + fp32 residual", a, want[:, :320] + r.cpu(), 2e-5)
    else:
        assert routed is None


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["linear", "conv", "conv_cat_res"])
def test_split_k(dtype, case):
    """Small output grids with a long K are split over K (fp32 slabs + ordered reduce)."""
    o = ops()
    if case == "linear":
        M, N, K = 256, 1280, 5120
        x, xf = q16(rnd(M, K, seed=43), dtype)
        w, wf = q16(rnd(N, K, seed=44) / K ** 0.5, dtype)
        b = rnd(N, seed=45)
        r, rf = q16(rnd(M, N, seed=46), dtype)
        d_ws = o.gemm_workspace_bytes(x, w, N, w_in=M)
        assert d_ws > 0, "this shape is expected to take the split-K path"
        check("split-K linear", o.linear(x, w, bias=b.to(DEV), residual=r), xf @ wf.T + b + rf, TOL[dtype])
        return
    n, h, wd, cin, cout = 2, 8, 20, 640, 320
    x, xf = q16(rnd(n, h, wd, cin, seed=47), dtype)
    skip, skipf = q16(rnd(n, h, wd, 640, seed=48), dtype)
    a1 = skip if case == "conv_cat_res" else None
    ctot = cin + (640 if a1 is not None else 0)
    wt = rnd(cout, ctot, 3, 3, seed=49) / (ctot * 9) ** 0.5
    wq, wqf = q16(wt.permute(0, 2, 3, 1).reshape(cout, -1), dtype)
    wref = wqf.reshape(cout, 3, 3, ctot).permute(0, 3, 1, 2)
    b = rnd(cout, seed=50)
    table = rnd(n, cout, seed=51)
    xin = xf.permute(0, 3, 1, 2)
    if a1 is not None:
        xin = torch.cat([xin, skipf.permute(0, 3, 1, 2)], 1)
    want = F.conv2d(xin, wref, b, padding=1) + table[:, :, None, None]
    kw = dict(n_img=n, h_in=h, w_in=wd, ksize=3, pad=1, bias=b.to(DEV), a1=a1, rowvec=table.to(DEV))
    assert o.gemm_workspace_bytes(x, wq, cout, **{k: v for k, v in kw.items() if k not in ("bias", "rowvec")}) > 0
    if case == "conv_cat_res":
        r, rf = q16(rnd(n * h * wd, cout, seed=52), dtype)
        kw["residual"] = r
        want = want + rf.view(n, h, wd, cout).permute(0, 3, 1, 2)
    got = o.conv_gemm(x, wq, cout, **kw)
    check("split-K conv " + case, got.view(n, h, wd, cout).permute(0, 3, 1, 2), want, TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
def test_tail_split_two_phase(dtype):
    """300 tiles of 256 rows on 256 CUs: 256 tiles run unsplit, the 44 left-over tile rows as a second
    launch with split K (pf_conv_gemm plans it from the shape); bias + residual epilogue on both parts."""
    o = ops()
    M, N, K = 300 * 256 - 37, 160, 2560
    x, xf = q16(rnd(M, K, seed=60), dtype)
    w, wf = q16(rnd(N, K, seed=61) / K ** 0.5, dtype)
    b = rnd(N, seed=62)
    r, rf = q16(rnd(M, N, seed=63), dtype)
    assert o.gemm_workspace_bytes(x, w, N, w_in=M) > 0
    check("tail-split linear", o.linear(x, w, bias=b.to(DEV), residual=r), xf @ wf.T + b + rf, TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [64, 128, 192, 256, 448])
def test_persistent_8wave_kernel_short_k_and_epilogue_modes(dtype, K):
    """Shapes that the plan gives to k_conv_gemm8 (>= 256 full-ish 256x160 tiles, so every block walks
    several tiles): 1..7 K steps exercise every prologue / tail combination of the three-slot ring
    (stage it+3 requested into the slot retired at the mid-step barrier), a ragged last row tile and
    each staged epilogue specialisation (bias, + residual, + per-image row vector incl. tiles that
    straddle an image boundary, GEGLU) plus the fp32 per-fragment path."""
    o = ops()
    n_img, hw = 130, 500                                           # rows_per_img = 500 >= 256: tiles cross image boundaries
    M, N = n_img * hw - 0, 320                                     # 65000 rows: 254 row tiles, the last one ragged
    x, xf = q16(rnd(M, K, seed=60 + K), dtype)
    w, wf = q16(rnd(N, K, seed=61) / K ** 0.5, dtype)
    b = rnd(N, seed=62)
    r, rf = q16(rnd(M, N, seed=63), dtype)
    base = xf @ wf.T + b

    def check_all(name, got, want, tol):
        check(name, got, want, tol)
        check(name + " (ragged last row tile)", got[-232:], want[-232:], tol)      # 65000 = 253 * 256 + 232
        check(name + " (first tile)", got[:256], want[:256], tol)

    check_all("bias", o.linear(x, w, bias=b.to(DEV)), base, TOL[dtype])
    check_all("bias+res", o.linear(x, w, bias=b.to(DEV), residual=r), base + rf, TOL[dtype])
    table = rnd(n_img, N, seed=64)
    got = o.conv_gemm(x, w, N, n_img=n_img, h_in=20, w_in=25, bias=b.to(DEV), rowvec=table.to(DEV))
    check_all("bias+rowvec", got, base + table.repeat_interleave(hw, 0), TOL[dtype])
    got = o.linear(x, w, bias=b.to(DEV), out_dtype=torch.float32)
    check_all("fp32 out", got, base, 2e-5)
    w2, w2f = q16(rnd(2 * N, K, seed=65) / K ** 0.5, dtype)
    b2 = rnd(2 * N, seed=66)
    wi, bi = o.interleave_geglu(w2, b2.to(DEV))
    y = xf @ w2f.T + b2
    check_all("geglu", o.linear(x, wi, bias=bi, geglu=True), y[:, :N] * F.gelu(y[:, N:]), TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_strided_views_and_rowvec(dtype):
    o = ops()
    M, C = 200, 128
    big, bigf = q16(rnd(M, 2 * C, seed=26), dtype)                # (q | k) buffer: use the k half as input
    w, wf = q16(rnd(64, C, seed=27) / C ** 0.5, dtype)
    check("linear on column view", o.conv_gemm(big[:, C:], w, 64, w_in=M), bigf[:, C:] @ wf.T, TOL[dtype])
    # per-image row vector from a wider table (time-embedding projection slice)
    table = rnd(4, 256, seed=28)
    x, xf = q16(rnd(4 * 50, C, seed=29), dtype)
    got = o.conv_gemm(x, w, 64, n_img=4, h_in=5, w_in=10, rowvec=table.to(DEV)[:, 128:])
    want = xf @ wf.T + table[:, 128:192].repeat_interleave(50, 0)
    check("rowvec", got, want, TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["s1", "s2", "up", "cat", "1x1cat", "pano"])
def test_conv_gemm_vs_conv2d(dtype, case):
    o = ops()
    n, h, w, cin, cout = 3, 10, 12, 128, 192
    if case == "pano":
        n, h, w = 2, 8, 20
    x, xf = q16(rnd(n, h, w, cin, seed=30), dtype)
    skip, skipf = q16(rnd(n, h, w, 64, seed=31), dtype)
    a1 = skip if "cat" in case else None
    ctot = cin + (64 if a1 is not None else 0)
    ks = 1 if case == "1x1cat" else 3
    wt = rnd(cout, ctot, ks, ks, seed=32) / (ctot * ks * ks) ** 0.5
    wq, wqf = q16(wt.permute(0, 2, 3, 1).reshape(cout, -1), dtype)
    wref = wqf.reshape(cout, ks, ks, ctot).permute(0, 3, 1, 2)
    b = rnd(cout, seed=33)
    xin = xf.permute(0, 3, 1, 2)
    if a1 is not None:
        xin = torch.cat([xin, skipf.permute(0, 3, 1, 2)], 1)
    kw = dict(n_img=n, h_in=h, w_in=w, ksize=ks, pad=ks // 2, bias=b.to(DEV), a1=a1)
    if case == "s2":
        want = F.conv2d(xin, wref, b, stride=2, padding=1)
        kw["stride"] = 2
    elif case == "up":
        want = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode="nearest"), wref, b, padding=1)
        kw["upsample"] = 1
    else:
        want = F.conv2d(xin, wref, b, padding=ks // 2)
    got = o.conv_gemm(x, wq, cout, **kw)
    check("conv " + case, got.view(n, want.shape[2], want.shape[3], cout).permute(0, 3, 1, 2), want, TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows", [77, 256, 1000])
def test_linear_transposed_batched(dtype, rows):
    B, K, N = 3, 128, 192
    x, xf = q16(rnd(B, rows, K, seed=34), dtype)
    w, wf = q16(rnd(N, K, seed=35) / K ** 0.5, dtype)
    vt = ops().linear_t(x, w)
    assert vt.shape[:2] == (B, N) and vt.shape[2] % 32 == 0 and vt.shape[2] >= rows
    check("linear_t", vt[:, :, :rows], torch.einsum("nk,brk->bnr", wf, xf), TOL[dtype])


# ------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, H, scale, bias=None):
    B, nq, Cq = q.shape
    D = Cq // H
    qh, kh, vh = (t.reshape(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    if bias is not None:
        s = s + bias
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, nq, Cq)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 3, 64, 256, 256), (2, 2, 64, 200, 77), (1, 5, 64, 1024, 1024), (2, 4, 32, 128, 320), (1, 2, 32, 96, 50),
                                 # the text cross-attention's shapes (77 keys: two key tiles, 51 masked keys) and other short key ranges
                                 (3, 5, 64, 4096, 77), (1, 10, 64, 1024, 77), (2, 20, 64, 64, 77), (1, 2, 64, 40, 96), (1, 2, 64, 33, 1),
                                 (1, 2, 64, 2080, 65), (1, 2, 64, 256, 97)])
def test_attention(dtype, cfg):
    B, H, D, nq, nk = cfg
    Cq = H * D
    q, qf = q16(rnd(B, nq, Cq, seed=36), dtype)
    k, kf = q16(rnd(B, nk, Cq, seed=37), dtype)
    v, vf = q16(rnd(B, nk, Cq, seed=38), dtype)
    ld = ((nk + 31) // 32) * 32
    vt = torch.full((B, Cq, ld), float("nan"), dtype=dtype, device=DEV)     # padding must never be read as data
    vt[:, :, :nk] = v.transpose(1, 2)
    out = ops().attention(q, k, vt, B, H, D, nq, nk, q_ld=Cq, k_ld=Cq, vt_ld=ld, q_bs=nq * Cq, k_bs=nk * Cq, vt_bs=Cq * ld)
    check("attention", out, attn_ref(qf, kf, vf, H, D ** -0.5), 2.5 * TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_fused_qk_buffer_and_bias(dtype):
    """EPA shape: q and k live in one (q|k) buffer; sparse-flagged additive bias shared over batch/heads."""
    B, H, D, nq, nk = 2, 2, 32, 128, 256
    Cq = H * D
    qk_q, qkf_q = q16(rnd(B * nq, 2 * Cq, seed=39), dtype)
    qk_k, qkf_k = q16(rnd(B * nk, 2 * Cq, seed=40), dtype)
    v, vf = q16(rnd(B, nk, Cq, seed=41), dtype)
    vt = v.transpose(1, 2).contiguous()
    g = torch.Generator().manual_seed(42)
    bias = torch.zeros(nq, nk)
    for (i, j) in [(0, 0), (1, 3), (3, 7), (2, 2)]:                 # a few non-zero 32x32 tiles
        bias[32 * i:32 * i + 32, 32 * j:32 * j + 32] = torch.rand(32, 32, generator=g) * 2 * (torch.rand(32, 32, generator=g) > 0.7)
    flags = (bias.reshape(nq // 32, 32, nk // 32, 32).abs().amax((1, 3)) > 0).to(torch.uint8)
    out = ops().attention(qk_q, qk_k[:, Cq:], vt, B, H, D, nq, nk, q_ld=2 * Cq, k_ld=2 * Cq, vt_ld=nk,
                          q_bs=nq * 2 * Cq, k_bs=nk * 2 * Cq, vt_bs=Cq * nk, bias=bias.to(DEV), flags=flags.to(DEV))
    want = attn_ref(qkf_q[:, :Cq].reshape(B, nq, Cq), qkf_k[:, Cq:].reshape(B, nk, Cq), vf, H, D ** -0.5, bias)
    check("attention+bias", out, want, 2.5 * TOL[dtype])
    # shift invariance used by the tables: bias and bias - 1 give the same result (models/pano/utils.py:72,76)
    want2 = attn_ref(qkf_q[:, :Cq].reshape(B, nq, Cq), qkf_k[:, Cq:].reshape(B, nk, Cq), vf, H, D ** -0.5, bias - 1)
    check("shift invariance", want, want2, 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 4, 256, 4096), (1, 10, 128, 1280), (2, 3, 200, 2084), (1, 2, 96, 1156)])
def test_attention_split_key_range(dtype, cfg, monkeypatch):
    """pf_attn_desc.workspace (round 6): a biased D = 32 launch with few query blocks and many keys (the panorama-query direction of an EPA block) splits its key
    range over several workgroups -- normalised 16-bit partial outputs + log-sum-exps, combined by a second launch.  Against the fp32 reference, against the unsplit
    launch (PF_ATTENTION_SPLIT=0), with ragged query / key counts (a partial last key tile in the last split, bias tiles flagged off)."""
    import ctypes as C
    from panfusion_amd import _lib
    B, H, nq, nk = cfg
    D, Cq = 32, H * 32
    q, qf = q16(rnd(B, nq, Cq, seed=61), dtype)
    k, kf = q16(rnd(B, nk, Cq, seed=62), dtype)
    v, vf = q16(rnd(B, nk, Cq, seed=63), dtype)
    ld = ((nk + 31) // 32) * 32
    vt = torch.zeros(B, Cq, ld, dtype=dtype, device=DEV)
    vt[:, :, :nk] = v.transpose(1, 2)
    g = torch.Generator().manual_seed(64)
    nqt, nkt32 = (nq + 31) // 32, (nk + 31) // 32
    bias_full = torch.rand(nqt * 32, nkt32 * 32, generator=g) * 2
    on = torch.rand(nqt, nkt32, generator=g) < 0.05                       # 5 % of the 32 x 32 tiles carry a bias (the tables: 1-2 %)
    bias_full = bias_full * on.repeat_interleave(32, 0).repeat_interleave(32, 1)
    bias = bias_full[:nq, :nk].contiguous()
    flags = on.to(torch.uint8)
    kw = dict(q_ld=Cq, k_ld=Cq, vt_ld=ld, q_bs=nq * Cq, k_bs=nk * Cq, vt_bs=Cq * ld, bias=bias.to(DEV), flags=flags.to(DEV))
    d = _lib.AttnDesc()
    d.B, d.H, d.D, d.nq, d.nk, d.o_ld, d.o_bs, d.bias = B, H, D, nq, nk, Cq, nq * Cq, 1
    assert _lib.lib().pf_attention_workspace_size(C.byref(d)) > 0, "this problem should split"
    out = ops().attention(q, k, vt, B, H, D, nq, nk, **kw)
    monkeypatch.setenv("PF_ATTENTION_SPLIT", "0")
    assert _lib.lib().pf_attention_workspace_size(C.byref(d)) == 0
    whole = ops().attention(q, k, vt, B, H, D, nq, nk, **kw)
    want = attn_ref(qf, kf, vf, H, D ** -0.5, bias)
    check("split attention vs fp32", out, want, 2.5 * TOL[dtype])
    check("unsplit attention vs fp32", whole, want, 2.5 * TOL[dtype])
    check("split vs unsplit", out, whole, 1.5 * TOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pp,cfg", [(1, (2, 5, 64, 512, 512)), (1, (1, 3, 64, 300, 1000)), (1, (1, 2, 64, 64, 136)), (1, (3, 2, 64, 257, 4096)),
                                    (2, (2, 5, 64, 512, 512)), (2, (1, 3, 64, 300, 1152)), (2, (1, 2, 64, 33, 256)), (2, (3, 2, 64, 257, 4096))])
def test_pingpong_attention_kernels(dtype, pp, cfg):
    """The two experimental 8-wave ping-pong kernels (PF_ATTENTION_PP = 1: one key tile per phase, ragged key counts that are
    multiples of 8; = 2: two tiles per phase, multiples of 128) against the fp32 reference and against the shipped kernel --
    DESIGN.md section 3.4: they are kept for what they measure, and have to stay correct to measure anything."""
    import os
    B, H, D, nq, nk = cfg
    Cq = H * D
    q, qf = q16(rnd(B, nq, Cq, seed=46), dtype)
    k, kf = q16(rnd(B, nk, Cq, seed=47), dtype)
    v, vf = q16(rnd(B, nk, Cq, seed=48), dtype)
    ld = ((nk + 31) // 32) * 32
    vt = torch.full((B, Cq, ld), float("nan"), dtype=dtype, device=DEV)     # padding must never be read as data
    vt[:, :, :nk] = v.transpose(1, 2)
    kw = dict(q_ld=Cq, k_ld=Cq, vt_ld=ld, q_bs=nq * Cq, k_bs=nk * Cq, vt_bs=Cq * ld)
    base = ops().attention(q, k, vt, B, H, D, nq, nk, **kw)
    os.environ["PF_ATTENTION_PP"] = str(pp)
    try:
        outs = [ops().attention(q, k, vt, B, H, D, nq, nk, **kw) for _ in range(3)]
    finally:
        del os.environ["PF_ATTENTION_PP"]
    want = attn_ref(qf, kf, vf, H, D ** -0.5)
    check("ping-pong attention", outs[0], want, 2.5 * TOL[dtype])
    check("ping-pong vs shipped kernel", outs[0], base.float(), 2.5 * TOL[dtype])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])          # (no scheduling dependence)


def test_gemm_and_attention_are_bit_reproducible_under_contention():
    """The tile GEMM kernels (persistent 8-wave kernel with its counted LDS-DMA waits: plain 3x3 conv with fp32 output + residual +
    GroupNorm moments, the split-precision K step, a short-K GEGLU linear; the 4-wave kernel with split K) and both attention
    instantiations: 12 launches each on the same operands, every third beside a large GEMM on another stream, into buffers pre-filled
    with NaN / a constant -- bit-identical results (a tile read before its DMA has landed shows up as a rare wrong tile, not as a
    parity failure: the weight-stationary kernel had exactly that, tests above)."""
    from panfusion_amd import engine
    o = ops()
    T = torch.float16
    g = torch.Generator(device=DEV).manual_seed(9)
    rn = lambda *sh: torch.randn(*sh, device=DEV, generator=g)
    n, h, w, c = 20, 64, 64, 320
    x = rn(n, h, w, c).to(T)
    w3 = (rn(c, 9 * c) / (9 * c) ** 0.5).to(T)
    res = rn(n * h * w, c)
    xs = rn(n * h * w, c)
    w1 = rn(c, c) / c ** 0.5
    pair = engine.split_operand(xs.view(n, h * w, c), dtype=T)
    ws3 = engine._split_weight(w1, 1, torch.device(DEV), T)
    wg, bg = o.interleave_geglu((rn(5120, 640) / 25.0).to(T), rn(5120))
    xg = rn(4096, 640).to(T)
    xk, wk = rn(256, 5120).to(T), (rn(1280, 5120) / 70.0).to(T)
    B, H, nq = 4, 5, 1024
    q, k = rn(B * nq, H * 64).to(T), rn(B * nq, H * 64).to(T)
    vt = rn(B, H * 64, nq).to(T)
    q32, k32, vt32 = rn(2 * 512, 160).to(T), rn(2 * 2048, 160).to(T), rn(2, 160, 2048).to(T)
    bias = torch.zeros(512, 2048, device=DEV)
    bias[::7, ::5] = 1.5
    flags = (bias.view(16, 32, 64, 32).abs().amax((1, 3)) > 0).to(torch.uint8).contiguous()
    cases = {
        "conv3x3 fp32 + residual + moments": lambda: o.conv_gemm(x, w3, c, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, residual=res, gn_stats=True, out_dtype=torch.float32),
        "conv3x3 16-bit + moments": lambda: o.conv_gemm(x, w3, c, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, gn_stats=True),
        "split-precision 1x1": lambda: engine.exact_gemm(pair, ws3, c, w_in=n * h * w, residual=res, out_dtype=torch.float32),
        "GEGLU linear": lambda: o.conv_gemm(xg, wg, 5120, w_in=4096, bias=bg, geglu=True),
        "split-K linear": lambda: o.conv_gemm(xk, wk, 1280, w_in=256),
        "attention D64": lambda: o.attention(q, k, vt, B, H, 64, nq, nq, q_ld=H * 64, k_ld=H * 64, vt_ld=nq, q_bs=nq * H * 64, k_bs=nq * H * 64, vt_bs=H * 64 * nq),
        "attention D32 + bias": lambda: o.attention(q32, k32, vt32, 2, 5, 32, 512, 2048, q_ld=160, k_ld=160, vt_ld=2048, q_bs=512 * 160, k_bs=2048 * 160,
                                                    vt_bs=160 * 2048, bias=bias, flags=flags),
    }
    side, big = torch.cuda.Stream(), rn(6144, 6144)
    for name, run in cases.items():
        ref = None
        for it in range(12):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    big @ big
            out = run()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
                assert torch.isfinite(ref.float()).all(), name
            else:
                assert torch.equal(out, ref), "%s: launch %d differs from the first" % (name, it)


def test_attention_online_softmax_rescale_branch():
    """A key row that dominates late forces the running-max rescale (guide rule 26)."""
    dtype = torch.float16
    B, H, D, nq, nk = 1, 1, 64, 32, 256
    q, qf = q16(rnd(B, nq, D, seed=43), dtype)
    kk = rnd(B, nk, D, seed=44)
    kk[0, 200] = qf[0, 5] * 3.0                                    # spike for query 5 at key tile 6
    k, kf = q16(kk, dtype)
    v, vf = q16(rnd(B, nk, D, seed=45), dtype)
    vt = v.transpose(1, 2).contiguous()
    out = ops().attention(q, k, vt, B, H, D, nq, nk, q_ld=D, k_ld=D, vt_ld=nk, q_bs=nq * D, k_bs=nk * D, vt_bs=D * nk, scale=1.0)
    check("attention rescale", out, attn_ref(qf, kf, vf, H, 1.0), 2.5 * TOL[dtype])


# ------------------------------------------------------------------------------------ dataset-side cropping
def test_py360_e2p_bit_exact_vs_reference_fixture():
    """external/py360convert e2p (dataset/PanoDataset.py:133-140) on the GPU, against the
    fixture the reference's own module wrote (scipy map_coordinates): identical bytes for uint8 RGB and float32,
    bilinear and nearest, poles / seam / asymmetric field of view; single-crop and batched entry points."""
    from panfusion_amd.external.py360convert import e2p, e2p_views
    g = golden("py360_e2p.npz")
    us, vs = g["cams"][:, 0], g["cams"][:, 1]
    from oracle import py360
    # nearest mode rounds the sampling position: a position within 1e-9 of a half-integer is a tie that the last
    # ulp of the BLAS matrix product decides in the reference (the pole-facing camera (90, 90) of the fixture has
    # one: x == z up to 1 ulp -> coor_x = 7.4999999999999964); such pixels are excluded, everything else is exact
    def no_tie(u, v):
        cx, cy = py360.coordinates(32, 64, (90, 90), u, v, (24, 24))
        cx, cy = py360._wrap_coord(cx, 64), py360._wrap_coord(cy, 34)
        frac = lambda c: np.abs((c + 0.5) - np.round(c + 0.5))
        return (frac(cx) > 1e-9) & (frac(cy) > 1e-9)
    keep = np.stack([no_tie(u, v) for u, v in zip(us, vs)])
    assert keep.mean() > 0.9                      # (axis-aligned cameras put whole columns exactly on half-integers)
    for mode in ("bilinear", "nearest"):
        for key in ("rgb", "gray"):
            got = e2p_views(g[key], (90, 90), us, vs, (24, 24), mode=mode)
            want = g[key + "_" + mode]
            assert got.dtype == want.dtype and got.shape == want.shape
            ok = (got == want) if mode == "bilinear" else ((got == want) | ~(keep[..., None] if got.ndim == 4 else keep))
            assert ok.all(), "%s %s: %d of %d values differ" % (key, mode, (~ok).sum(), want.size)
            assert (got != want).sum() <= 4 * (want.shape[-1] if want.ndim == 4 else 1)    # ... and ties are a handful of pixels
    assert np.array_equal(e2p_views(g["rgb"], (60, 45), us, vs, (18, 24)), g["rgb_fov60x45"])
    assert np.array_equal(e2p(g["rgb"], (90, 90), us[1], vs[1], (24, 24)), g["rgb_bilinear"][1])
    # tensor in, tensor out (no host round trip)
    t = e2p_views(torch.from_numpy(g["gray"]).to(DEV), (90, 90), us, vs, (24, 24))
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), g["gray_bilinear"])


def test_py360_e2p_dataset_size_vs_oracle():
    """The dataset's real shape: 20 icosahedron crops of 512^2 from a 1024x2048 RGB panorama, against the numpy
    oracle (oracle/py360.py, pinned to the reference) on a few of the cameras."""
    from oracle import py360
    from panfusion_amd.external.py360convert import e2p_views
    rng = np.random.default_rng(5)
    pano = (rng.random((1024, 2048, 3)) * 255).astype(np.uint8)
    thd, phd = ico()
    got = e2p_views(pano, (90, 90), thd, phd, (512, 512))
    assert got.shape == (20, 512, 512, 3) and got.dtype == np.uint8
    for i in (0, 7, 19):
        want = py360.e2p(pano, (90, 90), thd[i], phd[i], (512, 512))
        assert np.array_equal(got[i], want), "camera %d: %d values differ" % (i, (got[i] != want).sum())


def test_py360_e2p_random_cases_vs_oracle():
    """Random cameras / fields of view / in-plane rotations / odd image sizes, uint8 and float32, both modes, against the
    numpy oracle (pinned to the reference module); nearest-mode ties (positions within 1e-9 of a half-integer) excluded."""
    from oracle import py360
    from panfusion_amd.external.py360convert import e2p
    rng = np.random.default_rng(21)
    for trial in range(10):
        H = int(rng.integers(9, 60))
        W = 2 * H + int(rng.integers(0, 3))
        img = (rng.random((H, W, 3)) * 255).astype(np.uint8) if trial % 2 == 0 else rng.standard_normal((H, W, 2)).astype(np.float32)
        u, v = float(rng.uniform(-180, 180)), float(rng.uniform(-89, 89))
        fov = (float(rng.uniform(40, 120)), float(rng.uniform(40, 120)))
        hw = (int(rng.integers(5, 40)), int(rng.integers(5, 40)))
        rot = float(rng.uniform(-30, 30)) if trial % 3 == 0 else 0.0
        for mode in ("bilinear", "nearest"):
            want = py360.e2p(img, fov, u, v, hw, in_rot_deg=rot, mode=mode)
            got = e2p(img, fov, u, v, hw, in_rot_deg=rot, mode=mode)
            assert got.dtype == want.dtype and got.shape == want.shape
            if mode == "nearest":
                cx, cy = py360.coordinates(H, W, fov, u, v, hw, rot)
                cx, cy = py360._wrap_coord(cx, W), py360._wrap_coord(cy, H + 2)
                frac = lambda c: np.abs((c + 0.5) - np.round(c + 0.5))
                keep = ((frac(cx) > 1e-9) & (frac(cy) > 1e-9))[..., None]
                assert ((got == want) | ~keep).all(), (trial, mode)
            elif img.dtype == np.uint8:
                assert np.array_equal(got, want), (trial, mode, int((got != want).sum()))
            else:       # float32 bilinear: weights differ by the device's atan2 / sqrt ulps -> a few float32 ulp of the result
                assert np.allclose(got, want, rtol=0, atol=2e-6), (trial, mode, float(np.abs(got - want).max()))


# ------------------------------------------------------------------------------------ round 3: GroupNorm moments from the GEMM epilogue
def _moments_ref(y, rows):
    """[M, N] fp32 -> [M / rows, 2, N / 2]: per column PAIR (sum, sum of squares) over runs of `rows` rows."""
    yr = y.double().reshape(-1, rows, y.shape[1] // 2, 2)
    return torch.stack([yr.sum((1, 3)), (yr * yr).sum((1, 3))], 1).float()


GN_CASES = [
    # (n_img, h, w, cin, cout, ksize, what): the operand mixes of the layers that feed a GroupNorm.  (Sizes chosen so that
    # the plan is NOT a split-K one: those write the output from the reduce kernel and report pf_conv_gemm_gn_rows = 0.)
    (40, 32, 32, 128, 320, 3, "rowvec16"),     # resnet conv1: bias + time-embedding row, 16-bit out (8-wave kernel, MODE 1)
    (40, 32, 32, 128, 320, 3, "res32"),        # resnet conv2: bias + fp32 residual, fp32 out (8-wave kernel, fp32 epilogue)
    (40, 32, 32, 128, 320, 3, "res16"),        # all-16-bit scheme: bias + 16-bit residual (MODE 2)
    (40, 32, 32, 128, 256, 3, "plain16"),      # VAE resnet conv1 (MODE 0), 128-wide N tiles
    (40, 32, 32, 128, 320, 3, "plain32"),      # up-sampling conv: fp32 out, no residual
    (40, 32, 32, 64, 128, 3, "res32"),         # VAE widths, fp32 stream
    (2, 16, 24, 64, 128, 3, "rowvec16"),       # small problem: 4-wave kernel
    (2, 16, 24, 64, 128, 3, "res32"),
    (1, 1, 40960, 320, 320, 1, "res32"),       # proj_out / EPA FF2 as linear layers (image structure is the consumer's)
    (1, 1, 163840, 320, 320, 1, "res32"),
    (3, 8, 20, 64, 128, 3, "rowvec16"),        # 160 rows per image: the 32-row runs of the 64-row-tile kernel
    (2, 64, 132, 320, 320, 3, "res32"),        # the padded panorama at 64 x (128 + 4)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", GN_CASES)
def test_conv_gemm_leaves_groupnorm_moments(dtype, case, monkeypatch):
    """pf_conv_desc.gn_partial: the epilogue's per-column-pair moments == moments of the tensor it wrote, the output is
    unchanged by asking for them, and GroupNorm scale / shift from them == the statistics pass over the tensor.
    (Layers with a residual are served only under PF_GN_EPILOGUE_RES=1 -- off by default, it does not pay -- set here so
    that the residual variants of the moment phase stay covered.)"""
    monkeypatch.setenv("PF_GN_EPILOGUE_RES", "1")
    o = ops()
    monkeypatch.setattr(o._PLANS, "plans", {})          # (ops caches the library's per-shape plan; this switch changes it)
    n, h, w, cin, cout, ks, what = case
    x, xf = q16(rnd(n, h, w, cin, seed=60), dtype)
    wt, wf = q16(rnd(cout, ks * ks * cin, seed=61) / (ks * ks * cin) ** 0.5, dtype)
    b = rnd(cout, seed=62).to(DEV)
    M = n * h * w
    kw = dict(n_img=n, h_in=h, w_in=w, ksize=ks, pad=ks // 2, bias=b)
    if what == "rowvec16":
        kw["rowvec"] = rnd(n, cout + 64, seed=63).to(DEV)[:, 32:32 + cout]        # strided view, like temb_all[:, off:]
    elif what == "res32":
        kw["residual"] = (rnd(M, cout, seed=64) * 2 + 0.3).to(DEV)
    elif what == "res16":
        kw["residual"] = q16(rnd(M, cout, seed=64), dtype)[0]
    elif what == "plain32":
        kw["out_dtype"] = torch.float32
    base = o.conv_gemm(x, wt, cout, **kw)
    got = o.conv_gemm(x, wt, cout, gn_stats=True, **kw)
    assert torch.equal(base, got), "asking for the moments changed the output"
    st = getattr(got, "_pf_gn", None)
    assert st is not None, "no moments for %s" % (case,)
    part, rows = st
    assert (h * w) % rows == 0 and part.shape == (M // rows, 2, cout // 2)
    # the moments are taken on the fp32 values BEFORE the 16-bit rounding of a 16-bit output
    ref = _moments_ref(got.float().cpu(), rows)
    tol = 3e-3 if (dtype == torch.bfloat16 and got.dtype != torch.float32) else (6e-4 if got.dtype != torch.float32 else 2e-5)
    # sums cancel (zero-mean columns): compare them on the scale of sqrt(count * sum of squares), the Cauchy-Schwarz bound
    norm = (ref[:, 1] * 2 * rows).sqrt().clamp_min(1e-6)
    d = float(((part[:, 0].float().cpu() - ref[:, 0]) / norm).abs().max())
    assert d <= tol, "moment sums off by %.3e of their scale (tolerance %.1e)" % (d, tol)
    check("moment squares", part[:, 1], ref[:, 1], tol * 4)
    # consumer: GroupNorm over (this tensor) and over (this tensor | a second source without moments -> falls back)
    groups = 32 if cout % 64 == 0 else 16
    gam, bet = (rnd(cout, seed=65) * 0.2 + 1).to(DEV), (rnd(cout, seed=66) * 0.1).to(DEV)
    hw = h * w if ks == 3 else 4096
    nimg = M // hw
    t3 = o.carry(got.view(nimg, hw, cout), got)
    sc, sh = o.groupnorm_scale_shift(t3, None, nimg, hw, groups, 1e-5, gam, bet)
    plain = got.view(nimg, hw, cout).clone()                                     # (a clone carries no moments: statistics pass)
    sc0, sh0 = o.groupnorm_scale_shift(plain, None, nimg, hw, groups, 1e-5, gam, bet)
    check("scale from moments", sc, sc0, 2e-3 if got.dtype != torch.float32 else 2e-5)
    want = F.group_norm(got.float().cpu().view(nimg, hw, cout).permute(0, 2, 1), groups, gam.cpu(), bet.cpu(), 1e-5).permute(0, 2, 1)
    y = o.scale_shift_act(t3, None, nimg, hw, sc, sh, 0, out_dtype=torch.float32)
    check("groupnorm from moments", y, want, 1e-3 if got.dtype != torch.float32 else 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_groupnorm_moments_of_a_channel_concat(dtype):
    """norm1 of a decoder resnet: GroupNorm over (x | skip), both carrying moments from DIFFERENT kernels / run lengths."""
    o = ops()
    n, h, w = 24, 32, 32
    gen = lambda c, s: q16(rnd(n, h, w, c, seed=s), dtype)[0]
    wt = lambda co, ci, s: q16(rnd(co, 9 * ci, seed=s) / (9 * ci) ** 0.5, dtype)[0]
    a = o.conv_gemm(gen(64, 70), wt(320, 64, 71), 320, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, out_dtype=torch.float32, gn_stats=True)
    b = o.conv_gemm(gen(128, 72), wt(128, 128, 73), 128, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, out_dtype=torch.float32, gn_stats=True)
    assert getattr(a, "_pf_gn", None) is not None and getattr(b, "_pf_gn", None) is not None, (getattr(a, "_pf_gn", None), getattr(b, "_pf_gn", None))
    C = 448
    gam, bet = (rnd(C, seed=74) * 0.2 + 1).to(DEV), (rnd(C, seed=75) * 0.1).to(DEV)
    a3, b3 = o.carry(a.view(n, h * w, 320), a), o.carry(b.view(n, h * w, 128), b)
    sc, sh = o.groupnorm_scale_shift(a3, b3, n, h * w, 32, 1e-5, gam, bet)
    y = o.scale_shift_act(a3, b3, n, h * w, sc, sh, 1, out_dtype=torch.float32)
    cat = torch.cat([a.view(n, h * w, 320), b.view(n, h * w, 128)], -1).float().cpu()
    want = F.silu(F.group_norm(cat.permute(0, 2, 1), 32, gam.cpu(), bet.cpu(), 1e-5)).permute(0, 2, 1)
    check("concat groupnorm from moments", y, want, 2e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_scale_shift_act_with_raw_pair(dtype):
    """norm1 + SiLU of an fp32 stream tensor and, from the same pass, the un-normalised [hi | lo] pair (shortcut operand)."""
    o = ops()
    n, hw, c0, c1 = 3, 200, 192, 64
    x0, x1 = (rnd(n, hw, c0, seed=80) * 3 + 0.2).to(DEV), (rnd(n, hw, c1, seed=81) * 3).to(DEV)
    gam, bet = (rnd(c0 + c1, seed=82) * 0.2 + 1).to(DEV), (rnd(c0 + c1, seed=83) * 0.1).to(DEV)
    sc, sh = o.groupnorm_scale_shift(x0, x1, n, hw, 32, 1e-5, gam, bet)
    y_ref = o.scale_shift_act(x0, x1, n, hw, sc, sh, 1, out_dtype=dtype)
    p_ref = o.scale_shift_act(x0, x1, n, hw, None, None, 0, out_dtype=dtype, split=True)
    y, pair = o.scale_shift_act(x0, x1, n, hw, sc, sh, 1, out_dtype=dtype, raw_pair=True)
    assert torch.equal(y, y_ref) and torch.equal(pair, p_ref)
    hi, lo = unpair(pair)
    rec = hi.float() + lo.float()
    check("pair reconstructs the input", rec, torch.cat([x0, x1], -1), 2e-6 if dtype == torch.float16 else 4e-5)


# ------------------------------------------------------------------------------------ round 3: pad_pano / unpad_pano folded into the conv
WRAP_CASES = [
    # (n, h, w, cin, cout, what): the three padded module kinds of the panorama branch at the benchmark sizes and small ones
    (2, 64, 128, 320, 320, "resnet"),      # 8-wave kernel, M = 2 x 64 x 132
    (2, 16, 32, 128, 128, "resnet"),       # 4-wave kernel
    (2, 8, 16, 256, 256, "resnet"),        # split-K plan
    (2, 64, 128, 320, 320, "down"),
    (2, 16, 32, 128, 128, "down"),
    (2, 32, 64, 128, 128, "up"),
    (2, 8, 16, 256, 256, "up"),
    (3, 6, 10, 64, 64, "resnet"),          # ragged tiles
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", WRAP_CASES)
def test_conv_gemm_virtual_circular_padding(dtype, case):
    """pf_conv_desc.wrap_pad / crop == the panorama branch's pad_pano -> conv -> unpad_pano with materialised copies
    (utils/pano.py:74-105; MVGenModel.py:110-115 resnet convs, :138-144 down-sampling, :272-277 up-sampling): BIT-identical
    outputs (same products in the same order, only the addressing differs)."""
    o = ops()
    n, h, w, cin, cout, what = case
    x, _ = q16(rnd(n, h, w, cin, seed=90), dtype)
    wt, _ = q16(rnd(cout, 9 * cin, seed=91) / (9 * cin) ** 0.5, dtype)
    b = rnd(cout, seed=92).to(DEV)
    if what == "resnet":
        # conv1: virtual pad 2, all w + 4 columns;  conv2: reads the padded intermediate, writes the w central columns
        ref1 = o.conv_gemm(o.pad_width(x, 2), wt, cout, n_img=n, h_in=h, w_in=w + 4, ksize=3, pad=1, bias=b)
        got1 = o.conv_gemm(x, wt, cout, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, bias=b, wrap_pad=2)
        assert got1.shape == ref1.shape and torch.equal(got1, ref1)
        mid = ref1.view(n, h, w + 4, cout)
        wt2, _ = q16(rnd(cout, 9 * cout, seed=93) / (9 * cout) ** 0.5, dtype)
        res = (rnd(n * h * w, cout, seed=94)).to(DEV)
        ref2 = o.crop_width(o.conv_gemm(mid, wt2, cout, n_img=n, h_in=h, w_in=w + 4, ksize=3, pad=1, bias=b,
                                         residual=o.pad_width(res.view(n, h, w, cout), 2).view(-1, cout)).view(n, h, w + 4, cout), 2)
        got2 = o.conv_gemm(mid, wt2, cout, n_img=n, h_in=h, w_in=w + 4, ksize=3, pad=1, bias=b, residual=res, crop=2)
        # (the cropped problem has fewer rows: it may take another tile / split-K plan than the padded one -- same products,
        # another fp32 summation order)
        check("conv2 with the crop folded in", got2.view(n, h, w, cout), ref2, 1e-6)
    elif what == "down":
        ref = o.crop_width(o.conv_gemm(o.pad_width(x, 2), wt, cout, n_img=n, h_in=h, w_in=w + 4, ksize=3, stride=2, pad=1, bias=b)
                           .view(n, h // 2, w // 2 + 2, cout), 1)
        got = o.conv_gemm(x, wt, cout, n_img=n, h_in=h, w_in=w, ksize=3, stride=2, pad=1, bias=b, wrap_pad=2, crop=1)
        check("down-sampling conv", got.view(n, h // 2, w // 2, cout), ref, 1e-6 if got.dtype == torch.float32 else TOL[dtype])
    else:
        ref = o.crop_width(o.conv_gemm(o.pad_width(x, 1), wt, cout, n_img=n, h_in=h, w_in=w + 2, ksize=3, pad=1, upsample=1, bias=b)
                           .view(n, 2 * h, 2 * w + 4, cout), 2)
        got = o.conv_gemm(x, wt, cout, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, upsample=1, bias=b, wrap_pad=1, crop=2)
        check("up-sampling conv", got.view(n, 2 * h, 2 * w, cout), ref, TOL[dtype])


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_groupnorm_statistics_of_the_virtually_padded_tensor(dtype):
    o = ops()
    n, h, w, c0, c1 = 2, 16, 32, 128, 64
    x0 = (rnd(n, h, w, c0, seed=95) * 2 + 0.4).to(DEV).to(dtype)
    x1 = (rnd(n, h, w, c1, seed=96) * 2).to(DEV).to(dtype)
    gam, bet = (rnd(c0 + c1, seed=97) * 0.2 + 1).to(DEV), (rnd(c0 + c1, seed=98) * 0.1).to(DEV)
    p0, p1 = o.pad_width(x0, 2), o.pad_width(x1, 2)
    sc_ref, sh_ref = o.groupnorm_scale_shift(p0.view(n, -1, c0), p1.view(n, -1, c1), n, h * (w + 4), 32, 1e-5, gam, bet)
    sc, sh = o.groupnorm_scale_shift(x0.view(n, -1, c0), x1.view(n, -1, c1), n, h * w, 32, 1e-5, gam, bet, wrap=(w, 2))
    check("scale", sc, sc_ref, 1e-5)
    check("shift", sh, sh_ref, 2e-5)


# ------------------------------------------------------------------------------------ round 3: split-K combined inside the launch
SPLITK_INKERNEL_CASES = [
    # (n, h, w, cin, cout, ksize, extras)
    (40, 8, 8, 1280, 1280, 3, "res32"),        # 8-wave kernel, split K (the 8 x 8 level of the view branch)
    (40, 16, 16, 1280, 1280, 3, "rowvec16"),   # tail split: full rounds unsplit + split tail rows
    (2, 8, 20, 1280, 1280, 3, "res32"),        # the panorama's innermost level: 4-wave kernel, deep split
    (2, 16, 36, 640, 1280, 3, "plain16"),
    (1, 1, 2048, 1920, 640, 1, "res32"),       # long-K linear on few rows (128 tiles x 2 K slices: one round of the four-slot ring)
    (3, 6, 10, 256, 64, 3, "res16"),           # ragged tiles
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", SPLITK_INKERNEL_CASES)
def test_split_k_combined_in_kernel_equals_the_reduce_kernel(dtype, case, monkeypatch):
    """pf_conv_desc.tickets: the last-arriving K-slice workgroup combines the slabs in split order -- BIT-identical to the
    second-kernel combine, run after run (arrival order must not matter), and the counters are zero again afterwards."""
    o = ops()
    n, h, w, cin, cout, ks, what = case
    x, _ = q16(rnd(n, h, w, cin, seed=110), dtype)
    wt, _ = q16(rnd(cout, ks * ks * cin, seed=111) / (ks * ks * cin) ** 0.5, dtype)
    b = rnd(cout, seed=112).to(DEV)
    M = n * h * w
    kw = dict(n_img=n, h_in=h, w_in=w, ksize=ks, pad=ks // 2, bias=b)
    if what == "rowvec16":
        kw["rowvec"] = rnd(n, cout, seed=113).to(DEV)
    elif what == "res32":
        kw["residual"] = rnd(M, cout, seed=114).to(DEV)
    elif what == "res16":
        kw["residual"] = q16(rnd(M, cout, seed=114), dtype)[0]
    assert o.gemm_workspace_bytes(x, wt, cout, n_img=n, h_in=h, w_in=w, ksize=ks, pad=ks // 2) > 0, "not a split-K plan: %s" % (case,)
    monkeypatch.setattr(o, "SPLITK_INKERNEL", False)             # (ops reads PF_SPLITK_INKERNEL once, at import)
    ref = o.conv_gemm(x, wt, cout, **kw)
    monkeypatch.setattr(o, "SPLITK_INKERNEL", True)
    for _ in range(6):
        got = o.conv_gemm(x, wt, cout, **kw)
        assert torch.equal(got, ref)
    torch.cuda.synchronize()
    ring = o._TICKETS[("cuda", torch.cuda.current_device())][0]
    assert int(ring.abs().sum()) == 0, "arrival counters not reset"


def test_split_k_in_kernel_under_concurrent_streams():
    """Uneven load: split-K launches of two streams interleave on the chip (the panorama branch runs beside the view
    branch); every result must equal the serial one."""
    o = ops()
    saved, o.SPLITK_INKERNEL = o.SPLITK_INKERNEL, True            # (off by default: it does not pay; kept correct and tested)
    dtype = torch.float16
    mk = lambda n, hh, ww, ci, co, seed: (q16(rnd(n, hh, ww, ci, seed=seed), dtype)[0],
                                           q16(rnd(co, 9 * ci, seed=seed + 1) / (9 * ci) ** 0.5, dtype)[0], n, hh, ww, co)
    probs = [mk(40, 8, 8, 1280, 1280, 120), mk(2, 8, 20, 1280, 1280, 122), mk(2, 16, 36, 640, 1280, 124), mk(40, 8, 8, 2560, 1280, 126)]
    run = lambda pr: o.conv_gemm(pr[0], pr[1], pr[5], n_img=pr[2], h_in=pr[3], w_in=pr[4], ksize=3, pad=1)
    want = [run(pr) for pr in probs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(8):
        with torch.cuda.stream(s1):
            a = [run(probs[0]), run(probs[3])]
        with torch.cuda.stream(s2):
            bb = [run(probs[1]), run(probs[2])]
        outs.append((a, bb))
    torch.cuda.synchronize()
    o.SPLITK_INKERNEL = saved
    for a, bb in outs:
        assert torch.equal(a[0], want[0]) and torch.equal(a[1], want[3]) and torch.equal(bb[0], want[1]) and torch.equal(bb[1], want[2])


# ------------------------------------------------------------------------------------ the 32x32x16 tile kernel (pf_gemm32.hip)


def _conv_ref_gpu(xs, wq, cout, ks, bias, *, stride=1, up=False, wrap=0, crop=0):
    """fp32 reference on the device: the same 16-bit-rounded operands through torch's fp32 convolution."""
    xin = torch.cat([x.float().permute(0, 3, 1, 2) for x in xs], 1)
    ctot = xin.shape[1]
    wref = wq.float().reshape(cout, ks, ks, ctot).permute(0, 3, 1, 2).contiguous()
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if wrap:
        xin = F.pad(xin, (wrap, wrap, 0, 0), mode="circular")
    y = F.conv2d(xin, wref, bias, stride=stride, padding=ks // 2)
    if crop:
        y = y[..., crop:-crop]
    return y.permute(0, 2, 3, 1).reshape(-1, cout)


@pytest.fixture
def gemm32_on(monkeypatch):
    """The 32x32x16 tile kernel is off by default (no faster than the 16x16 kernel under the power cap, profiles/r6_gemm32_power_cap.txt):
    PF_GEMM32 is read per launch, the cached plans of ops.conv_gemm are dropped on both sides of the switch."""
    o = ops()
    o._PLANS.plans.clear()
    monkeypatch.setenv("PF_GEMM32", "1")
    yield
    o._PLANS.plans.clear()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["s1_rowvec", "tail_res32", "cat", "up_n640", "s2", "pano_wrap_crop", "ragged", "res16", "pair_out"])
def test_conv_gemm32_vs_conv2d(dtype, case, gemm32_on):
    """The 32x32x16 kernel (256 x 320 tiles, four-slot ring of 32-wide K stages) on every addressing mode and operand mix of the layers it
    serves: whole rounds of 256 tiles, the split-K tail launch, channel concat, fused x2 upsampling, stride 2, the panorama's virtual
    circular padding, a ragged last row tile; bias / per-image row vector / 16-bit and fp32 residual / fp32 and pair outputs."""
    o = ops()
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    n, h, w, cin, cout, ks = 16, 64, 64, 320, 320, 3
    kw, ref_kw = {}, {}
    c1 = 0
    if case == "tail_res32":
        n = 20                                                     # 320 tiles: one round + 64 tiles x 4 K slices
    elif case == "cat":
        c1 = 320
    elif case == "up_n640":
        h = w = 32
        cout = 640
        kw["upsample"] = 1
        ref_kw["up"] = True
    elif case == "s2":
        n = 64
        kw["stride"] = 2
        ref_kw["stride"] = 2
    elif case == "pano_wrap_crop":
        n, w = 8, 128
        kw.update(wrap_pad=2, crop=2)
        ref_kw.update(wrap=2, crop=2)
    elif case == "ragged":
        n, w = 17, 62                                              # 67456 rows = 263 tiles + 128 rows: 256 + 8 tiles x 5 K slices
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g).to(dtype)
    a1 = torch.randn(n, h, w, c1, device=DEV, generator=g).to(dtype) if c1 else None
    ctot = cin + c1
    wq = (torch.randn(cout, ks * ks * ctot, device=DEV, generator=g) / (ks * ks * ctot) ** 0.5).to(dtype)
    b = torch.randn(cout, device=DEV, generator=g)
    want = _conv_ref_gpu([x] + ([a1] if c1 else []), wq, cout, ks, b, **ref_kw)
    M = want.shape[0]
    kw.update(n_img=n, h_in=h, w_in=w, ksize=ks, pad=1, bias=b, a1=a1)
    tol = TOL[dtype]
    if case == "s1_rowvec":
        table = torch.randn(n, cout, device=DEV, generator=g)
        kw["rowvec"] = table
        want = want + table.repeat_interleave(M // n, 0)
    elif case in ("tail_res32", "ragged"):
        r = torch.randn(M, cout, device=DEV, generator=g)
        kw["residual"] = r
        want = want + r
        tol = 2e-3 if dtype == torch.bfloat16 else 3e-4           # fp32 out: only the operands are rounded (K = 2880 products)
    elif case == "res16":
        r = torch.randn(M, cout, device=DEV, generator=g).to(dtype)
        kw["residual"] = r
        want = want + r.float()
    elif case == "pair_out":
        kw["split_out"] = True
    assert o.conv_gemm(x, wq, cout, plan_only=True, **kw) == 2, "this shape is expected to take the 32x32x16 kernel"
    got = o.conv_gemm(x, wq, cout, **kw)
    if case == "pair_out":
        hi, lo = unpair(got)
        got = hi.float() + lo.float()
        tol = 2e-3 if dtype == torch.bfloat16 else 3e-4
    # the reference differs from the kernel only by fp32 summation order; the operands are identical
    check("conv32 " + case, got, want, tol)
    check("conv32 " + case + " (last rows)", got[-300:], want[-300:], tol)
    check("conv32 " + case + " (first tile)", got[:256], want[:256], tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("f32out", [False, True])
def test_conv_gemm32_groupnorm_moments(dtype, f32out, gemm32_on):
    """The 32x32x16 kernel's GroupNorm-moment by-product (pf_conv_desc.gn_partial, runs of 64 rows): scale / shift from the moments equal
    the statistics pass over the stored tensor."""
    o = ops()
    n, h, w, cin, cout = 16, 64, 64, 320, 320
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g).to(dtype)
    wq = (torch.randn(cout, 9 * cin, device=DEV, generator=g) / (9 * cin) ** 0.5).to(dtype)
    b = torch.randn(cout, device=DEV, generator=g)
    table = torch.randn(n, cout, device=DEV, generator=g)
    kw = dict(n_img=n, h_in=h, w_in=w, ksize=3, pad=1, bias=b, rowvec=None if f32out else table, out_dtype=torch.float32 if f32out else None)
    assert o.conv_gemm(x, wq, cout, plan_only=True, **kw) == 2
    y = o.conv_gemm(x, wq, cout, gn_stats=True, **kw)
    assert hasattr(y, "_pf_gn") and y._pf_gn[1] == 64, "the 32x32x16 plan of a whole-round problem emits moments over runs of 64 rows"
    gamma, beta = torch.randn(cout, device=DEV, generator=g), torch.randn(cout, device=DEV, generator=g)
    sc, sh = o.groupnorm_scale_shift(y, None, n, h * w, 32, 1e-5, gamma, beta)
    yf = y.float().view(n, h * w, 32, cout // 32)
    mean = yf.mean(dim=(1, 3), keepdim=True)
    var = yf.var(dim=(1, 3), unbiased=False, keepdim=True)
    rstd = (var + 1e-5).rsqrt()
    sc_ref = (rstd.expand(n, 1, 32, cout // 32).reshape(n, cout) * gamma)
    sh_ref = beta - (mean.expand(n, 1, 32, cout // 32).reshape(n, cout)) * sc_ref
    # 16-bit outputs: the moments are taken BEFORE the rounding, the reference after it
    tol = 2e-5 if f32out else (3e-3 if dtype == torch.bfloat16 else 4e-4)
    check("scale", sc, sc_ref, tol)
    check("shift", sh, sh_ref, max(tol, 1e-4) * 10)


# ------------------------------------------------------------------------------------ sub-pixel upsampling convolution


@pytest.mark.parametrize("case", ["conv3_8wave", "conv3_4wave", "conv3_ragged", "concat", "stride2_pad_hi", "subpixel", "splitk", "split3_1x1"])
def test_conv_gemm_packed_tap_offsets_same_bits(case, monkeypatch):
    """GemmParams.fastseg: layers without upsampling / circular wrap keep a linear pixel index + one validity bit per tap row /
    column per staged tile row, and a tap change is an add, a multiply-add and a select.  Same addresses -> the results must be bit-identical to
    the general per-tap computation (PF_CONV_FASTSEG=0, read per call): 8-wave and 4-wave plans, a ragged last tile, a channel concat (two sources,
    two leading dimensions), stride 2 with the VAE's bottom / right zero row, the sub-pixel upsampling form, a split-K plan, a split-precision 1x1."""
    from panfusion_amd import engine
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(11)
    n, h, w, c0, c1, cout, ks, stride, pad, pad_hi = {
        "conv3_8wave": (8, 32, 32, 320, 0, 320, 3, 1, 1, 0), "conv3_4wave": (2, 12, 20, 128, 0, 192, 3, 1, 1, 0), "conv3_ragged": (3, 13, 17, 192, 0, 320, 3, 1, 1, 0),
        "concat": (4, 16, 16, 320, 192, 320, 3, 1, 1, 0), "stride2_pad_hi": (2, 32, 32, 128, 0, 128, 3, 2, 0, 1), "subpixel": (3, 10, 12, 128, 0, 192, 3, 1, 1, 0),
        "splitk": (2, 8, 8, 1280, 0, 1280, 3, 1, 1, 0), "split3_1x1": (1, 1, 4096, 320, 0, 320, 1, 1, 0, 0)}[case]
    x0 = torch.randn(n, h, w, c0, device=DEV, generator=g).half()
    x1 = torch.randn(n, h, w, c1, device=DEV, generator=g).half() if c1 else None
    bias = torch.randn(cout, device=DEV, generator=g)
    kw = dict(n_img=n, h_in=h, w_in=w, ksize=ks, stride=stride, pad=pad, bias=bias)
    if case == "subpixel":
        conv = torch.nn.Conv2d(c0, cout, 3, padding=1).to(DEV)
        wt = engine._subpixel_weight(conv, DEV, torch.float16)
        kw.update(upsample=1, subpixel=True)
    elif case == "split3_1x1":
        wt32 = torch.randn(cout, c0, device=DEV, generator=g) / c0 ** 0.5
        wt = engine._split_weight(wt32, 1, DEV, torch.float16)
        x0 = engine.split_operand(torch.randn(n * h * w, c0, device=DEV, generator=g), dtype=torch.float16)
        kw = dict(w_in=n * h * w, bias=bias, split3=True, out_dtype=torch.float32)
    else:
        wt = (torch.randn(cout, ks * ks * (c0 + c1), device=DEV, generator=g) / (ks * ks * (c0 + c1)) ** 0.5).half()
        if pad_hi:
            kw.update(pad_hi=1)
    outs = []
    for env in ("0", "1"):
        monkeypatch.setenv("PF_CONV_FASTSEG", env)
        outs.append(o.conv_gemm(x0, wt, cout, a1=x1, **kw).clone())
    assert torch.isfinite(outs[0].float()).all() and outs[0].float().abs().max() > 0
    assert torch.equal(outs[0], outs[1]), "packed tap offsets changed the result (%s): max |d| %.3e" % (case, (outs[0].float() - outs[1].float()).abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["small", "pano", "level1_f32_gn", "level0_16bit", "splitk"])
def test_conv_gemm_subpixel_upsample(dtype, case):
    """pf_conv_desc.subpixel: nearest x2 + 3x3 convolution as four 2x2 phase convolutions on the low-resolution grid (4 instead of 9 MACs
    per output value and input channel).  Against (a) the same four phase convolutions in fp32 torch on the SAME rounded weights and
    (b) the textbook F.interpolate + conv2d with the fp32 weights (the algebra: only the weights' rounding differs).  Cases: a small
    one (4-wave kernel), the panorama's pad 1 / upsample / conv / crop 2 (virtual circular padding), the 16^2 -> 32^2 level with fp32
    output and the GroupNorm moments, the 32^2 -> 64^2 level in 16 bit (persistent 8-wave kernel), a split-K plan (8^2 level)."""
    from panfusion_amd import engine
    o = ops()
    torch.backends.cudnn.allow_tf32 = False
    n, h, w, cin, cout, wrap, crop, f32, gn = {"small": (3, 10, 12, 128, 192, 0, 0, False, False), "pano": (2, 8, 20, 128, 192, 1, 2, False, False),
                                                "level1_f32_gn": (40, 16, 16, 1280, 1280, 0, 0, True, True), "level0_16bit": (40, 32, 32, 640, 640, 0, 0, False, False),
                                                "splitk": (2, 8, 16, 1280, 1280, 1, 2, True, False)}[case]
    g = torch.Generator(device=DEV).manual_seed(3)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (9 * cin) ** 0.5)
        conv.bias.copy_(torch.randn(cout, device=DEV, generator=g))
    x = torch.randn(n, h, w, cin, device=DEV, generator=g).to(dtype)
    w4 = engine._subpixel_weight(conv, DEV, dtype)
    assert w4.shape == (4 * cout, 4 * cin)
    kw = dict(n_img=n, h_in=h, w_in=w, ksize=3, pad=1, upsample=1, bias=conv.bias.detach(), wrap_pad=wrap, crop=crop,
              out_dtype=torch.float32 if f32 else None)
    got = o.conv_gemm(x, w4, cout, subpixel=True, gn_stats=gn, **kw)
    xin = x.float().permute(0, 3, 1, 2)
    if wrap:
        xin = torch.cat([xin[..., -wrap:], xin, xin[..., :wrap]], -1)
    # (a) the same four phase convolutions on the same rounded weights
    w4f = w4.float().reshape(4, cout, 2, 2, cin)
    ya = torch.zeros(n, cout, 2 * xin.shape[2], 2 * xin.shape[3], device=DEV)
    for a in range(2):
        for b in range(2):
            ya[:, :, a::2, b::2] = F.conv2d(F.pad(xin, (1 - b, b, 1 - a, a)), w4f[2 * a + b].permute(0, 3, 1, 2).contiguous())
    ya = ya + conv.bias.detach()[None, :, None, None]
    # (b) the textbook form on the fp32 weights
    yb = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode="nearest"), conv.weight.detach(), conv.bias.detach(), padding=1)
    if crop:
        ya, yb = ya[..., crop:-crop], yb[..., crop:-crop]
    ho, wo = ya.shape[2:]
    assert got.shape == (n * ho * wo, cout)
    tol = (2e-3 if dtype == torch.bfloat16 else 3e-4) if f32 else TOL[dtype]
    check("subpixel " + case + " vs phase convs", got, ya.permute(0, 2, 3, 1).reshape(-1, cout), tol)
    check("subpixel " + case + " vs interpolate + conv3x3", got, yb.permute(0, 2, 3, 1).reshape(-1, cout), 8e-3 if dtype == torch.bfloat16 else 1.2e-3)
    if gn:
        assert hasattr(got, "_pf_gn"), "whole-round sub-pixel problems emit GroupNorm moments"
        gamma, beta = torch.randn(cout, device=DEV, generator=g), torch.randn(cout, device=DEV, generator=g)
        sc, sh = o.groupnorm_scale_shift(got, None, n, ho * wo, 32, 1e-5, gamma, beta)
        yf = got.float().view(n, ho * wo, 32, cout // 32)
        mean, var = yf.mean(dim=(1, 3), keepdim=True), yf.var(dim=(1, 3), unbiased=False, keepdim=True)
        sc_ref = (var + 1e-5).rsqrt().expand(n, 1, 32, cout // 32).reshape(n, cout) * gamma
        sh_ref = beta - mean.expand(n, 1, 32, cout // 32).reshape(n, cout) * sc_ref
        check("subpixel moments: scale", sc, sc_ref, 2e-5)
        check("subpixel moments: shift", sh, sh_ref, 2e-4)
    if case == "splitk":
        assert o.gemm_workspace_bytes(x, w4, cout, subpixel=True, **{k: v for k, v in kw.items() if k not in ("bias", "out_dtype")}) > 0, \
            "this shape is expected to take a split-K plan"


@pytest.mark.parametrize("wrap", [False, True])
def test_conv_gemm_subpixel_upsample_in_split_precision(wrap):
    """The level-0 upsampling convolution of the mixed scheme: sub-pixel form AND split precision (pf_conv_desc.subpixel + split3) -- the
    pair operand of the fp32 stream against [W_hi | W_lo] of the four phase weights, three products per K block.  Reproduces the fp32
    convolution (interpolate + conv3x3 on the fp32 weights and the fp32 input) to ~1e-6, where the single-pass fp16 form sits at 3e-4."""
    from panfusion_amd import engine
    o = ops()
    torch.backends.cudnn.allow_tf32 = False
    dtype = torch.float16
    n, h, w, cin, cout = (2, 16, 32, 640, 640) if wrap else (6, 32, 32, 640, 640)
    g = torch.Generator(device=DEV).manual_seed(9)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (9 * cin) ** 0.5)
        conv.bias.copy_(torch.randn(cout, device=DEV, generator=g))
    x = torch.randn(n, h, w, cin, device=DEV, generator=g)                      # the fp32 stream
    w4s = engine._split_weight(engine._subpixel_weight(conv, DEV, torch.float32), 4, DEV, dtype)
    assert w4s.shape == (4 * cout, 4 * 2 * cin)
    geo = dict(wrap_pad=1, crop=2) if wrap else {}
    got = engine.exact_gemm(engine.split_operand(x, dtype=dtype), w4s, cout, n_img=n, h_in=h, w_in=w, ksize=3, pad=1, upsample=1,
                            bias=conv.bias.detach(), out_dtype=torch.float32, subpixel=True, **geo)
    xin = x.permute(0, 3, 1, 2)
    if wrap:
        xin = torch.cat([xin[..., -1:], xin, xin[..., :1]], -1)
    want = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode="nearest"), conv.weight.detach(), conv.bias.detach(), padding=1)
    if wrap:
        want = want[..., 2:-2]
    check("split-precision sub-pixel upsampling conv", got, want.permute(0, 2, 3, 1).reshape(-1, cout), 5e-6)
