"""The mixed-precision scheme (fp32 residual streams + split-precision stream-path GEMMs, engine.py) on the
GPU: its kernels against fp32 torch statements, and END-TO-END PARITY AT SD-2-BASE WIDTHS against the CPU
oracle with the error budget printed block by block (north_star: <= 1e-3 rel-L2).  Needs an MI355X: `-m gpu`."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2, unpair

pytestmark = pytest.mark.gpu
DEV = "cuda"
D16 = [torch.float16, torch.bfloat16]
ONE_ROUNDING = {torch.bfloat16: 4e-3, torch.float16: 6e-4}


def ops():
    from panfusion_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# ------------------------------------------------------------------------------------ element-wise kernels
@pytest.mark.parametrize("dtype", D16)
def test_groupnorm_and_apply_on_fp32_stream(dtype):
    """GN statistics of an fp32 two-source concat; apply to 16-bit, to fp32 and to the split pair."""
    o = ops()
    n, hw, c0, c1, groups = 3, 24 * 17, 64, 32, 32
    x0, x1 = rnd(n, hw, c0, seed=1, scale=2.0) + 0.5, rnd(n, hw, c1, seed=2)
    gamma, beta = rnd(c0 + c1, seed=3) * 0.1 + 1, rnd(c0 + c1, seed=4) * 0.1
    sc, sh = o.groupnorm_scale_shift(x0, x1, n, hw, groups, 1e-5, gamma, beta)
    x = torch.cat([x0, x1], -1)
    want = F.silu(F.group_norm(x.transpose(1, 2), groups, gamma, beta, 1e-5).transpose(1, 2))
    y32 = o.scale_shift_act(x0, x1, n, hw, sc, sh, 1, out_dtype=torch.float32)
    assert y32.dtype == torch.float32 and rel_l2(y32.cpu(), want.cpu()) <= 2e-6
    y16 = o.scale_shift_act(x0, x1, n, hw, sc, sh, 1, out_dtype=dtype)
    assert y16.dtype == dtype and rel_l2(y16.cpu(), want.cpu()) <= ONE_ROUNDING[dtype]
    pair = o.scale_shift_act(x0, x1, n, hw, sc, sh, 1, out_dtype=dtype, split=True)
    C = c0 + c1
    assert pair.shape == (n, hw, 2 * C)
    hi, lo = unpair(pair)
    assert torch.equal(hi, y32.to(dtype))                                   # hi is the plain rounding ...
    assert torch.equal(lo, (y32 - hi.float()).to(dtype))                    # ... lo the rounded remainder
    # identity (no statistics): the raw stream as a split operand
    raw = o.scale_shift_act(x0, None, 1, n * hw, None, None, 0, out_dtype=dtype, split=True)
    assert torch.equal(unpair(raw)[0].reshape(n, hw, c0), x0.to(dtype))
    # 16-bit source, split output
    x16 = x0.to(dtype)
    p16 = o.scale_shift_act(x16, None, 1, n * hw, None, None, 0, split=True)
    assert torch.equal(unpair(p16)[0].reshape(n, hw, c0), x16) and float(unpair(p16)[1].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", D16)
@pytest.mark.parametrize("rows,C", [(5000, 320), (77, 1280), (4096, 64)])
def test_layernorm_fp32_input(dtype, rows, C):
    o = ops()
    x = rnd(rows, C, seed=5, scale=3.0)
    g, b = rnd(C, seed=6) * 0.1 + 1, rnd(C, seed=7) * 0.1
    pe = rnd(rows // 7 if rows % 7 == 0 else rows, C, seed=8)
    got = o.layernorm(x, g, b, 1e-5, pe=pe, out_dtype=dtype)
    want = F.layer_norm(x + pe.repeat(rows // pe.shape[0], 1), (C,), g, b, 1e-5)
    assert got.dtype == dtype and rel_l2(got.cpu(), want.cpu()) <= ONE_ROUNDING[dtype]


def test_add_pad_crop_conv_in_out_fp32():
    o = ops()
    a, b = rnd(3, 8, 12, 64, seed=9), rnd(3, 8, 12, 64, seed=10).half()
    assert torch.equal(o.add(a, b), a + b.float())
    assert torch.equal(o.add(b, a), (b.float() + a).half())
    padded = o.pad_width(a, 2)
    assert torch.equal(padded, torch.cat([a[:, :, -2:], a, a[:, :, :2]], 2))
    assert torch.equal(o.crop_width(padded, 2), a)
    x = rnd(2, 4, 8, 16, seed=11)
    w, bias = rnd(3, 3, 4, 64, seed=12) * 0.2, rnd(64, seed=13)
    for wrap in (False, True):
        xi = torch.cat([x[..., -1:], x, x[..., :1]], -1) if wrap else x
        want = F.conv2d(xi, w.permute(3, 2, 0, 1), bias, padding=1)
        want = want[..., 1:-1] if wrap else want
        got = o.conv_in(x, w, bias, 64, torch.float32, wrap=wrap)
        assert got.dtype == torch.float32 and rel_l2(got.permute(0, 3, 1, 2).cpu(), want.cpu()) <= 2e-6
    y = rnd(2, 8, 16, 64, seed=14)
    wo, bo = rnd(4, 3, 3, 64, seed=15) * 0.1, rnd(4, seed=16)
    got = o.conv_out(y, wo, bo, 4)
    want = F.conv2d(y.permute(0, 3, 1, 2), wo.permute(0, 3, 1, 2), bo, padding=1)
    assert rel_l2(got.cpu(), want.cpu()) <= 2e-6


# ------------------------------------------------------------------------------------ GEMM epilogues
@pytest.mark.parametrize("dtype", D16)
@pytest.mark.parametrize("M,N,K", [(40 * 1024, 640, 640),      # 8-wave persistent kernel
                                   (4096, 320, 1280),          # 4-wave kernel
                                   (1000, 320, 5120),          # 4-wave, split-K (the reduce kernel's epilogue)
                                   (333, 132, 64)])            # ragged tile, N % 32 != 0
def test_gemm_fp32_residual_and_output(dtype, M, N, K):
    """bias + fp32 residual -> fp32 output (the residual stream of the mixed scheme) and bias -> fp32."""
    o = ops()
    x = rnd(M, K, seed=20).to(dtype)
    w = (rnd(N, K, seed=21) / K ** 0.5).to(dtype)
    b, res = rnd(N, seed=22), rnd(M, N, seed=23)
    prod = x.float() @ w.float().T
    got = o.linear(x, w, bias=b, residual=res)
    assert got.dtype == torch.float32 and torch.isfinite(got).all()
    assert rel_l2(got.cpu(), (prod + b + res).cpu()) <= 3e-6
    got = o.linear(x, w, bias=b, out_dtype=torch.float32)
    assert rel_l2(got.cpu(), (prod + b).cpu()) <= 3e-6
    # a 16-bit residual with fp32 output still works (generic epilogue)
    r16 = res.to(dtype)
    got = o.linear(x, w, bias=b, residual=r16, out_dtype=torch.float32)
    assert rel_l2(got.cpu(), (prod + b + r16.float()).cpu()) <= 3e-6


@pytest.mark.parametrize("dtype", D16)
@pytest.mark.parametrize("M,N,K", [(40 * 1024, 640, 2560), (4096, 320, 1280), (1000, 320, 5120), (333, 128, 64)])
def test_gemm_split_pair_epilogue(dtype, M, N, K):
    """PF_EPILOGUE_SPLIT: bias (+ fp32 residual) -> the 16-bit pair of the fp32 result, per block of 32 columns [hi | lo] (FF2
    feeding the split-precision proj_out); hi is the 16-bit rounding of the fp32 epilogue output, lo the rounded remainder."""
    o = ops()
    x = rnd(M, K, seed=40).to(dtype)
    w = (rnd(N, K, seed=41) / K ** 0.5).to(dtype)
    b, res = rnd(N, seed=42), rnd(M, N, seed=43)
    for r in (res, None):
        want = o.linear(x, w, bias=b, residual=r, out_dtype=torch.float32)
        pair = o.linear(x, w, bias=b, residual=r, split_out=True)
        assert pair.dtype == dtype and pair.shape == (M, 2 * N)
        hi, lo = unpair(pair)
        assert torch.equal(hi, want.to(dtype))
        assert torch.equal(lo, (want - hi.float()).to(dtype))


@pytest.mark.parametrize("dtype", D16)
def test_split_precision_gemm_reproduces_fp32(dtype):
    """exact_gemm: pair operand x [W_hi | W_lo] weights, three products per K block in one launch == the fp32 product to ~2^-2p."""
    from panfusion_amd import engine
    o = ops()
    # 1x1: resnet shortcut over a channel concat of two fp32 stream tensors
    n, hw, c0, c1, N = 4, 32 * 32, 640, 320, 320
    x0, x1 = rnd(n, hw, c0, seed=30, scale=2.0), rnd(n, hw, c1, seed=31)
    w = rnd(N, c0 + c1, seed=32) / (c0 + c1) ** 0.5
    b = rnd(N, seed=33)
    want = torch.cat([x0, x1], -1).reshape(-1, c0 + c1).double() @ w.double().T + b.double()
    a = engine.split_operand(x0, x1, dtype=dtype)
    w3 = engine._split_weight(w, 1, DEV, dtype)
    got = engine.exact_gemm(a, w3, N, w_in=n * hw, bias=b, out_dtype=torch.float32)
    plain = o.linear(torch.cat([x0, x1], -1).reshape(-1, c0 + c1).to(dtype), w.to(dtype), bias=b, out_dtype=torch.float32)
    e_exact, e_plain = rel_l2(got.cpu(), want.float().cpu()), rel_l2(plain.cpu(), want.float().cpu())
    print("split-precision 1x1 %s: %.2e (single pass %.2e)" % (dtype, e_exact, e_plain))
    assert e_exact <= (2e-6 if dtype == torch.float16 else 4e-5) and e_exact < e_plain / 20
    # 3x3 stride 2 (downsampling conv) with a fp32 residual on top
    n, h, wd, C = 3, 16, 24, 128
    x = rnd(n, h, wd, C, seed=34)
    wt = rnd(C, C, 3, 3, seed=35) / (9 * C) ** 0.5
    wantc = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), b[:C].double(), stride=2, padding=1).permute(0, 2, 3, 1)
    w3 = engine._split_weight(wt.permute(0, 2, 3, 1).reshape(C, -1), 9, DEV, dtype)
    gotc = engine.exact_gemm(engine.split_operand(x, dtype=dtype), w3, C, n_img=n, h_in=h, w_in=wd, ksize=3, stride=2, pad=1,
                             bias=b[:C].contiguous(), out_dtype=torch.float32)
    e = rel_l2(gotc.view(wantc.shape).cpu(), wantc.float().cpu())
    assert e <= (2e-6 if dtype == torch.float16 else 4e-5), e


# ------------------------------------------------------------------------------------ end to end, full widths
BLOCKS = ("__init__", "resnet", "attention", "downsample", "upsample")


class _Recorder:
    """Block outputs of the oracle (NCHW fp32) and of the HIP branches (NHWC stream dtype), in driver order."""

    def __init__(self):
        self.items = []

    def wrap(self, cls, to_nchw):
        saved = {n: getattr(cls, n) for n in BLOCKS}
        rec = self

        def make(name):
            def f(self, *a, **k):
                out = saved[name](self, *a, **k)
                x = self.h
                pano = getattr(self, "pano", None)
                if pano is None:
                    pano = bool(self.pad) or getattr(self, "_is_pano", False)
                rec.items.append(("%s.%s" % ("pano" if pano else "pers", name.strip("_")), to_nchw(x)))
                return out
            return f
        for n in BLOCKS:
            setattr(cls, n, make(n))
        return saved

    @staticmethod
    def restore(cls, saved):
        for n, f in saved.items():
            setattr(cls, n, f)


@pytest.fixture(scope="module")
def full_width():
    """SD-2-base widths, one CFG sample, m = 2 views of 64x64 latents + the 64x128 panorama latent; seeded synthetic
    weights (LoRA attached, EPA output projections re-randomised).  The oracle forward runs once."""
    from oracle import mvgen as MV
    from oracle import sd2_unet as U
    torch.manual_seed(0)
    cfg = dict(U.SD2_BASE)
    unet, pano_unet = U.UNet2DConditionModel(**cfg), U.UNet2DConditionModel(**cfg)
    unet.add_lora(4)
    pano_unet.add_lora(4)
    U.init_synthetic(unet, 101)
    U.init_synthetic(pano_unet, 102)
    om = MV.DualBranchDenoiser(unet, pano_unet, None, None, True)
    U.init_synthetic(om.cp_blocks_encoder, 103)
    U.init_synthetic(om.cp_blocks_mid, 104)
    U.init_synthetic(om.cp_blocks_decoder, 105)
    MV.randomize_epa(om, 106)
    g = torch.Generator().manual_seed(7)
    b, m = 1, 2
    inp = dict(lat=torch.randn(b, m, 4, 64, 64, generator=g), pl=torch.randn(b, 1, 4, 64, 128, generator=g),
               pe=torch.randn(b, m, 77, 1024, generator=g), ppe=torch.randn(b, 1, 77, 1024, generator=g),
               t=torch.full((b, m), 981, dtype=torch.long),
               cams={"FoV": torch.full((b, m), 90), "theta": torch.tensor([[36.0, 180.0]], dtype=torch.float64),
                     "phi": torch.tensor([[52.6, -10.8]], dtype=torch.float64)})
    rec = _Recorder()
    saved = rec.wrap(MV._Branch, lambda x: x.detach().float().clone())
    try:
        with torch.no_grad():
            ws, wp = om(inp["lat"], inp["pl"], inp["t"], inp["pe"], inp["ppe"], inp["cams"])
    finally:
        rec.restore(MV._Branch, saved)
    return om, inp, (ws, wp), rec.items


@pytest.mark.parametrize("dtype,precision,tol", [(torch.float16, "mixed", 1e-3), (torch.float16, "fast", 4e-3),
                                                 (torch.bfloat16, "mixed", 1.5e-2), (torch.bfloat16, "fast", 3e-2)])
def test_full_width_denoiser_vs_oracle(full_width, dtype, precision, tol):
    """north_star parity bar at the benchmark WIDTHS: rel-L2 of both epsilon outputs against the fp32 oracle, and
    the error after every block of both branches (the error budget as a table, printed with -s)."""
    from panfusion_amd import engine
    from panfusion_amd.models.pano import MultiViewBaseModel
    om, inp, (ws, wp), oracle_blocks = full_width
    model = MultiViewBaseModel(om.unet, om.pano_unet, None, None, True, compute_dtype=dtype, precision=precision)
    model.two_streams = False                                   # one stream: block order == the oracle's
    model.load_state_dict({k: v for k, v in om.state_dict().items() if k.startswith("cp_blocks")}, strict=False)
    rec = _Recorder()
    saved = rec.wrap(engine.Branch, lambda x: x.detach().float().permute(0, 3, 1, 2).cpu())
    try:
        d = lambda x: x.to(DEV)
        s, ps = model(d(inp["lat"]), d(inp["pl"]), d(inp["t"]), d(inp["pe"]), d(inp["ppe"]), inp["cams"])
    finally:
        rec.restore(engine.Branch, saved)
    es, ep = rel_l2(s.cpu(), ws), rel_l2(ps.cpu(), wp)
    print("\nfull-width e2e  %s / %s:  views %.3e  pano %.3e   (tolerance %.1e)" % (dtype, precision, es, ep, tol))
    # the driver's block order differs between the two implementations only in how branches interleave:
    # compare per branch, in order
    for tag in ("pers", "pano"):
        a = [(n, x) for n, x in oracle_blocks if n.startswith(tag)]
        bb = [(n, x) for n, x in rec.items if n.startswith(tag)]
        assert [n for n, _ in a] == [n for n, _ in bb], "block sequences differ"
        print("  %s branch, rel-L2 after each block:" % tag)
        print("   " + "  ".join("%s %.1e" % (n.split(".")[1][:4], rel_l2(y, x)) for (n, x), (_, y) in zip(a, bb)))
    assert torch.isfinite(s).all() and torch.isfinite(ps).all()
    assert es <= tol and ep <= tol, (es, ep)
