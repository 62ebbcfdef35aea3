"""VAE decode on the HIP kernels (reference PanFusion.py:166-172, PanoGenerator.py:213-238)
against the CPU oracle restatement of diffusers' AutoencoderKL decoder (oracle/vae.py, parity unpinned like the
UNet) and fp32 torch statements of the new kernels.  Needs an MI355X: `-m gpu`."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from panfusion_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("rows,n", [(300, 4096), (7, 9216), (5, 12000), (64, 64)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_softmax_rows(rows, n, dtype):
    s = rnd(rows, n, seed=1, scale=30.0)
    got = ops().softmax_rows(s, 512 ** -0.5, dtype)
    want = torch.softmax(s * 512 ** -0.5, -1)
    assert got.dtype == dtype and rel_l2(got.cpu(), want.cpu()) <= (6e-4 if dtype == torch.float16 else 4e-3)
    assert float((got.float().sum(-1) - 1).abs().max()) < (2e-3 if dtype == torch.float16 else 2e-2)


def test_tensor_to_image_exact():
    x = rnd(3, 3, 40, 56, seed=2, scale=0.8)
    x[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, 1.0 / 255 - 1], device=DEV)
    got = ops().tensor_to_image(x)
    want = ((x / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1)
    assert got.dtype == torch.uint8 and torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_batched_scores_and_pv_gemm(dtype):
    """q k^T with fp32 output and P V through pf_conv_gemm's batch mode (the VAE attention's two GEMMs)."""
    o = ops()
    b, N, C = 3, 1024, 512
    q, k = rnd(b * N, C, seed=3).to(dtype), rnd(b * N, C, seed=4).to(dtype)
    s = o.conv_gemm(q, k, N, w_in=N, batch=b, a_bstride=N * C, w_bstride=N * C, out_bstride=N * N, out_dtype=torch.float32)
    want = torch.einsum("bik,bjk->bij", q.float().view(b, N, C), k.float().view(b, N, C))
    assert s.shape == (b, N, N) and rel_l2(s.cpu(), want.cpu()) <= 3e-6
    p = torch.softmax(want * C ** -0.5, -1).to(dtype)
    vt = rnd(b, C, N, seed=5).to(dtype)
    got = o.conv_gemm(p.view(b * N, N), vt, C, w_in=N, batch=b, a_bstride=N * N, w_bstride=C * N, out_bstride=N * C)
    wantv = torch.einsum("bij,bcj->bic", p.float(), vt.float())
    assert rel_l2(got.cpu(), wantv.cpu()) <= (6e-4 if dtype == torch.float16 else 4e-3)


def _models(cfg, seed):
    from oracle import sd2_unet as U
    from oracle import vae as OV
    from panfusion_amd.models.vae_params import VAEDecoderParams
    ov = OV.AutoencoderKLDecoder(**cfg)
    U.init_synthetic(ov, seed)
    params = VAEDecoderParams(**cfg)
    half = lambda keys: {k: v for k, v in ov.state_dict().items() if k.startswith(keys)}
    params.load_state_dict(half(("decoder.", "post_quant_conv.")), strict=True)      # identical key sets per half (diffusers names)
    return ov, params


def _encoder(ov, cfg):
    from panfusion_amd.models.vae_params import VAEEncoderParams
    enc = VAEEncoderParams(**cfg)
    enc.load_state_dict({k: v for k, v in ov.state_dict().items() if k.startswith(("encoder.", "quant_conv."))}, strict=True)
    return enc


@pytest.mark.parametrize("dtype,precision,tol", [(torch.float16, "mixed", 1e-3), (torch.float16, "fast", 4e-3),
                                                 (torch.bfloat16, "fast", 3e-2)])
def test_vae_decode_tiny_vs_oracle(dtype, precision, tol):
    from oracle import vae as OV
    from panfusion_amd import vae as PV
    ov, params = _models(OV.tiny_vae_config(width=64, groups=8), 61)
    dec = PV.VAEDecoder(params, compute_dtype=dtype, precision=precision)
    g = torch.Generator().manual_seed(6)
    lat, pano = torch.randn(1, 3, 4, 16, 16, generator=g), torch.randn(1, 1, 4, 16, 32, generator=g)
    with torch.no_grad():
        wi, wp = OV.decode_views_and_pano(lat, pano, ov, latent_pad=4)
    got = PV.decode_latent(lat.to(DEV), dec)
    err = rel_l2(got.cpu(), wi)
    print("VAE decode tiny %s/%s: rel-L2 %.3e" % (dtype, precision, err))
    assert err <= tol
    gi, gp = PV.decode_views_and_pano(lat.to(DEV), pano.to(DEV), dec, latent_pad=4)
    assert gp.shape == (1, 1, 128, 256, 3)
    assert float((gp.cpu().int() - OV.tensor_to_image(wp).int()).abs().float().mean()) < 1.0
    assert float((gi.cpu().int() - OV.tensor_to_image(wi).int()).abs().float().mean()) < 1.0


def test_vae_decode_sd2_widths_vs_oracle():
    """The SD-2 VAE decoder at its real widths (128 / 256 / 512 / 512, one attention head of width 512 over 4096
    tokens): one 64x64 view latent -> 512x512 image (1.27 TFLOP), HIP fp16 mixed vs the fp32 oracle."""
    from oracle import vae as OV
    from panfusion_amd import vae as PV
    ov, params = _models(dict(OV.SD2_VAE), 62)
    dec = PV.VAEDecoder(params, compute_dtype=torch.float16)
    lat = torch.randn(1, 1, 4, 64, 64, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        want = OV.decode_latent(lat, ov)
    got = PV.decode_latent(lat.to(DEV), dec)
    err = rel_l2(got.cpu(), want)
    print("VAE decode SD-2 widths, fp16 %s: rel-L2 %.3e" % (dec.precision, err))
    assert got.shape == (1, 1, 3, 512, 512) and torch.isfinite(got).all() and err <= 1e-3


# ------------------------------------------------------------------------------------ encoder (training images -> latents)
@pytest.mark.parametrize("dtype,precision,tol", [(torch.float16, "mixed", 1e-3), (torch.float16, "fast", 4e-3), (torch.bfloat16, "fast", 3e-2)])
def test_vae_encode_tiny_vs_oracle(dtype, precision, tol):
    """PanoGenerator.encode_image (PanoGenerator.py:214-225) on the views and on the circularly padded panorama of a training
    step (PanFusion.py:66-71): moments and the scaled sample for a given normal draw."""
    from oracle import vae as OV
    from panfusion_amd import vae as PV
    cfg = OV.tiny_vae_config(width=64, groups=8)
    ov, _ = _models(cfg, 71)
    enc = PV.VAEEncoder(_encoder(ov, cfg), compute_dtype=dtype, precision=precision)
    g = torch.Generator().manual_seed(8)
    imgs, pano = torch.rand(1, 3, 3, 64, 64, generator=g) * 2 - 1, torch.rand(1, 1, 3, 64, 192, generator=g) * 2 - 1
    for x in (imgs, pano):
        eps = torch.randn(x.shape[0], x.shape[1], 4, x.shape[3] // 8, x.shape[4] // 8, generator=g)
        with torch.no_grad():
            dist = ov.encode(x.flatten(0, 1)).latent_dist
            want = OV.encode_image(x, ov, eps=eps.flatten(0, 1))
        mean, logvar = enc.encode(x.flatten(0, 1).to(DEV))
        em, el = rel_l2(mean.cpu(), dist.mean), rel_l2(logvar.cpu(), dist.logvar)
        got = PV.encode_image(x.to(DEV), enc, eps=eps.to(DEV))
        ez = rel_l2(got.cpu(), want)
        print("VAE encode tiny %s/%s %s: mean %.2e logvar %.2e sample %.2e" % (dtype, precision, tuple(x.shape[-2:]), em, el, ez))
        assert got.shape == want.shape and max(em, el, ez) <= tol


def test_vae_encode_sd2_widths_vs_oracle():
    """The SD-2 VAE encoder at its real widths on one 256^2 training view (0.57 TFLOP): fp16 mixed vs the fp32 oracle."""
    from oracle import vae as OV
    from panfusion_amd import vae as PV
    cfg = dict(OV.SD2_VAE)
    ov, _ = _models(cfg, 72)
    enc = PV.VAEEncoder(_encoder(ov, cfg), compute_dtype=torch.float16)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 1, 3, 256, 256, generator=g) * 2 - 1
    eps = torch.randn(1, 1, 4, 32, 32, generator=g)
    with torch.no_grad():
        want = OV.encode_image(x, ov, eps=eps.flatten(0, 1))
    got = PV.encode_image(x.to(DEV), enc, eps=eps.to(DEV))
    err = rel_l2(got.cpu(), want)
    print("VAE encode SD-2 widths, fp16 %s: rel-L2 %.3e" % (enc.precision, err))
    assert got.shape == (1, 1, 4, 32, 32) and err <= 1e-3
