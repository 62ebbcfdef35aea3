"""The VAE on the HIP kernels: DECODE of the sampled latents -- the step right after the denoising loop
(reference ``models/pano/PanFusion.py:166-172``, ``PanoGenerator.py:213-238``,
``models/modules/utils.py:9-15``) -- and ENCODE of the training images -- the step right before the denoiser call of
a training step (``PanFusion.py:66-71``, ``PanoGenerator.encode_image``, ``PanoGenerator.py:214-225``; VAEEncoder below).

``VAEDecoder(vae)`` takes a module tree with diffusers' ``AutoencoderKL`` attribute names (diffusers' own object,
or ``models.vae_params.VAEDecoderParams``), repacks ``post_quant_conv`` + ``decoder`` once into the kernel layouts
and runs ``decode`` as a sequence of C-ABI calls: the 3x3 convolutions / upsampling convolutions on the MFMA
implicit-GEMM kernel, GroupNorm + SiLU on the statistics / apply kernels, the 4-channel input convolution and the
3-channel output convolution on the boundary kernels, and the mid-block attention (ONE head of width 512) as
two batched GEMMs with a row-softmax kernel in between (its 4096 x 4096 scores per view are materialised in fp32:
1.3 GB for 20 views -- memory is laid out for 288 GB of HBM, not for a flash kernel at a head width no MFMA
fragment mapping of ``k_attention_lds`` covers).  Same two storage schemes as the denoiser (engine.py): "mixed"
(fp32 residual stream, split-precision shortcut convolutions; default for fp16) or "fast".
"""
from types import SimpleNamespace as NS

import torch

from . import engine, ops


def _conv1_as_3x3(conv, dev, cout_pad):
    """1x1 conv (post_quant_conv, 4 -> 4) as the centre tap of a 3x3 kernel for pf_conv_in: weights
    [3][3][cin][cout_pad], bias [cout_pad] (output channels zero-padded to the kernel's multiple of 8)."""
    co, ci = conv.weight.shape[:2]
    w = torch.zeros(3, 3, ci, cout_pad)
    w[1, 1, :, :co] = conv.weight.detach().float().reshape(co, ci).t()
    b = torch.zeros(cout_pad)
    b[:co] = conv.bias.detach().float()
    return w.to(dev), b.to(dev)


def pack_vae_decoder(vae, dev, dtype, mixed):
    d = vae.decoder
    v = NS(dtype=dtype, mixed=mixed, stream=torch.float32 if mixed else dtype)
    v.scaling_factor = float(vae.config.scaling_factor)
    v.lat_c = vae.post_quant_conv.weight.shape[1]
    v.pq_w, v.pq_b = _conv1_as_3x3(vae.post_quant_conv, dev, 8)
    ci = d.conv_in
    v.c_top = ci.weight.shape[0]
    w_in = torch.zeros(3, 3, 8, v.c_top)                                   # reads the 8-channel padded post_quant output
    w_in[:, :, :ci.weight.shape[1]] = ci.weight.detach().float().permute(2, 3, 1, 0)
    v.w_in, v.b_in = w_in.to(dev), engine._bias(ci, dev)

    def res(r):
        p = engine.pack_resnet_plain(r, dev, dtype, mixed)
        return p

    v.mid_res = [res(r) for r in d.mid_block.resnets]
    a = d.mid_block.attentions[0]
    Cc = a.to_q.weight.shape[0]
    att = NS(C=Cc, norm=engine._norm(a.group_norm, dev))
    att.wq, att.bq = engine._w16(a.to_q.weight, dev, dtype), engine._bias(a.to_q, dev)
    att.wk, att.bk = engine._w16(a.to_k.weight, dev, dtype), engine._bias(a.to_k, dev)
    att.wv = engine._w16(a.to_v.weight, dev, dtype)
    wo = a.to_out[0].weight.detach().float()
    # softmax rows sum to one: P (V0 + 1 b_v^T) = P V0 + 1 b_v^T, so the value bias moves behind the output projection
    att.wo = engine._w16(wo, dev, dtype)
    att.bo = (a.to_out[0].bias.detach().float() + wo @ a.to_v.bias.detach().float()).to(dev).contiguous()
    v.att = att
    v.up = []
    for blk in d.up_blocks:
        b = NS(resnets=[res(r) for r in blk.resnets], up=None)
        if blk.upsamplers is not None:
            c = blk.upsamplers[0].conv
            b.up = NS(w=engine._conv3_weight(c, dev, dtype), b=engine._bias(c, dev), c=c.weight.shape[0])
        v.up.append(b)
    v.c_level1 = d.up_blocks[-2].resnets[0].conv1.weight.shape[0] if len(d.up_blocks) > 1 else v.c_top
    v.norm_out = engine._norm(d.conv_norm_out, dev)
    co = d.conv_out
    v.c_img = co.weight.shape[0]
    v.w_out, v.b_out = engine._f32(co.weight.detach().permute(0, 2, 3, 1), dev), engine._bias(co, dev)
    return v


def _attention(att, x, dtype):
    """diffusers Attention(heads=1, dim_head=C, residual_connection=True) on x [n, h, w, C] (stream dtype)."""
    n, h, w, Cc = x.shape
    N = h * w
    if N % 64:
        raise ValueError("VAE attention needs h*w to be a multiple of 64 (got %dx%d)" % (h, w))
    sc, sh = ops.groupnorm_scale_shift(x, None, n, N, att.norm.groups, att.norm.eps, att.norm.g, att.norm.b)
    t = ops.scale_shift_act(x, None, n, N, sc, sh, 0, out_dtype=dtype).view(n * N, Cc)
    q = ops.linear(t, att.wq, bias=att.bq)
    k = ops.linear(t, att.wk, bias=att.bk)
    vt = ops.linear_t(t.view(n, N, Cc), att.wv, ld=N)                      # [n, C, N]: keys contiguous
    out = torch.empty(n * N, Cc, device=x.device, dtype=x.dtype)
    xs = x.view(n * N, Cc)
    # a few images at a time: the fp32 score matrix of one image is N^2 * 4 bytes (64 MB at 64x64, 340 MB for the
    # padded panorama); 8 of them keep the GEMMs large without holding more than ~2.7 GB
    per = max(1, min(n, (1 << 31) // (N * N * 4))) if N * N * 4 < (1 << 31) else 1
    per = min(per, 8)
    for i0 in range(0, n, per):
        b = min(per, n - i0)
        rows = slice(i0 * N, (i0 + b) * N)
        s = ops.conv_gemm(q[rows], k[rows], N, w_in=N, batch=b, a_bstride=N * Cc, w_bstride=N * Cc, out_bstride=N * N,
                          out_dtype=torch.float32)                         # [b, N, N] = q k^T
        p = ops.softmax_rows(s.view(b, N, N), Cc ** -0.5, dtype)
        o = ops.conv_gemm(p.view(b * N, N), vt[i0:i0 + b], Cc, w_in=N, batch=b, a_bstride=N * N, w_bstride=Cc * N,
                          out_bstride=N * Cc)                              # [b, N, C] = P V0
        ops.linear(o.view(b * N, Cc), att.wo, bias=att.bo, residual=xs[rows], out=out[rows])
    return out.view(n, h, w, Cc)


class VAEDecoder:
    def __init__(self, vae, compute_dtype=torch.float16, precision=None):
        self.vae, self.compute_dtype = vae, compute_dtype
        self.precision = precision or engine.default_precision(compute_dtype)
        self._packed = {}

    def packed(self, device):
        key = (str(device), self.compute_dtype, self.precision)
        if key not in self._packed:
            self._packed[key] = pack_vae_decoder(self.vae, device, self.compute_dtype, self.precision == "mixed")
        return self._packed[key]

    def repack(self):
        self._packed.clear()

    @torch.no_grad()
    def decode(self, z, chunk=None):
        """``vae.decode(z).sample``: z (n, 4, h, w) on the GPU -> image (n, 3, 8h, 8w) fp32.  chunk: images per pass
        (default: as many as keep every GEMM operand under the kernel's 2 GiB addressing limit -- 7 of the 512^2
        views at a time, the padded panorama alone)."""
        v = self.packed(z.device)
        z = z.float().contiguous()
        n = z.shape[0]
        if chunk is None:
            # pf_conv_gemm addresses an operand with 32-bit byte offsets (< 2 GiB): the largest one is the split pair
            # [hi | lo] of the second-finest level's width at full resolution (8h x 8w x 2 C1 16-bit values per image)
            per_image = 64 * z.shape[2] * z.shape[3] * 4 * v.c_level1
            chunk = max(1, ((1 << 31) - 1) // per_image)
        if n > chunk:
            return torch.cat([self.decode(z[i:i + chunk], chunk) for i in range(0, n, chunk)])
        dt = v.dtype
        # post_quant_conv (1x1 as the centre tap of the 3x3 boundary kernel), then conv_in on its NCHW view
        pq = ops.conv_in(z, v.pq_w, v.pq_b, 8, torch.float32)              # NHWC [n, h, w, 8] fp32
        x = ops.conv_in(ops.nhwc_to_nchw(pq, torch.float32), v.w_in, v.b_in, v.c_top, v.stream)
        x = engine.run_resnet_plain(v.mid_res[0], x)
        x = _attention(v.att, x, dt)
        x = engine.run_resnet_plain(v.mid_res[1], x)
        for blk in v.up:
            for r in blk.resnets:
                x = engine.run_resnet_plain(r, x)
            if blk.up is not None:
                nn_, h, w, Cc = x.shape
                y = ops.conv_gemm(engine.to16(x, dt), blk.up.w, blk.up.c, n_img=nn_, h_in=h, w_in=w, ksize=3, pad=1, upsample=1,
                                  bias=blk.up.b, out_dtype=v.stream)
                x = y.view(nn_, 2 * h, 2 * w, blk.up.c)
        nn_, h, w, Cc = x.shape
        sc, sh = ops.groupnorm_scale_shift(x, None, nn_, h * w, v.norm_out.groups, v.norm_out.eps, v.norm_out.g, v.norm_out.b)
        y = ops.scale_shift_act(x, None, nn_, h * w, sc, sh, 1, out_dtype=v.stream).view(nn_, h, w, Cc)
        return ops.conv_out(y, v.w_out, v.b_out, v.c_img)                  # fp32 NCHW


def decode_latent(latents, decoder):
    """``PanoGenerator.decode_latent(latents, vae)`` (PanoGenerator.py:213-220): (b, m, c, h, w) -> (b, m, 3, 8h, 8w)."""
    b = latents.shape[0]
    z = (latents.float() * (1.0 / decoder.packed(latents.device).scaling_factor)).flatten(0, 1)
    return decoder.decode(z).unflatten(0, (b, -1)).to(latents.dtype)


def decode_views_and_pano(latents, pano_latent, decoder, latent_pad=8):
    """The tail of ``PanFusion.inference`` (PanFusion.py:166-172): the m views as they are; the panorama latent
    circularly padded by ``latent_pad`` columns (``pad_pano``, utils/pano.py:74-99), decoded, the image cropped by
    8 * latent_pad pixels per side (``unpad_pano``).  Returns uint8 images (b, m, H, W, 3) and (b, 1, H, W, 3)
    (``tensor_to_image``, models/modules/utils.py:9-15)."""
    from .utils.pano import pad_pano, unpad_pano
    images = decode_latent(latents, decoder)
    pano = unpad_pano(decode_latent(pad_pano(pano_latent, latent_pad), decoder), 8 * latent_pad)
    to_u8 = lambda x: ops.tensor_to_image(x.flatten(0, 1)).unflatten(0, x.shape[:2])
    return to_u8(images), to_u8(pano.contiguous())


# ---------------------------------------------------------------------------------------------- encoder (training)
def pack_vae_encoder(vae, dev, dtype, mixed):
    """``vae.encoder`` + ``vae.quant_conv`` (diffusers AutoencoderKL names) -> kernel layouts."""
    e = vae.encoder
    v = NS(dtype=dtype, mixed=mixed, stream=torch.float32 if mixed else dtype)
    v.scaling_factor = float(vae.config.scaling_factor)
    ci = e.conv_in
    v.c0 = ci.weight.shape[0]
    v.w_in, v.b_in = engine._f32(ci.weight.detach().permute(2, 3, 1, 0), dev), engine._bias(ci, dev)       # [3, 3, cin, cout]
    res = lambda r: engine.pack_resnet_plain(r, dev, dtype, mixed)
    v.down = []
    for blk in e.down_blocks:
        b = NS(resnets=[res(r) for r in blk.resnets], down=None)
        if blk.downsamplers is not None:
            c = blk.downsamplers[0].conv
            b.down = NS(w=engine._conv3_weight(c, dev, dtype), b=engine._bias(c, dev), c=c.weight.shape[0], w3=None)
            if mixed:                 # stream -> stream linear map: split precision, as the UNet's Downsample2D (engine.pack_unet)
                b.down.w3 = engine._split_weight(c.weight.detach().float().permute(0, 2, 3, 1).reshape(c.weight.shape[0], -1), 9, dev, dtype)
        v.down.append(b)
    v.mid_res = [res(r) for r in e.mid_block.resnets]
    a = e.mid_block.attentions[0]
    att = NS(C=a.to_q.weight.shape[0], norm=engine._norm(a.group_norm, dev))
    att.wq, att.bq = engine._w16(a.to_q.weight, dev, dtype), engine._bias(a.to_q, dev)
    att.wk, att.bk = engine._w16(a.to_k.weight, dev, dtype), engine._bias(a.to_k, dev)
    att.wv = engine._w16(a.to_v.weight, dev, dtype)
    wo = a.to_out[0].weight.detach().float()
    att.wo = engine._w16(wo, dev, dtype)
    att.bo = (a.to_out[0].bias.detach().float() + wo @ a.to_v.bias.detach().float()).to(dev).contiguous()
    v.att = att
    v.norm_out = engine._norm(e.conv_norm_out, dev)
    co = e.conv_out
    v.c_mom = co.weight.shape[0]                                            # 2 * latent channels
    v.w_out, v.b_out = engine._f32(co.weight.detach().permute(0, 2, 3, 1), dev), engine._bias(co, dev)
    v.q_w, v.q_b = _conv1_as_3x3(vae.quant_conv, dev, v.c_mom)
    return v


class VAEEncoder:
    """``vae.encode(x).latent_dist`` on the HIP kernels (the top of every training step, PanFusion.py:66-71):
    ``moments(x)`` -> fp32 NHWC (mean | logvar), ``encode(x)`` -> (mean, logvar) NCHW, ``sample(x, eps)`` -> z."""

    def __init__(self, vae, compute_dtype=torch.float16, precision=None):
        self.vae, self.compute_dtype = vae, compute_dtype
        self.precision = precision or engine.default_precision(compute_dtype)
        self._packed = {}

    def packed(self, device):
        key = (str(device), self.compute_dtype, self.precision)
        if key not in self._packed:
            self._packed[key] = pack_vae_encoder(self.vae, device, self.compute_dtype, self.precision == "mixed")
        return self._packed[key]

    def repack(self):
        self._packed.clear()

    @torch.no_grad()
    def moments(self, x, chunk=None):
        """x (n, 3, H, W) in [-1, 1] on the GPU -> fp32 NHWC [n, H/8, W/8, 2 L] = quant_conv(encoder(x))."""
        v = self.packed(x.device)
        x = x.float().contiguous()
        n, _, H, W = x.shape
        if chunk is None:             # 2 GiB operand addressing of pf_conv_gemm: the split pair of the first level at full size
            chunk = max(1, ((1 << 31) - 1) // (H * W * 4 * v.c0))
        if n > chunk:
            return torch.cat([self.moments(x[i:i + chunk], chunk) for i in range(0, n, chunk)])
        dt = v.dtype
        h = ops.conv_in(x, v.w_in, v.b_in, v.c0, v.stream)
        for blk in v.down:
            for r in blk.resnets:
                h = engine.run_resnet_plain(r, h)
            if blk.down is not None:                                        # F.pad(x, (0, 1, 0, 1)) + conv3x3 stride 2, no padding
                nn_, hh, ww, Cc = h.shape
                d = blk.down
                kw = dict(n_img=nn_, h_in=hh, w_in=ww, ksize=3, stride=2, pad=0, pad_hi=1, bias=d.b)
                if d.w3 is not None:
                    y = engine.exact_gemm(engine.split_operand(h, dtype=dt), d.w3, d.c, out_dtype=v.stream, **kw)
                else:
                    y = ops.conv_gemm(engine.to16(h, dt), d.w, d.c, out_dtype=v.stream, **kw)
                h = y.view(nn_, hh // 2, ww // 2, d.c)
        h = engine.run_resnet_plain(v.mid_res[0], h)
        h = _attention(v.att, h, dt)
        h = engine.run_resnet_plain(v.mid_res[1], h)
        nn_, hh, ww, Cc = h.shape
        sc, sh = ops.groupnorm_scale_shift(h, None, nn_, hh * ww, v.norm_out.groups, v.norm_out.eps, v.norm_out.g, v.norm_out.b)
        y = ops.scale_shift_act(h, None, nn_, hh * ww, sc, sh, 1, out_dtype=v.stream).view(nn_, hh, ww, Cc)
        mom = ops.conv_out(y, v.w_out, v.b_out, v.c_mom)                    # fp32 NCHW [n, 2L, h, w]
        return ops.conv_in(mom, v.q_w, v.q_b, v.c_mom, torch.float32)       # quant_conv (1x1) -> NHWC

    def encode(self, x):
        m = self.moments(x).permute(0, 3, 1, 2)
        L = m.shape[1] // 2
        return m[:, :L].contiguous(), m[:, L:].clamp(-30.0, 20.0).contiguous()

    @torch.no_grad()
    def sample(self, x, eps=None, generator=None, scale=1.0):
        """``latent_dist.sample()`` (times ``scale``); eps: the standard-normal draw (n, L, H/8, W/8), drawn here if None."""
        mom = self.moments(x)
        n, h, w, L2 = mom.shape
        if eps is None:
            eps = torch.randn(n, L2 // 2, h, w, device=x.device, dtype=torch.float32, generator=generator)
        return ops.vae_sample(mom, eps.float().contiguous(), scale)


def encode_image(x_input, encoder, eps=None, generator=None):
    """``PanoGenerator.encode_image(x_input, vae)`` (PanoGenerator.py:214-225): (b, l, 3, H, W) images in [-1, 1] ->
    (b, l, 4, H/8, W/8) latents = ``latent_dist.sample() * scaling_factor``."""
    b = x_input.shape[0]
    v = encoder.packed(x_input.device)
    z = encoder.sample(x_input.flatten(0, 1), None if eps is None else eps.flatten(0, 1), generator, v.scaling_factor)
    return z.unflatten(0, (b, -1)).to(x_input.dtype)
