"""Plain-PyTorch CPU restatement of diffusers==0.24.0 ``AutoencoderKL`` (decoder; encoder for the training step) as the
reference uses it after the sampling loop.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Reference call sites (paths relative to /root/reference):
  * models/pano/PanoGenerator.py:213-220 ``decode_latent``: ``1 / vae.config.scaling_factor * latents``,
    ``(b m) c h w``, ``vae.decode(latents.to(vae.dtype)).sample``;
  * models/pano/PanFusion.py:166-172: the 20 view latents, then the panorama latent circularly padded by
    ``latent_pad`` columns (PanoGenerator.py:222-234), decoded and cropped by ``8 * latent_pad`` pixels;
  * models/modules/utils.py:9-15 ``tensor_to_image``: ``(x / 2 + 0.5).clamp(0, 1) * 255``, round, uint8, HWC.
The VAE is loaded from ``stabilityai/stable-diffusion-2-base`` (PanoGenerator.py:120-125, fp16).

diffusers is NOT vendored under /root/reference and not installed here: PARITY UNPINNED for this file.  It
restates the published semantics of the pinned version for the SD-2 VAE config (``block_out_channels
(128, 256, 512, 512)``, ``layers_per_block 2`` -> 3 resnets per decoder block, ``latent_channels 4``,
``norm_num_groups 32``, GroupNorm eps 1e-6, SiLU, ``scaling_factor 0.18215``, mid-block attention with ONE head
of width 512):
  decode(z) = decoder(post_quant_conv(z));
  decoder: conv_in 3x3 -> mid (resnet, attention, resnet) -> 4 up blocks (3 resnets each, nearest x2 + conv 3x3
  after the first three) -> GroupNorm -> SiLU -> conv_out 3x3;
  resnet (no time embedding): x + conv2(silu(GN(conv1(silu(GN(x)))))), 1x1 ``conv_shortcut`` when channels change;
  attention: h = GN(x) as tokens; softmax(q k^T / sqrt(C)) v with biased q/k/v/out projections; + x.
State-dict keys equal diffusers' (``post_quant_conv.weight``, ``decoder.mid_block.attentions.0.to_q.weight``,
``decoder.up_blocks.2.resnets.0.conv_shortcut.weight``, ...).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

SD2_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
               norm_num_groups=32, scaling_factor=0.18215)


class VAEResnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class VAEAttention(nn.Module):
    """diffusers Attention(heads=1, dim_head=C, bias=True, norm_num_groups, residual_connection=True)."""

    def __init__(self, ch, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(b, c, h, w) + x


class _Up(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.upsamplers = None


class Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        boc = tuple(block_out_channels)
        top = boc[-1]
        self.conv_in = nn.Conv2d(latent_channels, top, 3, padding=1)
        self.mid_block = _Block()
        self.mid_block.attentions = nn.ModuleList([VAEAttention(top, groups)])
        self.mid_block.resnets.append(VAEResnet(top, top, groups))
        self.mid_block.resnets.append(VAEResnet(top, top, groups))
        self.up_blocks = nn.ModuleList()
        prev = top
        for i, ch in enumerate(boc[::-1]):
            blk = _Block()
            for j in range(layers_per_block + 1):
                blk.resnets.append(VAEResnet(prev if j == 0 else ch, ch, groups))
            if i != len(boc) - 1:
                blk.upsamplers = nn.ModuleList([_Up(ch)])
            self.up_blocks.append(blk)
            prev = ch
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid_block.resnets[0](h)
        h = self.mid_block.attentions[0](h)
        h = self.mid_block.resnets[1](h)
        for blk in self.up_blocks:
            for r in blk.resnets:
                h = r(h)
            if blk.upsamplers is not None:
                h = blk.upsamplers[0](h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class _Down(nn.Module):
    """diffusers Downsample2D(use_conv=True, padding=0): F.pad(x, (0, 1, 0, 1)) then Conv2d(ch, ch, 3, stride=2, padding=0)."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Encoder(nn.Module):
    """diffusers Encoder of AutoencoderKL (DownEncoderBlock2D x len(block_out_channels), `layers_per_block` resnets each,
    a down-sampler after all but the last; UNetMidBlock2D; GroupNorm -> SiLU -> conv_out to 2 * latent_channels)."""

    def __init__(self, in_channels, latent_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        prev = boc[0]
        for i, ch in enumerate(boc):
            blk = _Block()
            for j in range(layers_per_block):
                blk.resnets.append(VAEResnet(prev if j == 0 else ch, ch, groups))
            blk.downsamplers = nn.ModuleList([_Down(ch)]) if i != len(boc) - 1 else None
            self.down_blocks.append(blk)
            prev = ch
        top = boc[-1]
        self.mid_block = _Block()
        self.mid_block.attentions = nn.ModuleList([VAEAttention(top, groups)])
        self.mid_block.resnets.append(VAEResnet(top, top, groups))
        self.mid_block.resnets.append(VAEResnet(top, top, groups))
        self.conv_norm_out = nn.GroupNorm(groups, top, eps=1e-6)
        self.conv_out = nn.Conv2d(top, 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for blk in self.down_blocks:
            for r in blk.resnets:
                h = r(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0](h)
        h = self.mid_block.resnets[0](h)
        h = self.mid_block.attentions[0](h)
        h = self.mid_block.resnets[1](h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class DiagonalGaussian:
    """diffusers DiagonalGaussianDistribution: moments (n, 2L, h, w) = (mean | logvar), logvar clamped to [-30, 20]."""

    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, eps=None):
        if eps is None:
            eps = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


class _Latent:
    def __init__(self, dist):
        self.latent_dist = dist


class _Sample:
    def __init__(self, sample):
        self.sample = sample


class _Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class AutoencoderKLDecoder(nn.Module):
    """``post_quant_conv`` + ``decoder`` with diffusers' names; ``.decode(z).sample``, ``.config.scaling_factor``
    and ``.dtype`` are what the reference touches (PanoGenerator.py:213-220)."""

    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.18215):
        super().__init__()
        self.config = _Config(scaling_factor=scaling_factor, latent_channels=latent_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              norm_num_groups=norm_num_groups, out_channels=out_channels)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        # the encoder half (training: PanoGenerator.encode_image, PanoGenerator.py:214-225)
        self.encoder = Encoder(out_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def decode(self, z):
        return _Sample(self.decoder(self.post_quant_conv(z)))

    def encode(self, x):
        return _Latent(DiagonalGaussian(self.quant_conv(self.encoder(x))))


AutoencoderKL = AutoencoderKLDecoder          # (the class carries both halves since the training row)


def tiny_vae_config(width=32, groups=8):
    """Same topology as the SD-2 VAE decoder at small widths (CPU-sized parity cases); widths are multiples of
    64 from the second level so that every MFMA-kernel layer of the product has a legal channel count."""
    return dict(latent_channels=4, out_channels=3, block_out_channels=(width, 2 * width, 4 * width, 4 * width),
                layers_per_block=2, norm_num_groups=groups, scaling_factor=0.18215)


def decode_latent(latents, vae):
    """PanoGenerator.py:213-220 on (b, m, c, h, w) latents -> (b, m, 3, 8h, 8w)."""
    b = latents.shape[0]
    z = (1 / vae.config.scaling_factor * latents).flatten(0, 1)
    image = vae.decode(z.to(vae.dtype)).sample
    return image.unflatten(0, (b, -1)).to(latents.dtype)


def decode_views_and_pano(latents, pano_latent, vae, latent_pad=8):
    """The tail of PanFusion.inference (PanFusion.py:166-172): views decoded as they are; the panorama latent is
    circularly padded by ``latent_pad`` columns, decoded, and the image cropped by 8 * latent_pad pixels."""
    from . import geometry as G
    images = decode_latent(latents, vae)
    pano = G.unpad_pano(decode_latent(G.pad_pano(pano_latent, latent_pad), vae), 8 * latent_pad)
    return images, pano


def encode_image(x_input, vae, generator=None, eps=None):
    """PanoGenerator.py:214-225 on (b, l, 3, H, W) images in [-1, 1] -> (b, l, 4, H/8, W/8) latents:
    ``vae.encode(x).latent_dist.sample() * scaling_factor`` (eps: the normal draw, for a deterministic comparison)."""
    b = x_input.shape[0]
    z = vae.encode(x_input.flatten(0, 1).to(vae.dtype)).latent_dist.sample(generator=generator, eps=eps)
    return (z * vae.config.scaling_factor).unflatten(0, (b, -1)).to(x_input.dtype)


def tensor_to_image(image):
    """models/modules/utils.py:9-15: float image in [-1, 1] -> uint8 (... h w c)."""
    image = ((image / 2 + 0.5).clamp(0, 1) * 255).round()
    return image.to(torch.uint8).movedim(-3, -1).contiguous()
