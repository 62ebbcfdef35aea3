"""Parameter tree of the DECODER half of diffusers ``AutoencoderKL`` (SD-2 VAE) with diffusers' attribute and
state-dict names, without forward code -- the counterpart of ``sd2_unet_params.UNetParams`` for the step after
the sampling loop (reference ``models/pano/PanoGenerator.py:213-220``, ``PanFusion.py:166-172``).

When diffusers is installed, pass its ``AutoencoderKL`` straight to ``panfusion_amd.vae.VAEDecoder``; this
container exists for environments without diffusers (``load_state_dict(vae_state_dict, strict=False)`` ignores
the encoder keys).
"""
import torch.nn as nn

from .sd2_unet_params import _Holder

SD2_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
               norm_num_groups=32, scaling_factor=0.18215)


def _resnet(cin, cout, groups):
    r = _Holder()
    r.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
    r.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
    r.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
    r.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
    r.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
    return r


class _Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class VAEDecoderParams(_Holder):
    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.18215):
        super().__init__()
        boc, g = tuple(block_out_channels), norm_num_groups
        self.config = _Config(scaling_factor=scaling_factor, latent_channels=latent_channels, block_out_channels=boc,
                              layers_per_block=layers_per_block, norm_num_groups=g, out_channels=out_channels)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        d = _Holder()
        top = boc[-1]
        d.conv_in = nn.Conv2d(latent_channels, top, 3, padding=1)
        mid = _Holder()
        att = _Holder()
        att.group_norm = nn.GroupNorm(g, top, eps=1e-6)
        att.to_q, att.to_k, att.to_v = nn.Linear(top, top), nn.Linear(top, top), nn.Linear(top, top)
        att.to_out = nn.ModuleList([nn.Linear(top, top), nn.Dropout(0.0)])
        mid.attentions = nn.ModuleList([att])
        mid.resnets = nn.ModuleList([_resnet(top, top, g), _resnet(top, top, g)])
        d.mid_block = mid
        d.up_blocks = nn.ModuleList()
        prev = top
        for i, ch in enumerate(boc[::-1]):
            b = _Holder()
            b.resnets = nn.ModuleList([_resnet(prev if j == 0 else ch, ch, g) for j in range(layers_per_block + 1)])
            b.upsamplers = None
            if i != len(boc) - 1:
                up = _Holder()
                up.conv = nn.Conv2d(ch, ch, 3, padding=1)
                b.upsamplers = nn.ModuleList([up])
            d.up_blocks.append(b)
            prev = ch
        d.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        d.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)
        self.decoder = d

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype


class VAEEncoderParams(_Holder):
    """The ENCODER half (``encoder`` + ``quant_conv``) with diffusers' names: what ``PanoGenerator.encode_image``
    (PanoGenerator.py:214-225) runs on the views and the padded panorama at the top of every training step."""

    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.18215):
        super().__init__()
        boc, g = tuple(block_out_channels), norm_num_groups
        self.config = _Config(scaling_factor=scaling_factor, latent_channels=latent_channels, block_out_channels=boc,
                              layers_per_block=layers_per_block, norm_num_groups=g, out_channels=out_channels)
        e = _Holder()
        e.conv_in = nn.Conv2d(out_channels, boc[0], 3, padding=1)
        e.down_blocks = nn.ModuleList()
        prev = boc[0]
        for i, ch in enumerate(boc):
            b = _Holder()
            b.resnets = nn.ModuleList([_resnet(prev if j == 0 else ch, ch, g) for j in range(layers_per_block)])
            b.downsamplers = None
            if i != len(boc) - 1:
                dn = _Holder()
                dn.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)
                b.downsamplers = nn.ModuleList([dn])
            e.down_blocks.append(b)
            prev = ch
        top = boc[-1]
        mid = _Holder()
        att = _Holder()
        att.group_norm = nn.GroupNorm(g, top, eps=1e-6)
        att.to_q, att.to_k, att.to_v = nn.Linear(top, top), nn.Linear(top, top), nn.Linear(top, top)
        att.to_out = nn.ModuleList([nn.Linear(top, top), nn.Dropout(0.0)])
        mid.attentions = nn.ModuleList([att])
        mid.resnets = nn.ModuleList([_resnet(top, top, g), _resnet(top, top, g)])
        e.mid_block = mid
        e.conv_norm_out = nn.GroupNorm(g, top, eps=1e-6)
        e.conv_out = nn.Conv2d(top, 2 * latent_channels, 3, padding=1)
        self.encoder = e
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype
