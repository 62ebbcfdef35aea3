"""CPU restatement of PanFusion's dual-branch denoiser and EPA fusion block.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Follows (paths relative to /root/reference):
  * models/pano/MVGenModel.py:9-297   -> DualBranchDenoiser
  * models/pano/modules.py:8-59       -> EPABlock  (reference name: WarpAttn)
  * models/modules/transformer.py:8-74,130-162 -> _EPATransformer and friends
Parameter names equal the reference's, so state dicts move both ways; pinned
against the imported reference by tests/test_oracle_vs_reference.py and the
fixtures tools/make_golden.py wrote with the reference's own classes.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import geometry as G
from . import third_party as tp


# ----------------------------------------------------------------------------
# EPA transformer block (transformer.py)
# ----------------------------------------------------------------------------
class _BiasedCrossAttention(nn.Module):
    """transformer.py:40-74: bias-free q/k/v, heads of width 32, additive bias
    shared by all heads, output projection zero-initialised."""

    def __init__(self, dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Linear(inner, dim)
        nn.init.zeros_(self.to_out.weight)
        nn.init.zeros_(self.to_out.bias)

    def forward(self, x, context, bias):
        b, n, _ = x.shape
        h = self.heads

        def heads_first(t):
            return t.reshape(b, t.shape[1], h, -1).transpose(1, 2).reshape(b * h, t.shape[1], -1)

        q, k, v = heads_first(self.to_q(x)), heads_first(self.to_k(context)), heads_first(self.to_v(context))
        bias = bias.repeat_interleave(h, dim=0)
        o = tp.memory_efficient_attention(q, k, v, attn_bias=bias)
        o = o.reshape(b, h, n, -1).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(o)


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        a, gate = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(gate)


class _FeedForward(nn.Module):
    """transformer.py:18-37 (glu=True): GEGLU(dim, 4 dim) -> Linear zero-init."""

    def __init__(self, dim):
        super().__init__()
        last = nn.Linear(dim * 4, dim)
        nn.init.zeros_(last.weight)
        nn.init.zeros_(last.bias)
        self.net = nn.Sequential(_GEGLU(dim, dim * 4), nn.Dropout(0.0), last)

    def forward(self, x):
        return self.net(x)


class _EPATransformer(nn.Module):
    """transformer.py:130-162: ONE LayerNorm (norm1) normalises both the
    PE-augmented query and the context."""

    def __init__(self, dim):
        super().__init__()
        self.attn1 = _BiasedCrossAttention(dim, dim // 32, 32)
        self.ff = _FeedForward(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)

    def forward(self, x, context, bias, query_pe):
        q = self.norm1(x + query_pe)
        c = self.norm1(context)
        x = self.attn1(q, c, bias) + x
        return self.ff(self.norm2(x)) + x


class _PE(nn.Module):
    def __init__(self, n_freqs):
        super().__init__()
        self.register_buffer("freq_bands", G.spherical_freq_bands(n_freqs))

    def forward(self, coords):
        return G.spherical_pe(coords, self.freq_bands)


class EPABlock(nn.Module):
    """Equirectangular-Perspective Attention (modules.py:8-59).

    pers_x (b*m, C, h, w), equi_x (b, C, H, W), cameras dict of (b*m,) ->
    same shapes.  Both directions share the transformer; the second direction
    queries with the ORIGINAL pers_x (modules.py:51-55).
    """

    def __init__(self, dim):
        super().__init__()
        self.transformer = _EPATransformer(dim)
        self.pe = _PE(dim // 4)

    def forward(self, pers_x, equi_x, cameras):
        bm, c, ph, pw = pers_x.shape
        b, _, eh, ew = equi_x.shape
        m = bm // b
        bias_e, bias_p = G.get_masks(ph, pw, eh, ew, cameras, dtype=pers_x.dtype)
        coords_p, coords_e = G.get_coords(ph, pw, eh, ew, cameras, dtype=pers_x.dtype)

        pe_p = self.pe(coords_p).reshape(b, m * ph * pw, c)                  # (b, mP, C)
        pe_e = self.pe(coords_e).reshape(1, eh * ew, c).expand(b, -1, -1)    # (b, E, C)
        tok_p = pers_x.reshape(b, m, c, ph * pw).permute(0, 1, 3, 2).reshape(b, m * ph * pw, c)
        tok_e = equi_x.reshape(b, c, eh * ew).transpose(1, 2)

        # panorama queries the views: bias (b, E, m*P)
        be = bias_e.reshape(b, m, eh * ew, ph * pw).permute(0, 2, 1, 3).reshape(b, eh * ew, m * ph * pw)
        out_e = self.transformer(tok_e, tok_p + pe_p, be, pe_e)
        # views query the panorama: bias (b, m*P, E)
        bp = bias_p.reshape(b, m * ph * pw, eh * ew)
        out_p = self.transformer(tok_p, tok_e + pe_e, bp, pe_p)

        out_p = out_p.reshape(b, m, ph * pw, c).permute(0, 1, 3, 2).reshape(bm, c, ph, pw)
        out_e = out_e.transpose(1, 2).reshape(b, c, eh, ew)
        return out_p, out_e


# ----------------------------------------------------------------------------
# dual-branch denoiser (MVGenModel.py)
# ----------------------------------------------------------------------------
class _Branch:
    """One UNet being driven layer by layer.  ``wrap`` = number of circularly
    padded columns the pano branch adds around each conv-bearing module
    (MVGenModel.py:110-115 etc.); 0 for the perspective branch."""

    def __init__(self, unet, x, timestep, text, pano, pad):
        self.u, self.text, self.pano, self.pad = unet, text, pano, pad and pano
        self.temb = unet.time_embedding(unet.time_proj(timestep).to(unet.dtype))
        self.h = self._wrapped(unet.conv_in, x, 1, 1)
        self.skips = [self.h]

    def _wrapped(self, fn, x, pad_in, crop_out, *args):
        if not self.pad:
            return fn(x, *args)
        return G.unpad_pano(fn(G.pad_pano(x, pad_in), *args), crop_out)

    def resnet(self, res, skip=False):
        x = self.h
        if skip:
            x = torch.cat([x, self.skips.pop()], dim=1)
        self.h = self._wrapped(res, x, 2, 2, self.temb)

    def attention(self, attn):
        self.h = attn(self.h, encoder_hidden_states=self.text).sample

    def push(self):
        self.skips.append(self.h)

    def downsample(self, down):          # pad 2, stride-2 conv, crop 1  (:138-144)
        self.h = self._wrapped(down, self.h, 2, 1)

    def upsample(self, up):              # pad 1, nearest x2 + conv, crop 2  (:272-277)
        self.h = self._wrapped(up, self.h, 1, 2)

    def head(self):                      # norm/act un-padded, conv_out padded by 1 (:279-294)
        y = self.u.conv_act(self.u.conv_norm_out(self.h))
        return self._wrapped(self.u.conv_out, y, 1, 1)

    def add_to_skips(self, residuals):
        self.skips = [s + r for s, r in zip(self.skips, residuals)]


class DualBranchDenoiser(nn.Module):
    """Oracle for MultiViewBaseModel: same constructor, attributes and forward
    signature (MVGenModel.py:9-39)."""

    def __init__(self, unet, pano_unet, pers_cn=None, pano_cn=None, pano_pad=True):
        super().__init__()
        self.unet, self.pano_unet = unet, pano_unet
        self.pers_cn, self.pano_cn, self.pano_pad = pers_cn, pano_cn, pano_pad
        if unet is not None:
            self.cp_blocks_encoder = nn.ModuleList(
                [EPABlock(blk.downsamplers[-1].out_channels)
                 for blk in unet.down_blocks if blk.downsamplers is not None])
            self.cp_blocks_mid = EPABlock(unet.mid_block.resnets[-1].out_channels)
            self.cp_blocks_decoder = nn.ModuleList(
                [EPABlock(blk.upsamplers[0].channels)
                 for blk in unet.up_blocks if blk.upsamplers is not None])
            self.trainable_parameters = [(list(self.cp_blocks_mid.parameters())
                                          + list(self.cp_blocks_decoder.parameters())
                                          + list(self.cp_blocks_encoder.parameters()), 1.0)]

    def forward(self, latents, pano_latent, timestep, prompt_embd, pano_prompt_embd, cameras,
                pers_layout_cond=None, pano_layout_cond=None):
        two = self.unet is not None
        branches = []
        if two:
            b, m = latents.shape[:2]
            cameras = {k: v.flatten(0, 1) for k, v in cameras.items()}
            pers = _Branch(self.unet, latents.flatten(0, 1), timestep.reshape(-1),
                           prompt_embd.flatten(0, 1), pano=False, pad=False)
            pano_t = timestep[:, 0]
            branches.append(pers)
        else:
            pano_t = timestep
        flat_pano = pano_latent.flatten(0, 1)
        flat_pano_text = pano_prompt_embd.flatten(0, 1)

        cn_out = {}
        if two and self.pers_cn is not None and pers_layout_cond is not None:
            cn_out[id(pers)] = self.pers_cn(latents.flatten(0, 1), timestep.reshape(-1),
                                            encoder_hidden_states=pers.text,
                                            controlnet_cond=pers_layout_cond.flatten(0, 1),
                                            return_dict=False)
        pano = _Branch(self.pano_unet, flat_pano, pano_t, flat_pano_text, pano=True, pad=self.pano_pad)
        if self.pano_cn is not None and pano_layout_cond is not None:
            cn_out[id(pano)] = self.pano_cn(flat_pano, pano_t, encoder_hidden_states=flat_pano_text,
                                            controlnet_cond=pano_layout_cond.flatten(0, 1),
                                            return_dict=False)
        branches.append(pano)

        def fuse(block):
            if two:
                pers.h, pano.h = block(pers.h, pano.h, cameras)

        # encoder: resnet (-> attention) per layer, downsample, EPA after each downsample (:98-152)
        for i in range(len(self.pano_unet.down_blocks)):
            for br in branches:
                blk = br.u.down_blocks[i]
                for j, res in enumerate(blk.resnets):
                    br.resnet(res)
                    if blk.has_cross_attention:
                        br.attention(blk.attentions[j])
                    br.push()
                if blk.downsamplers is not None:
                    br.downsample(blk.downsamplers[0])
                    br.push()
            if self.pano_unet.down_blocks[i].downsamplers is not None:
                fuse(self.cp_blocks_encoder[i] if two else None)

        for br in branches:              # ControlNet residuals onto the skip stack (:154-170)
            if id(br) in cn_out:
                br.add_to_skips(cn_out[id(br)][0])

        # mid: resnet, attention, resnet; EPA (:172-207)
        for br in branches:
            mid = br.u.mid_block
            br.resnet(mid.resnets[0])
            for a, res in zip(mid.attentions, mid.resnets[1:]):
                br.attention(a)
                br.resnet(res)
            if id(br) in cn_out:
                br.h = br.h + cn_out[id(br)][1]
        fuse(self.cp_blocks_mid if two else None)

        # decoder: (skip-cat, resnet, attention) x3, EPA BEFORE each upsample (:210-277)
        for i in range(len(self.pano_unet.up_blocks)):
            for br in branches:
                blk = br.u.up_blocks[i]
                for j, res in enumerate(blk.resnets):
                    br.resnet(res, skip=True)
                    if blk.has_cross_attention:
                        br.attention(blk.attentions[j])
            if self.pano_unet.up_blocks[i].upsamplers is not None:
                fuse(self.cp_blocks_decoder[i] if two else None)
                for br in branches:
                    br.upsample(br.u.up_blocks[i].upsamplers[0])

        pano_sample = pano.head().unflatten(0, (-1, 1))
        sample = pers.head().unflatten(0, (b, m)) if two else None
        return sample, pano_sample


@torch.no_grad()
def randomize_epa(module, seed, gain=0.5):
    """Re-randomise the zero-initialised EPA projections (transformer.py:29-30,
    54-55) so the block is not the identity in parity tests."""
    gen = torch.Generator().manual_seed(seed)
    for name, p in sorted(module.named_parameters()):
        if name.endswith(("attn1.to_out.weight", "ff.net.2.weight")):
            p.copy_(torch.randn(p.shape, generator=gen) * gain / p.shape[1] ** 0.5)
        elif name.endswith(("attn1.to_out.bias", "ff.net.2.bias")):
            p.copy_(0.05 * torch.randn(p.shape, generator=gen))
    return module
