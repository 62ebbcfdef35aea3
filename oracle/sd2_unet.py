"""Plain-PyTorch CPU restatement of the diffusers==0.24.0 SD-2-base UNet module
tree that PanFusion's denoiser drives sub-module by sub-module.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

diffusers is a third-party dependency of the reference that is NOT vendored
under /root/reference and not installed here (environment.yaml:12,
environment_strict.yaml:59), so this file restates its published semantics
 -- PARITY UNPINNED for this file.  What anchors it is the
reference's own call sites: every attribute MultiViewBaseModel touches exists
with the diffusers name (models/pano/MVGenModel.py:19-32,55-60,86-91,98-144,
172-198,210-277,279-294), so the reference's unmodified MultiViewBaseModel
runs on top of this tree (tools/make_golden.py does exactly that).

State-dict keys equal diffusers' (conv_in.weight, down_blocks.0.resnets.0.norm1.weight,
down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight, ...), LoRA as
``<linear>.lora_layer.{down,up}.weight`` (the post-migration layout of
diffusers 0.24, PanoGenerator.py:101-111).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SD2_BASE = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                layers_per_block=2, num_heads=(5, 10, 20, 20), cross_attention_dim=1024,
                norm_num_groups=32, cross_attn_blocks=(True, True, True, False),
                time_embed_dim=None)


class _Sample:
    """Stand-in for diffusers' Transformer2DModelOutput (only ``.sample`` is read,
    MVGenModel.py:106)."""

    def __init__(self, sample):
        self.sample = sample


class LoRALinearLayer(nn.Module):
    def __init__(self, in_f, out_f, rank):
        super().__init__()
        self.down = nn.Linear(in_f, rank, bias=False)
        self.up = nn.Linear(rank, out_f, bias=False)

    def forward(self, x):
        return self.up(self.down(x))


class LoRACompatibleLinear(nn.Linear):
    """y = W x + b + 1.0 * up(down(x)) when a lora_layer is attached."""

    def __init__(self, in_f, out_f, bias=True):
        super().__init__(in_f, out_f, bias=bias)
        self.lora_layer = None

    def set_lora(self, rank):
        self.lora_layer = LoRALinearLayer(self.in_features, self.out_features, rank)

    def forward(self, x):
        y = super().forward(x)
        if self.lora_layer is not None:
            y = y + self.lora_layer(x)
        return y


class Timesteps(nn.Module):
    """Sinusoidal timestep features, flip_sin_to_cos=True, freq_shift=0: [cos | sin]."""

    def __init__(self, num_channels):
        super().__init__()
        self.num_channels = num_channels

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
        arg = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = (nn.Conv2d(in_channels, out_channels, 1)
                              if in_channels != out_channels else None)

    def forward(self, x, temb):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """diffusers Attention with the vanilla AttnProcessor (SURVEY.md §8a note):
    softmax(q k^T * head_dim^-1/2) v, N x N materialised, no bias on q/k/v."""

    def __init__(self, query_dim, cross_attention_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_q = LoRACompatibleLinear(query_dim, inner, bias=False)
        self.to_k = LoRACompatibleLinear(ctx, inner, bias=False)
        self.to_v = LoRACompatibleLinear(ctx, inner, bias=False)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(inner, query_dim), nn.Dropout(0.0)])

    def set_lora(self, rank):
        for lin in (self.to_q, self.to_k, self.to_v, self.to_out[0]):
            lin.set_lora(rank)

    def forward(self, x, context=None):
        context = x if context is None else context
        b, n, _ = x.shape
        h = self.heads

        def split(t):
            return t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)

        q, k, v = split(self.to_q(x)), split(self.to_k(context)), split(self.to_v(context))
        probs = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype),
                              q, k.transpose(-1, -2), beta=0, alpha=self.scale).softmax(dim=-1)
        o = torch.bmm(probs, v).reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """diffusers block: self-attn, text cross-attn, GEGLU FF; pre-LN, eps 1e-5."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    """use_linear_projection=True: GN(eps 1e-6) -> tokens -> Linear -> block ->
    Linear -> image -> + residual."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states=None):
        b, c, h, w = x.shape
        t = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        t = self.proj_in(t)
        for blk in self.transformer_blocks:
            t = blk(t, encoder_hidden_states)
        t = self.proj_out(t)
        return _Sample(t.reshape(b, h, w, c).permute(0, 3, 1, 2) + x)


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = self.out_channels = channels
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = self.out_channels = channels
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self, has_cross_attention):
        super().__init__()
        self.has_cross_attention = has_cross_attention
        self.resnets = nn.ModuleList()
        if has_cross_attention:
            self.attentions = nn.ModuleList()
        self.downsamplers = None
        self.upsamplers = None


class UNet2DConditionModel(nn.Module):
    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, num_heads=(5, 10, 20, 20), cross_attention_dim=1024,
                 norm_num_groups=32, cross_attn_blocks=(True, True, True, False),
                 time_embed_dim=None):
        super().__init__()
        boc = tuple(block_out_channels)
        g = norm_num_groups
        temb = time_embed_dim or boc[0] * 4
        self.config = dict(in_channels=in_channels, out_channels=out_channels,
                           block_out_channels=boc, layers_per_block=layers_per_block,
                           num_heads=tuple(num_heads), cross_attention_dim=cross_attention_dim,
                           norm_num_groups=g, cross_attn_blocks=tuple(cross_attn_blocks),
                           time_embed_dim=temb)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], temb)

        def tfm(ch, heads):
            return Transformer2DModel(heads, ch // heads, ch, cross_attention_dim, g)

        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, out_ch in enumerate(boc):
            blk = _Block(cross_attn_blocks[i])
            for j in range(layers_per_block):
                blk.resnets.append(ResnetBlock2D(ch if j == 0 else out_ch, out_ch, temb, g))
                if blk.has_cross_attention:
                    blk.attentions.append(tfm(out_ch, num_heads[i]))
            if i != len(boc) - 1:
                blk.downsamplers = nn.ModuleList([Downsample2D(out_ch)])
            self.down_blocks.append(blk)
            ch = out_ch

        self.mid_block = _Block(True)
        self.mid_block.resnets.append(ResnetBlock2D(boc[-1], boc[-1], temb, g))
        self.mid_block.attentions.append(tfm(boc[-1], num_heads[-1]))
        self.mid_block.resnets.append(ResnetBlock2D(boc[-1], boc[-1], temb, g))

        self.up_blocks = nn.ModuleList()
        rev = boc[::-1]
        rev_heads = tuple(num_heads)[::-1]
        rev_cross = tuple(cross_attn_blocks)[::-1]
        prev = rev[0]
        for i, out_ch in enumerate(rev):
            skip_last = rev[min(i + 1, len(rev) - 1)]
            blk = _Block(rev_cross[i])
            for j in range(layers_per_block + 1):
                skip = skip_last if j == layers_per_block else out_ch
                inp = prev if j == 0 else out_ch
                blk.resnets.append(ResnetBlock2D(inp + skip, out_ch, temb, g))
                if blk.has_cross_attention:
                    blk.attentions.append(tfm(out_ch, rev_heads[i]))
            if i != len(rev) - 1:
                blk.upsamplers = nn.ModuleList([Upsample2D(out_ch)])
            self.up_blocks.append(blk)
            prev = out_ch

        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def attention_modules(self):
        return [m for m in self.modules() if isinstance(m, Attention)]

    def add_lora(self, rank=4):
        """PanoGenerator.py:132-151: rank-4 LoRA on q/k/v/out of every attention."""
        for a in self.attention_modules():
            a.set_lora(rank)

    def forward(self, sample, timestep, encoder_hidden_states):
        """Plain single-branch UNet forward (used for self-checks and the CPU
        baseline; the dual-branch driver never calls it, like the reference)."""
        emb = self.time_embedding(self.time_proj(timestep).to(self.dtype))
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            for j, res in enumerate(blk.resnets):
                h = res(h, emb)
                if blk.has_cross_attention:
                    h = blk.attentions[j](h, encoder_hidden_states).sample
                skips.append(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0](h)
                skips.append(h)
        h = self.mid_block.resnets[0](h, emb)
        h = self.mid_block.attentions[0](h, encoder_hidden_states).sample
        h = self.mid_block.resnets[1](h, emb)
        for blk in self.up_blocks:
            for j, res in enumerate(blk.resnets):
                h = res(torch.cat([h, skips.pop()], dim=1), emb)
                if blk.has_cross_attention:
                    h = blk.attentions[j](h, encoder_hidden_states).sample
            if blk.upsamplers is not None:
                h = blk.upsamplers[0](h)
        return self.conv_out(self.conv_act(self.conv_norm_out(h)))


class ControlNetConditioningEmbedding(nn.Module):
    """diffusers 0.24 ControlNetConditioningEmbedding: conv_in 3->16, then per level
    (conv c->c, conv c->c' stride 2) over (16, 32, 96, 256), SiLU after every conv but the last,
    zero-initialised conv_out 256 -> block_out_channels[0].  512x1024 image -> 64x128 features."""

    def __init__(self, out_channels, cond_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(cond_channels, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(block_out_channels) - 1):
            ci, co = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(ci, ci, 3, padding=1))
            self.blocks.append(nn.Conv2d(ci, co, 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(block_out_channels[-1], out_channels, 3, padding=1)
        nn.init.zeros_(self.conv_out.weight)
        nn.init.zeros_(self.conv_out.bias)

    def forward(self, cond):
        h = F.silu(self.conv_in(cond))
        for blk in self.blocks:
            h = F.silu(blk(h))
        return self.conv_out(h)


class ControlNetModel(nn.Module):
    """diffusers==0.24.0 ControlNetModel as the reference uses it (PanoGenerator.py:153-157
    ``ControlNetModel.from_unet(unet)``; called at MVGenModel.py:68-83 with
    ``(sample, timestep, encoder_hidden_states=, controlnet_cond=, return_dict=False)``).
    PARITY UNPINNED (diffusers is not available here): restated from the pinned version's
    published behaviour: encoder copy of the UNet (conv_in, time embedding,
    down blocks, mid block), conditioning embedding ADDED to conv_in(sample), one zero-initialised
    1x1 conv per skip tensor (12) and one for the mid output; conditioning_scale 1.0, no guess mode.
    Plain zero-padded convolutions on the UN-padded panorama latent (no circular padding)."""

    def __init__(self, in_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 num_heads=(5, 10, 20, 20), cross_attention_dim=1024, norm_num_groups=32,
                 cross_attn_blocks=(True, True, True, False), time_embed_dim=None,
                 conditioning_embedding_out_channels=(16, 32, 96, 256), **_ignored):
        super().__init__()
        boc = tuple(block_out_channels)
        g = norm_num_groups
        temb = time_embed_dim or boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(
            boc[0], 3, tuple(conditioning_embedding_out_channels))

        def tfm(ch, heads):
            return Transformer2DModel(heads, ch // heads, ch, cross_attention_dim, g)

        def zero_conv(ch):
            c = nn.Conv2d(ch, ch, 1)
            nn.init.zeros_(c.weight)
            nn.init.zeros_(c.bias)
            return c

        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([zero_conv(boc[0])])
        ch = boc[0]
        for i, out_ch in enumerate(boc):
            blk = _Block(cross_attn_blocks[i])
            for j in range(layers_per_block):
                blk.resnets.append(ResnetBlock2D(ch if j == 0 else out_ch, out_ch, temb, g))
                if blk.has_cross_attention:
                    blk.attentions.append(tfm(out_ch, num_heads[i]))
                self.controlnet_down_blocks.append(zero_conv(out_ch))
            if i != len(boc) - 1:
                blk.downsamplers = nn.ModuleList([Downsample2D(out_ch)])
                self.controlnet_down_blocks.append(zero_conv(out_ch))
            self.down_blocks.append(blk)
            ch = out_ch
        self.mid_block = _Block(True)
        self.mid_block.resnets.append(ResnetBlock2D(boc[-1], boc[-1], temb, g))
        self.mid_block.attentions.append(tfm(boc[-1], num_heads[-1]))
        self.mid_block.resnets.append(ResnetBlock2D(boc[-1], boc[-1], temb, g))
        self.controlnet_mid_block = zero_conv(boc[-1])

    @classmethod
    def from_unet(cls, unet, conditioning_embedding_out_channels=(16, 32, 96, 256)):
        cn = cls(conditioning_embedding_out_channels=conditioning_embedding_out_channels, **unet.config)
        for name in ("conv_in", "time_embedding", "down_blocks", "mid_block"):   # load_weights_from_unet=True
            getattr(cn, name).load_state_dict(getattr(unet, name).state_dict(), strict=False)
        return cn

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None, return_dict=False):
        emb = self.time_embedding(self.time_proj(timestep).to(self.dtype))
        h = self.conv_in(sample) + self.controlnet_cond_embedding(controlnet_cond)
        skips = [h]
        for blk in self.down_blocks:
            for j, res in enumerate(blk.resnets):
                h = res(h, emb)
                if blk.has_cross_attention:
                    h = blk.attentions[j](h, encoder_hidden_states).sample
                skips.append(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0](h)
                skips.append(h)
        h = self.mid_block.resnets[0](h, emb)
        h = self.mid_block.attentions[0](h, encoder_hidden_states).sample
        h = self.mid_block.resnets[1](h, emb)
        down = tuple(z(s) for z, s in zip(self.controlnet_down_blocks, skips))
        return down, self.controlnet_mid_block(h)


def tiny_config(width=32, cross_attention_dim=64, heads=(1, 2, 4, 4), groups=8):
    """Same topology as SD-2-base, small widths, for CPU-sized parity cases."""
    return dict(in_channels=4, out_channels=4,
                block_out_channels=(width, 2 * width, 4 * width, 4 * width),
                layers_per_block=2, num_heads=heads, cross_attention_dim=cross_attention_dim,
                norm_num_groups=groups, cross_attn_blocks=(True, True, True, False))


@torch.no_grad()
def init_synthetic(module, seed, gain=1.0):
    """Seeded synthetic weights (no SD-2 checkpoint is available offline):
    fan-in-scaled normals for matrices/convs, norm gamma ~ 1 + 0.1 N, small
    random biases; zero-initialised layers (EPA to_out / FF out, LoRA up) are
    re-randomised so that no block degenerates to the identity.  CPU generator
    => identical on every machine."""
    gen = torch.Generator().manual_seed(seed)
    for name, p in sorted(module.named_parameters()):
        if p.dim() >= 2:
            fan_in = p[0].numel()
            std = gain / math.sqrt(fan_in)
            if "lora_layer" in name:
                std = 0.3 / math.sqrt(fan_in)
            p.copy_(torch.randn(p.shape, generator=gen) * std)
        elif name.endswith("weight"):      # norm gamma
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen))
        else:                               # biases / norm beta
            p.copy_(0.05 * torch.randn(p.shape, generator=gen))
    return module
